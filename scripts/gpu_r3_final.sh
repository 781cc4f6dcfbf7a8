# round-3 evidence in one call: GPU tests, kernel-trace stats + HBM traffic passes, SQ issue passes, cache pass, the bench line,
# the workspace sweep, the per-rank shares, the multi-GPU rehearsals.  Outputs under gpurun_out/$1 (C5: scripts/gpu_r3_c5.sh).
tag=${1:-r3_final}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/$tag/gpu_tests.log 2>&1 || { tail -40 gpurun_out/$tag/gpu_tests.log; exit 1; }
tail -2 gpurun_out/$tag/gpu_tests.log
fi
bash scripts/gpu_profile_bench.sh $tag/prof || exit 1
bash scripts/gpu_pmc_valu.sh $tag/valu || exit 1
bash scripts/gpu_pmc_cache.sh $tag/cache || exit 1
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --steps 5 --warmup 2 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { tail -5 gpurun_out/$tag/bench.err; exit 1; }
cat gpurun_out/$tag/bench.json
for gb in 4 8 16 30 60 120; do
  RT_WORKSPACE_GB=$gb timeout -k 10 200 python bench.py --steps 3 --warmup 2 --no-cpu --no-roofline > gpurun_out/$tag/ws_$gb.json 2> gpurun_out/$tag/ws_$gb.err || { tail -5 gpurun_out/$tag/ws_$gb.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/$tag/ws_$gb.json')); print('workspace $gb GB:', round(d['ms_per_step'],1), 'ms', d['config']['chunks_per_frame'], 'chunks', round(d['value']), 'Mrays/s')"
done
timeout -k 10 300 python scripts/gpu_rank_share.py > gpurun_out/$tag/rank_share.txt 2>&1 || { tail -5 gpurun_out/$tag/rank_share.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/$tag/rank_share.txt
timeout -k 10 300 python bench.py --multi-abi --gpus 1 --steps 3 --warmup 2 > gpurun_out/$tag/multi_abi_1.json 2> gpurun_out/$tag/multi_abi_1.err || { tail -5 gpurun_out/$tag/multi_abi_1.err; exit 1; }
cut -c1-900 gpurun_out/$tag/multi_abi_1.json
