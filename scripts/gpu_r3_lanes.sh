# round 3: two pipelines side by side -- parity-critical tests, bench with 1 / 2 lanes, workspace sweep, N = 8 share
set -e
tag=${1:-r3lanes}
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_chunks.py tests/test_zz_no_alloc.py tests/test_gpu_split.py tests/test_gpu_variants.py -x -q --timeout 200 --timeout-method thread > gpurun_out/$tag/gpu_tests.log 2>&1 || { tail -40 gpurun_out/$tag/gpu_tests.log; exit 1; }
tail -3 gpurun_out/$tag/gpu_tests.log
for lanes in 2 1; do
RT_LANES=$lanes RT_VERBOSE=1 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-cpu > gpurun_out/$tag/bench_lanes$lanes.json 2> gpurun_out/$tag/bench_lanes$lanes.err || { tail -20 gpurun_out/$tag/bench_lanes$lanes.err; exit 1; }
grep "\[rt\] frame" gpurun_out/$tag/bench_lanes$lanes.err | tail -2
python - <<PY
import json
d=json.load(open("gpurun_out/$tag/bench_lanes$lanes.json"))
print("lanes $lanes: Mrays/s", round(d["value"]), "ms/frame", round(d["ms_per_step"],1), d["config"]["chunks_per_frame"], "chunks", {k: round(v,1) for k,v in d.get("kernel_ms",{}).items()})
PY
done
for gb in ${SWEEP:-8 16 32 96}; do
  RT_WORKSPACE_GB=$gb timeout -k 10 120 python bench.py --steps 3 --warmup 2 --no-cpu --no-roofline > gpurun_out/$tag/ws_$gb.json 2> gpurun_out/$tag/ws_$gb.err && echo ws $gb GB: $(python -c "import json;d=json.load(open('gpurun_out/$tag/ws_$gb.json'));print(round(d['ms_per_step'],1), 'ms', d['config']['chunks_per_frame'], 'chunks')") || exit 1
done
timeout -k 10 300 python scripts/gpu_rank_share.py > gpurun_out/$tag/rank_share.txt 2>&1 || { tail -5 gpurun_out/$tag/rank_share.txt; exit 1; }
grep slowest gpurun_out/$tag/rank_share.txt
