# round 3: kernel trace of the pipelined frame (kept: launch gaps are read from it), workspace sweep, N = 8 share
tag=${1:-r3sweep}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-roofline > gpurun_out/$tag/trace.log 2>&1 || { tail -5 gpurun_out/$tag/trace.log; exit 1; }
find gpurun_out/$tag/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/$tag/kernel_stats.csv \;
find gpurun_out/$tag/trace -name "*_kernel_trace.csv" -exec cp {} gpurun_out/$tag/kernel_trace.csv \;
rm -rf gpurun_out/$tag/trace
head -12 gpurun_out/$tag/kernel_stats.csv | cut -c1-160
for gb in ${SWEEP:-8 16 32 48 96}; do
  RT_WORKSPACE_GB=$gb timeout -k 10 120 python bench.py --steps 3 --warmup 2 --no-cpu --no-roofline > gpurun_out/$tag/ws_$gb.json 2> gpurun_out/$tag/ws_$gb.err && echo ws $gb GB: $(python -c "import json;d=json.load(open('gpurun_out/$tag/ws_$gb.json'));print(round(d['ms_per_step'],1), 'ms', d['config']['chunks_per_frame'], 'chunks')") || exit 1
done
timeout -k 10 300 python scripts/gpu_rank_share.py > gpurun_out/$tag/rank_share.txt 2>&1 || { tail -5 gpurun_out/$tag/rank_share.txt; exit 1; }
grep slowest gpurun_out/$tag/rank_share.txt
