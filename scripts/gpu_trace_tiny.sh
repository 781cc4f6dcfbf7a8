# kernel trace of tiny frames: where the fixed per-chunk cost goes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tiny
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tiny/trace -- python3 scripts/gpu_fixed_cost.py > gpurun_out/tiny/log.txt 2>&1 || { tail -5 gpurun_out/tiny/log.txt; exit 1; }
find gpurun_out/tiny/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/tiny/kernel_stats.csv \;
find gpurun_out/tiny -name "*_kernel_trace.csv" -delete
cut -c1-140 gpurun_out/tiny/kernel_stats.csv | head -16
