# kernel trace of one bench.py configuration: bash scripts/gpu_trace_config.sh <tag> <bench.py args...>
tag=$1; shift
out=gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu --no-roofline > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out -name "*_kernel_trace.csv" -delete
cut -c1-160 $out/kernel_stats.csv | head -9
