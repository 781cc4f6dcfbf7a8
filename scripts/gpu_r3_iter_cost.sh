# where the per-iteration cost sits: unfused kernel-trace stats of the 4K frame at two workspace sizes
tag=${1:-r3iter}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for gb in 8 96; do
  RT_FUSE=${FUSE:-0} RT_WORKSPACE_GB=$gb RT_VERBOSE=1 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace$gb -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-roofline > gpurun_out/$tag/trace$gb.log 2>&1 || { tail -5 gpurun_out/$tag/trace$gb.log; exit 1; }
  find gpurun_out/$tag/trace$gb -name "*kernel_stats.csv" -exec cp {} gpurun_out/$tag/kernel_stats_$gb.csv \;
  rm -rf gpurun_out/$tag/trace$gb
  grep "frame done" gpurun_out/$tag/trace$gb.log | tail -1
  head -9 gpurun_out/$tag/kernel_stats_$gb.csv | cut -d, -f1-4 | cut -c1-150
done
