# rocprofv3 evidence for the bench.py workload: kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes -> gpurun_out/$1
tag=${1:-prof_r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -- python3 bench.py --sizing-frame --steps 3 --warmup 0 --no-cpu > gpurun_out/$tag/trace.log 2>&1 || { tail -5 gpurun_out/$tag/trace.log; exit 1; }
timeout -s KILL 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d gpurun_out/$tag/fetch -- python3 bench.py --sizing-frame --steps 1 --warmup 0 --no-cpu --no-roofline > gpurun_out/$tag/fetch.log 2>&1 || { tail -5 gpurun_out/$tag/fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d gpurun_out/$tag/write -- python3 bench.py --sizing-frame --steps 1 --warmup 0 --no-cpu --no-roofline > gpurun_out/$tag/write.log 2>&1 || { tail -5 gpurun_out/$tag/write.log; exit 1; }
python3 scripts/pmc_traffic.py gpurun_out/$tag/fetch gpurun_out/$tag/write "competition scene 3840x2160 @64spp depth 3" > gpurun_out/$tag/pmc_traffic.json
find gpurun_out/$tag/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/$tag/kernel_stats.csv \;
find gpurun_out/$tag -name "*_kernel_trace.csv" -delete; find gpurun_out/$tag -name "*counter_collection.csv" -delete
head -8 gpurun_out/$tag/kernel_stats.csv; grep '"metric"' gpurun_out/$tag/trace.log | cut -c1-200
