set -e
tag=${1:-r2_pmc}
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/$tag/gpu_tests.log 2>&1 || { tail -40 gpurun_out/$tag/gpu_tests.log; exit 1; }
tail -2 gpurun_out/$tag/gpu_tests.log
bash scripts/gpu_pmc_valu.sh $tag
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { tail -20 gpurun_out/$tag/bench.err; exit 1; }
cat gpurun_out/$tag/bench.json
