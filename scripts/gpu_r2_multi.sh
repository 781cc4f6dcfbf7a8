set -e
tag=${1:-r2_multi}
mkdir -p gpurun_out/$tag
timeout -k 10 300 python -u -m pytest tests/test_gpu_multi.py -m gpu -x -v -s --timeout 200 --timeout-method thread > gpurun_out/$tag/multi.log 2>&1 || { tail -40 gpurun_out/$tag/multi.log; exit 1; }
tail -5 gpurun_out/$tag/multi.log
timeout -k 10 400 python -u scripts/gpu_c5_probe.py 733 > gpurun_out/$tag/c5.log 2>&1 || { tail -20 gpurun_out/$tag/c5.log; exit 1; }
cat gpurun_out/$tag/c5.log
