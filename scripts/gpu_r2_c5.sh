set -e
tag=${1:-r2_c5}
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -u -m pytest tests/test_gpu_c5.py -m gpu -x -v -s --timeout 400 --timeout-method thread > gpurun_out/$tag/c5.log 2>&1 || { tail -40 gpurun_out/$tag/c5.log; exit 1; }
grep -E "stress|passed|PASSED" gpurun_out/$tag/c5.log
