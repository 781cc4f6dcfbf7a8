# the GPU suite with -s: every parity test prints what it achieved -> gpurun_out/$1/margins.log
set -e
tag=${1:-margins}
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v -s --timeout 300 --timeout-method thread > gpurun_out/$tag/gpu_tests_verbose.log 2>&1 || { tail -40 gpurun_out/$tag/gpu_tests_verbose.log; exit 1; }
grep -E "primid|psnr|prim ids|sampled tiles|shadow rays traced|PASSED|passed|FAILED" gpurun_out/$tag/gpu_tests_verbose.log > gpurun_out/$tag/margins.log
tail -3 gpurun_out/$tag/margins.log
