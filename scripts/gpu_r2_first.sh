# round 2, first GPU call: the GPU suite (verbose, -s: the parity margins are printed), the wip no-alloc test, baseline bench + profile
set -e
tag=${1:-r2_first}
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v -s --timeout 300 --timeout-method thread > gpurun_out/$tag/gpu_tests.log 2>&1 || { tail -40 gpurun_out/$tag/gpu_tests.log; exit 1; }
tail -3 gpurun_out/$tag/gpu_tests.log
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -m gpu -x -v -s --timeout 120 --timeout-method thread > gpurun_out/$tag/no_alloc.log 2>&1 || { tail -40 gpurun_out/$tag/no_alloc.log; exit 1; }
tail -3 gpurun_out/$tag/no_alloc.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { tail -20 gpurun_out/$tag/bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/$tag/bench.json"))
print("Mrays/s", round(d["value"]), "ms/frame", round(d["ms_per_step"],1), d.get("kernel_ms"))
PY
timeout -k 10 300 python scripts/gpu_prof.py 1920 1080 16 > gpurun_out/$tag/prof.txt 2>&1 || { tail -20 gpurun_out/$tag/prof.txt; exit 1; }
cat gpurun_out/$tag/prof.txt
