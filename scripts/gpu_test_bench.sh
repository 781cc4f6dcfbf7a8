# GPU tests + headline bench; outputs under gpurun_out/$1
set -e
tag=${1:-run}
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/$tag/gpu_tests.log 2>&1 || { tail -30 gpurun_out/$tag/gpu_tests.log; exit 1; }
tail -3 gpurun_out/$tag/gpu_tests.log
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/$tag/bench.json"))
print("Mrays/s", round(d["value"]), "ms/frame", round(d["ms_per_step"],1), d.get("kernel_ms"))
r=d.get("roofline",{}); print({k:r.get(k) for k in ("kernel","achieved","frac","records_per_ray","traffic")}, d.get("roofline_valu",{}).get("frac"))
PY
