# GPU tests + headline bench; outputs under gpurun_out/$1
set -e
tag=${1:-run}
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/$tag/gpu_tests.log 2>&1 || { tail -30 gpurun_out/$tag/gpu_tests.log; exit 1; }
tail -3 gpurun_out/$tag/gpu_tests.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { tail -20 gpurun_out/$tag/bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/$tag/bench.json"))
print("Mrays/s", round(d["value"]), "ms/frame", round(d["ms_per_step"],1), {k: round(v,1) for k,v in d.get("kernel_ms",{}).items()})
r=d.get("roofline",{}); print({k:r.get(k) for k in ("kernel","achieved","frac","records_per_ray")}, r.get("lanes"))
PY
