# round 3, first contact of the pipelined frame loop with the GPU: the parity-critical tests, then a short bench
set -e
tag=${1:-r3first}
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_chunks.py tests/test_zz_no_alloc.py tests/test_gpu_split.py -x -q --timeout 200 --timeout-method thread > gpurun_out/$tag/gpu_tests.log 2>&1 || { tail -40 gpurun_out/$tag/gpu_tests.log; exit 1; }
tail -3 gpurun_out/$tag/gpu_tests.log
RT_VERBOSE=1 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-cpu > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { tail -20 gpurun_out/$tag/bench.err; exit 1; }
tail -12 gpurun_out/$tag/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/$tag/bench.json"))
print("Mrays/s", round(d["value"]), "ms/frame", round(d["ms_per_step"],1), {k: round(v,1) for k,v in d.get("kernel_ms",{}).items()})
r=d.get("roofline",{}); print({k:r.get(k) for k in ("kernel","achieved","frac","records_per_ray")}, r.get("lanes"))
PY
