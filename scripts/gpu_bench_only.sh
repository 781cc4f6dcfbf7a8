set -e
tag=${1:-run}
mkdir -p gpurun_out/$tag
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu ${@:2} > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/$tag/bench.json"))
print("Mrays/s", round(d["value"]), "ms/frame", round(d["ms_per_step"],1), {k:round(v,1) for k,v in d.get("kernel_ms",{}).items()}, d["bvh"])
PY
