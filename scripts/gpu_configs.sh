mkdir -p gpurun_out/configs
timeout -k 10 200 python bench.py --scene sample --width 1920 --height 1080 --spp 16 --steps 3 --warmup 1 --no-cpu > gpurun_out/configs/c2.json 2> gpurun_out/configs/c2.err || exit 1
timeout -k 10 300 python bench.py --scene bunny --width 3840 --height 2160 --spp 64 --steps 2 --warmup 1 --no-cpu > gpurun_out/configs/c3.json 2> gpurun_out/configs/c3.err || exit 1
python - <<PY
import json
for c in ("c2","c3"):
    d=json.load(open(f"gpurun_out/configs/{c}.json")); print(c, d["config"]["workload"], round(d["value"]), "Mrays/s", round(d["ms_per_step"],1), "ms/frame", round(d["rays_per_primary"],2), "rays/primary", {k:round(v,1) for k,v in d["kernel_ms"].items()})
PY
