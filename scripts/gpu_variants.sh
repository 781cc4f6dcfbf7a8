# bench.py (no CPU leg) on every library variant in build_variants/ (on the GPU box: the tree is a scratch copy)
mkdir -p gpurun_out/variants
cp raytracer_amd/csrc/libraytracer_hip.so /tmp/base.so
for so in /tmp/base.so build_variants/*.so; do
  name=$(basename $so .so)
  cp $so raytracer_amd/csrc/libraytracer_hip.so
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/variants/$name.json 2> gpurun_out/variants/$name.err || { echo $name failed; tail -3 gpurun_out/variants/$name.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/variants/$name.json"))
l=d["roofline"].get("lanes",{})
print("$name", round(d["ms_per_step"],1), "ms", round(d["value"]), "Mrays/s", {k:round(v,1) for k,v in d["kernel_ms"].items() if k in ("ms_packet","ms_closest","ms_shade","ms_any")}, "lanes %.1f/%.1f" % (l.get("lanes_per_node_phase",0), l.get("lanes_per_triangle_phase",0)), flush=True)
PY
done
cp /tmp/base.so raytracer_amd/csrc/libraytracer_hip.so
