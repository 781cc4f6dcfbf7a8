# bench.py (no CPU leg) on every library variant in build_variants/ (on the GPU box: the tree is a scratch copy), REPS times
# each, interleaved, so that box-to-box and run-to-run differences do not pass for differences between the variants
mkdir -p gpurun_out/variants
cp raytracer_amd/csrc/libraytracer_hip.so /tmp/base.so
for rep in $(seq 1 ${REPS:-2}); do
for so in /tmp/base.so build_variants/*.so; do
  name=$(basename $so .so)
  cp $so raytracer_amd/csrc/libraytracer_hip.so
  timeout -k 10 200 python bench.py --steps 3 --warmup 2 --no-cpu ${BENCH_ARGS:-} > gpurun_out/variants/$name.json 2> gpurun_out/variants/$name.err || { echo $name failed; tail -3 gpurun_out/variants/$name.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/variants/$name.json"))
print("$name", round(d["ms_per_step"],1), "ms", round(d["value"]), "Mrays/s", {k:round(v,1) for k,v in d.get("kernel_ms",{}).items() if v > 0.5}, flush=True)
PY
done
done
cp /tmp/base.so raytracer_amd/csrc/libraytracer_hip.so
