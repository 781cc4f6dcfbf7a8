# traversal kernels compiled for 7 / 6 blocks per CU (72 / 80 VGPRs) against the 64-VGPR build, each at its own grid
cp raytracer_amd/csrc/libraytracer_hip.so /tmp/base.so
for rep in 1 2; do
for v in "base 8" "mb7 7" "mb6 6"; do
  set -- $v
  [ $1 = base ] && cp /tmp/base.so raytracer_amd/csrc/libraytracer_hip.so || cp build_variants/$1.so raytracer_amd/csrc/libraytracer_hip.so
  RT_TRACE_BLOCKS=$2 timeout -k 10 200 python bench.py --steps 3 --warmup 2 --no-cpu --no-roofline > /tmp/m.json 2>/tmp/m.err || { tail -3 /tmp/m.err; cp /tmp/base.so raytracer_amd/csrc/libraytracer_hip.so; exit 1; }
  python3 -c "
import json; d=json.load(open('/tmp/m.json')); print('$1 blocks $2:', round(d['ms_per_step'],1), 'ms')"
done
done
cp /tmp/base.so raytracer_amd/csrc/libraytracer_hip.so
