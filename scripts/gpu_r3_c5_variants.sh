# the C5 diagonal camera (one lane, 4K @16) under library variants (build_variants/*.so), plus the tree's own
cp raytracer_amd/csrc/libraytracer_hip.so /tmp/base.so
for so in /tmp/base.so build_variants/*.so; do
  cp $so raytracer_amd/csrc/libraytracer_hip.so
  for cam in stress-diagonal stress; do
    timeout -k 10 300 python3 scripts/c5_frame.py $cam:733 3840 2160 16 > /tmp/c5v.json 2>/tmp/c5v.err || { tail -3 /tmp/c5v.err; cp /tmp/base.so raytracer_amd/csrc/libraytracer_hip.so; exit 1; }
    python3 -c "
import json; d=json.load(open('/tmp/c5v.json')); print('$(basename $so .so)', '$cam', round(d['ms_total'],1), 'ms, trace', round(d['ms_trace'],1), 'shade', round(d['ms_shade'],1), d['chunks'], 'chunks')"
  done
done
cp /tmp/base.so raytracer_amd/csrc/libraytracer_hip.so
