# workspace sizes under two libraries (build_variants/head.so = before, the tree's = after), interleaved
cp raytracer_amd/csrc/libraytracer_hip.so /tmp/new.so
for gb in 4 8 16 30; do
  for so in build_variants/head.so /tmp/new.so; do
    cp $so raytracer_amd/csrc/libraytracer_hip.so
    RT_WORKSPACE_GB=$gb timeout -k 10 200 python bench.py --steps 3 --warmup 2 --no-cpu --no-roofline > /tmp/ws.json 2>/tmp/ws.err || { tail -3 /tmp/ws.err; exit 1; }
    python3 -c "
import json; d=json.load(open('/tmp/ws.json')); print('$gb GB', '$so'.split('/')[-1], round(d['ms_per_step'],1), 'ms', d['config']['chunks_per_frame'], 'chunks')"
  done
done
cp /tmp/new.so raytracer_amd/csrc/libraytracer_hip.so
