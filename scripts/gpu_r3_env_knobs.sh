# environment knobs on the headline frame, one after the other on one box (no rebuild)
run() { env "$@" timeout -k 10 200 python bench.py --steps 3 --warmup 2 --no-cpu --no-roofline > /tmp/k.json 2>/tmp/k.err || { tail -3 /tmp/k.err; exit 1; }
  python3 -c "
import json; d=json.load(open('/tmp/k.json')); print('$*', round(d['ms_per_step'],1), 'ms', d['config']['chunks_per_frame'], 'chunks')"; }
run RT_NOOP=1
run RT_ANY_ORDERED=1
run RT_OCCLUDER_CACHE=0
run RT_SPLIT_BUDGET=0
run RT_SPLIT_BUDGET=64
run RT_SPLIT_BUDGET=256
run RT_L0_STORE=1
run RT_PRESPLIT_MIN=0.35
run RT_PRESPLIT_MIN=0.7
run RT_NOOP=2
