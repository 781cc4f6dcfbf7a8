# bench.py (with the in-kernel counters) under several settings of ONE environment variable: VAR=name VALUES="a b c"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3env
for v in $VALUES; do
  env $VAR=$v timeout -k 10 200 python bench.py --steps 2 --warmup 2 --no-cpu $BENCH_ARGS > gpurun_out/r3env/b_$v.json 2> gpurun_out/r3env/b_$v.err || { tail -5 gpurun_out/r3env/b_$v.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r3env/b_$v.json"))
r=d.get("roofline",{})
print("$VAR=$v:", round(d["ms_per_step"],1), "ms", {k: round(x,1) for k,x in d.get("kernel_ms",{}).items() if x > 0.5}, "records/ray", r.get("records_per_ray"), (r.get("lanes") or {}).get("node_phases"), (r.get("lanes") or {}).get("triangle_phases"), flush=True)
PY
done
