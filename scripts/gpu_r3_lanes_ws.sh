# competition 4K frame: one / two pipelines at several workspace sizes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3lanes_ws
for gb in 8 16 30; do for lanes in 1 2; do
  RT_WORKSPACE_GB=$gb RT_LANES=$lanes timeout -k 10 200 python bench.py --steps 3 --warmup 2 --no-cpu --no-roofline > gpurun_out/r3lanes_ws/b_${gb}_$lanes.json 2> gpurun_out/r3lanes_ws/b_${gb}_$lanes.err || { tail -5 gpurun_out/r3lanes_ws/b_${gb}_$lanes.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r3lanes_ws/b_${gb}_$lanes.json"))
print("workspace $gb GB lanes $lanes:", round(d["ms_per_step"],1), "ms", d["config"]["chunks_per_frame"], "chunks", flush=True)
PY
done; done
