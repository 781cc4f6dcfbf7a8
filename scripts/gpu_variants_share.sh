# every library variant in build_variants/: the 4K frame (bench.py) and the N = 8 share in one chunk, REPS times, interleaved
mkdir -p gpurun_out/variants
cp raytracer_amd/csrc/libraytracer_hip.so /tmp/base.so
cat > /tmp/share.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
from raytracer_amd.dist import DistributedFrame
W, H, SPP = 3840, 2160, 64
r = Renderer(0).set_scene(FlatScene.builtin("competition"))
dev = torch.device("cuda", 0)
out = []
for N in (8, 1):
    df = DistributedFrame(r, W, H, 0, N, dev, None); df.world = 1
    f = lambda: r.render_tiles_device(W, H, 0, N, df.t8.data_ptr(), 0, df.tp.data_ptr(), spp=SPP)
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): st = f()
    torch.cuda.synchronize(); out.append(f"N={N}: {(time.perf_counter()-t0)/3*1e3:.1f} ms ({st['chunks']} chunks)")
print("  ".join(out))
PY
for rep in $(seq 1 ${REPS:-2}); do
for so in /tmp/base.so build_variants/*.so; do
  name=$(basename $so .so)
  cp $so raytracer_amd/csrc/libraytracer_hip.so
  echo "$name $(timeout -k 10 200 python /tmp/share.py 2>&1 | tail -1)"
done
done
cp /tmp/base.so raytracer_amd/csrc/libraytracer_hip.so
