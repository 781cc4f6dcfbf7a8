"""Are the FLOAT images (not only their 8-bit truncation) identical across runs, chunk sizes and partitions?
The work fetch is dynamic, so a lane's history -- and with it the occluder cache's content -- differs from run to run:
any dependence of a shadow query's answer on that history would show here.  python scripts/gpu_float_determinism.py [scene W H SPP]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
scene = sys.argv[1] if len(sys.argv) > 1 else "competition"
W, H, SPP = (int(x) for x in (sys.argv[2:5] if len(sys.argv) > 4 else (3840, 2160, 64)))
r = Renderer(0).set_scene(FlatScene.builtin(scene))


def run(rank=0, world=1, **kw):
    n = r.tile_counts(W, H, rank, world)[0]
    tf = torch.zeros(n, 256, 3, dtype=torch.float32, device="cuda:0")
    r.render_tiles_device(W, H, rank, world, 0, tf.data_ptr(), 0, spp=SPP, **kw)
    torch.cuda.synchronize()
    return tf


a = run()
bad = 0
for name, b in (("same configuration again", run()), ("chunks of 3 Mi samples", run(chunk_samples=3 << 20)),
                ("rank 0 of 2", run(0, 2)), ("rank 1 of 2", run(1, 2))):
    ref = a if "rank" not in name else a[int(name[5])::2]
    diff = (ref.view(torch.int32) != b.view(torch.int32)).any(dim=-1).sum().item()
    print(f"{name}: {diff} of {ref.shape[0] * 256} pixels differ in their float bits", flush=True)
    bad += diff
sys.exit(1 if bad else 0)
