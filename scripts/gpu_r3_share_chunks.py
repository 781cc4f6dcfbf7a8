"""An N = 8 share of the 4K / 64 spp frame (rank 4's tiles, 66 M samples) cut into 1, 2, 4, 8, 16 chunks: does the pipeline's
level mix beat the single chunk the planner picks (the fewest that fit)?  Caller-given chunk sizes get worst-case queues."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
from raytracer_amd.dist import DistributedFrame
W, H, SPP = 3840, 2160, 64
r = Renderer(0).set_scene(FlatScene.builtin("competition"))
dev = torch.device("cuda", 0)
n, rank = 8, 4
df = DistributedFrame(r, W, H, rank, n, dev, None)
df.world = 1
samples = (W // 16) * ((H + 15) // 16) * 256 * SPP // n
for chunks in (0, 1, 2, 4, 8, 16):
    cs = 0 if chunks == 0 else samples // chunks + 256 * SPP
    def frame():
        return r.render_tiles_device(W, H, rank, n, df.t8.data_ptr(), 0, df.tp.data_ptr(), spp=SPP, chunk_samples=cs)
    frame(); frame(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        st = frame()
    torch.cuda.synchronize()
    print(f"chunks asked {chunks or 'library'}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms, {st['chunks']} chunks", flush=True)
