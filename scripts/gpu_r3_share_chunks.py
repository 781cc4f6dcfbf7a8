"""N = 8 share of the 4K / 64 spp frame (rank 0's tiles) against the number of chunks it is cut into."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
from raytracer_amd.dist import DistributedFrame
W, H, SPP, N = 3840, 2160, 64, 8
r = Renderer(0).set_scene(FlatScene.builtin("competition"))
dev = torch.device("cuda", 0)
df = DistributedFrame(r, W, H, 0, N, dev, None)
df.world = 1
total = r.tile_counts(W, H, 0, N)[0] * 256 * SPP
for nc in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24):
    chunk = 0 if nc == 0 else (total + nc - 1) // nc
    def frame():
        return r.render_tiles_device(W, H, 0, N, df.t8.data_ptr(), 0, df.tp.data_ptr(), spp=SPP, chunk_samples=chunk)
    try:
        frame(); frame(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            st = frame()
        torch.cuda.synchronize()
        print(f"chunks asked {nc} (0 = library default): {st['chunks']} chunks, {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms", flush=True)
    except Exception as e:
        print(f"chunks asked {nc}: {e}", flush=True)
