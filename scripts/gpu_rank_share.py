"""What one rank of an N-GPU frame costs on this GPU: rank 0's interleaved tiles of the 4K/64spp competition frame,
timed alone (no gather).  frame_ms(N=1) / (N * share_ms(N)) is the scaling efficiency the tile partition allows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
from raytracer_amd.dist import DistributedFrame
W, H, SPP = 3840, 2160, 64
r = Renderer(0).set_scene(FlatScene.builtin("competition"))
dev = torch.device("cuda", 0)
base = None
for n in (1, 2, 4, 8):
    worst = 0.0
    for rank in sorted({0, n // 2, n - 1}):
        df = DistributedFrame(r, W, H, rank, n, dev, None)
        df.world = 1  # no collective: the share alone
        def frame():
            return r.render_tiles_device(W, H, rank, n, df.t8.data_ptr(), 0, df.tp.data_ptr(), spp=SPP)
        frame(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            st = frame()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 2 * 1e3
        worst = max(worst, ms)
        print(f"N={n} rank {rank}: {ms:.1f} ms, {st['rays_closest'] + st['rays_shadow']} rays", flush=True)
    base = base or worst
    print(f"N={n}: slowest share {worst:.1f} ms -> efficiency bound {base / (n * worst):.3f}", flush=True)
