"""RT_TAILPROF: when the waves of each launch start and end -- N = 8 share of the 4K frame in 1 and 8 chunks, and the whole frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
from raytracer_amd.dist import DistributedFrame
W, H, SPP = 3840, 2160, 64
r = Renderer(0).set_scene(FlatScene.builtin("competition"))
dev = torch.device("cuda", 0)
for N, nc in ((8, 1), (8, 8), (1, 0)):
    df = DistributedFrame(r, W, H, 0, N, dev, None)
    df.world = 1
    total = r.tile_counts(W, H, 0, N)[0] * 256 * SPP
    chunk = 0 if nc == 0 else (total + nc - 1) // nc
    for k in range(3):
        if k == 2:
            os.environ["RT_TAILPROF"] = "1"
            print(f"== N={N} chunks asked {nc}", file=sys.stderr, flush=True)
        st = r.render_tiles_device(W, H, 0, N, df.t8.data_ptr(), 0, df.tp.data_ptr(), spp=SPP, chunk_samples=chunk)
        os.environ.pop("RT_TAILPROF", None)
    print(f"   {st['chunks']} chunks, {st['ms_total']:.1f} ms", file=sys.stderr, flush=True)
