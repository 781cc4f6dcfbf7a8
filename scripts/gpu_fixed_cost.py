import sys, time
sys.path.insert(0, "/root/repo")
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
import torch
r = Renderer(0).set_scene(FlatScene.builtin("competition"))
dev = torch.device("cuda", 0)
from raytracer_amd.dist import DistributedFrame
for (w, h, spp) in ((320, 208, 4), (64, 48, 4)):
    df = DistributedFrame(r, w, h, 0, 1, dev, None)
    f = lambda: r.render_tiles_device(w, h, 0, 1, df.t8.data_ptr(), 0, df.tp.data_ptr(), spp=spp)
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(50): st = f()
    dt = (time.perf_counter() - t0) / 50 * 1e3
    print(f"{w}x{h}@{spp}: {dt:.2f} ms wall per frame, GPU {st['ms_total']:.2f} ms")
