"""C5 frame time against the number of chunks (workspace) and one / two pipelines side by side (RT_LANES)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
W, H, SPP = 7680, 4320, int(sys.argv[1]) if len(sys.argv) > 1 else 64
r = Renderer(0).set_scene(FlatScene.builtin("stress:733"))
n = r.tile_counts(W, H, 0, 1)[0]
t8 = torch.zeros(n, 256, 3, dtype=torch.uint8, device="cuda:0")
tp = torch.zeros(n, 256, dtype=torch.int32, device="cuda:0")
r.render_tiles_device(W // 4, H // 4, 0, 1, t8.data_ptr(), 0, tp.data_ptr(), spp=SPP)  # sizing frame
for gb, lanes in ((30, 1), (30, 2), (60, 1), (60, 2), (120, 1), (120, 2)):
    os.environ["RT_WORKSPACE_GB"] = str(gb)
    os.environ["RT_LANES"] = str(lanes)
    for k in range(2):
        st = r.render_tiles_device(W, H, 0, 1, t8.data_ptr(), 0, tp.data_ptr(), spp=SPP)
    print(f"workspace {gb} GB, lanes {lanes}: {st['ms_total']:.0f} ms, {st['chunks']} chunks", flush=True)
