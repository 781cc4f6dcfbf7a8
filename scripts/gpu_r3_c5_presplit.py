"""pre-splitting on the 50 M-triangle scene: commit time against frame time (7680x4320 @64 spp)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
W, H, SPP = 7680, 4320, 64
s = FlatScene.builtin("stress:733")
for budget in ("0", "0.25", "1.0"):
    os.environ["RT_PRESPLIT_MAX_TRIS"] = "1000000000"
    os.environ["RT_PRESPLIT_BUDGET"] = budget
    os.environ["RT_VERBOSE"] = "1"
    t0 = time.time()
    r = Renderer(0).set_scene(s)
    cs = r.commit_stats()
    os.environ.pop("RT_VERBOSE")
    n = r.tile_counts(W, H, 0, 1)[0]
    t8 = torch.zeros(n, 256, 3, dtype=torch.uint8, device="cuda:0")
    tp = torch.zeros(n, 256, dtype=torch.int32, device="cuda:0")
    r.render_tiles_device(W // 4, H // 4, 0, 1, t8.data_ptr(), 0, tp.data_ptr(), spp=SPP)
    for k in range(2):
        st = r.render_tiles_device(W, H, 0, 1, t8.data_ptr(), 0, tp.data_ptr(), spp=SPP)
    print(f"budget {budget}: commit {(cs['ms_bvh_build'] + cs['ms_reference_order'] + cs['ms_records'] + cs['ms_upload']) / 1e3:.2f} s (BVH {cs['ms_bvh_build'] / 1e3:.2f}), "
          f"{cs['bvh_nodes']} nodes; frame {st['ms_total']:.0f} ms, {st['chunks']} chunks", flush=True)
    r.close()
    del t8, tp
