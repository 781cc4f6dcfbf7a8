"""exploration for tests/test_gpu_c5.py: timings and counters of stress:733 at a reduced frame"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 733
t0 = time.time(); scene = FlatScene.builtin(f"stress:{rep}"); t1 = time.time()
print(f"flatten {scene.n_triangles} triangles: {t1 - t0:.1f} s", flush=True)
r = Renderer(0).set_scene(scene); t2 = time.time()
print(f"commit {t2 - t1:.1f} s", r.commit_stats(), flush=True)
for (w, h, spp, chunk) in ((960, 540, 4, 0), (960, 540, 4, 1 << 18), (1920, 1080, 4, 0)):
    t = time.time()
    o = r.render(w, h, spp=spp, chunk_samples=chunk, collect_stats=1, want=("rgb8", "primid"))
    st = o["stats"]
    print(w, h, spp, chunk, f"{time.time() - t:.2f} s", {k: st[k] for k in ("ms_total", "rays_closest", "rays_shadow", "queue_overflow", "bvh_max_stack", "bvh_depth", "split_items", "max_walk_any", "max_walk_closest")}, flush=True)
