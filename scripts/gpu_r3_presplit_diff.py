"""Where do the float images with and without pre-splitting differ?  (diagnostic for tests/test_gpu_presplit.py)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import get_scene  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "competition"
w, h, spp = 1280, 720, int(sys.argv[2]) if len(sys.argv) > 2 else 16
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
scene = get_scene(name, False)
out = {}
for mode in ("0", "1"):
    os.environ["RT_PRESPLIT"] = mode
    r = Renderer(0).set_scene(scene)
    out[mode] = r.render(w, h, spp=spp, max_depth=depth, collect_stats=1, want=("rgb8", "rgbf", "primid"))
    r.close()
a, b = out["0"]["rgbf"], out["1"]["rgbf"]
d = np.abs(a - b).max(axis=2)
ys, xs = np.nonzero(d)
print(f"{name} depth {depth}: {len(ys)} of {w * h} pixels differ, max {d.max():.3e}")
for k in ("rays_closest", "rays_shadow", "shadow_zero", "shadow_unoccluded", "hits_shaded"):
    print(k, out["0"]["stats"][k], out["1"]["stats"][k])
for y, x in list(zip(ys, xs))[:20]:
    print(y, x, a[y, x], b[y, x], hex(out["0"]["primid"][y, x]))
