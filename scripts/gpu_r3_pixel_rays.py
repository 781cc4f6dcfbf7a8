"""The closest-hit rays of ONE pixel, from the oracle (RO_ONLY_PIXEL + ro_set_ray_log), traced one by one through rt_trace_batch
with and without pre-splitting: which ray's hit differs, and what the reference's own walk (the oracle) says about it.
usage: gpu_r3_pixel_rays.py i j spp [cpu]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
i, j, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
log = os.path.join(ROOT, "gpurun_out", f"pixel_{i}_{j}.rays")
os.makedirs(os.path.dirname(log), exist_ok=True)
os.environ["RO_ONLY_PIXEL"] = f"{i},{j}"
from conftest import get_scene  # noqa: E402
from oracle.binding import Oracle, read_ray_log, set_ray_log  # noqa: E402

scene = get_scene("competition", False)
o = Oracle(scene)
set_ray_log(log)
o.render(1280, 720, spp=spp, max_depth=3, threads=1)
set_ray_log(None)
rays, ohit, _, otri, odist = read_ray_log(log)
print(f"pixel ({i},{j}): {len(rays)} closest-hit rays, {ohit.sum()} hits")
if len(sys.argv) > 4 and sys.argv[4] == "cpu":
    sys.exit(0)
from raytracer_amd.renderer import Renderer  # noqa: E402
res = {}
for mode in ("0", "1"):
    os.environ["RT_PRESPLIT"] = mode
    r = Renderer(0).set_scene(scene)
    _, hits, _, _ = r.trace_batch(rays, depth=0, want_shadow=False)
    res[mode] = hits
    r.close()
first = scene.mesh_first_triangle()
for k in range(len(rays)):
    h0, h1 = res["0"]["hit"][k], res["1"]["hit"][k]
    t0, t1 = res["0"]["triangle"][k], res["1"]["triangle"][k]
    if h0 != h1 or t0 != t1 or h0 != ohit[k]:
        print(f"ray {k}: o {rays[k, :3]} d {rays[k, 3:]}")
        print(f"   oracle hit {ohit[k]} tri {otri[k]} dist {odist[k]!r}")
        for m in ("0", "1"):
            print(f"   presplit {m}: hit {res[m]['hit'][k]} object {res[m]['object'][k]} triangle {res[m]['triangle'][k]} "
                  f"dist {res[m]['distance'][k]!r} P {res[m]['intersection'][k]}")
        ob, tr = res["0"]["object"][k], res["0"]["triangle"][k]
        if h0 and tr >= 0:
            g = first[ob] + tr
            print(f"   triangle {g}: A {scene.v0[g]} B {scene.v1[g]} C {scene.v2[g]}")
print("done")
