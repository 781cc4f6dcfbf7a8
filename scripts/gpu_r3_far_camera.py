"""GPU vs oracle for the competition scene seen from `distance` away (tests/test_gpu_cameras.py: far camera): which kinds of
primitives the two disagree on"""
import copy, math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import get_scene
from raytracer_amd import _native as N
from raytracer_amd.renderer import Renderer
from oracle.binding import Oracle
distance = float(sys.argv[1])
scene = get_scene("competition", False)
base = scene.camera
lo = np.minimum(np.minimum(scene.v0.min(0), scene.v1.min(0)), scene.v2.min(0))
hi = np.maximum(np.maximum(scene.v0.max(0), scene.v1.max(0)), scene.v2.max(0))
centre = 0.5 * (lo + hi)
cam = copy.copy(base)
for k in range(3):
    cam.origin[k] = float(centre[k])
cam.origin[2] = float(centre[2] - math.copysign(distance, base.dz))
for k, v in enumerate(np.eye(4, dtype=np.float32).reshape(-1)):
    cam.rot[k] = float(v)
cam.fov_deg = float(2.0 * math.degrees(math.atan(0.4 * float((hi - lo)[:2].max()) / distance)))
r = Renderer(0).set_scene(scene)
for depth in (0, 3):
    gpu = r.render(96, 72, spp=4, camera=cam, max_depth=depth)
    cpu = Oracle(scene).render(96, 72, spp=4, camera=cam, max_depth=depth)
    g, c = gpu["primid"], cpu["primid"]
    kind = lambda p: np.where(p == N.PRIM_MISS, 0, np.where(p >= N.PRIM_ANALYTIC, 1, 2))
    names = ["miss", "analytic", "triangle"]
    print(f"distance {distance:g} depth {depth}: ids agree {(g == c).mean():.4f}; rays {gpu['stats']['rays_closest']} / {cpu['stats']['rays_closest']}")
    for a in range(3):
        for b in range(3):
            n = int(((kind(g) == a) & (kind(c) == b)).sum())
            if n:
                same = int(((kind(g) == a) & (kind(c) == b) & (g == c)).sum())
                print(f"   gpu {names[a]:9s} oracle {names[b]:9s}: {n} pixels, {same} with the same id")
