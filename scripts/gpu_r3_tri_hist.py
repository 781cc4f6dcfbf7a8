"""which triangles the closest-hit walks test most (CPU walk of the committed tree over the fixture rays + camera rays)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
from oracle.binding import Oracle
scene = FlatScene.builtin("competition")
r = Renderer(0).set_scene(scene)
nodes, order = r.get_bvh(); info = r.get_bvh_info()
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "rays_competition.npz"))
rays = np.ascontiguousarray(g["rays"], np.float32)
tests = np.zeros(len(order), np.uint32)
o = Oracle(scene)
cn, ct, slot = o.walk_counts(nodes, order, info, rays, tests)
print(f"{len(rays)} rays: {cn.mean():.2f} node steps, {ct.mean():.2f} triangle tests per ray (no analytic primitives in this walk)")
v0, v1, v2 = scene.v0[order], scene.v1[order], scene.v2[order]
area = 0.5 * np.linalg.norm(np.cross(v1 - v0, v2 - v0), axis=1)
ext = np.maximum(np.maximum(v0, v1), v2) - np.minimum(np.minimum(v0, v1), v2)
box_area = 2 * (ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0])
idx = np.argsort(-tests.astype(np.int64))
tot = tests.sum()
cum = np.cumsum(tests[idx])
for k in (10, 100, 1000, 10000):
    print(f"top {k} slots: {cum[k - 1] / tot:.3f} of all tests")
print("top 15:", [(int(order[i]), int(tests[i]), round(float(area[i]), 3), round(float(box_area[i]), 3), int(scene.tri_object[order[i]])) for i in idx[:15]])
big = np.argsort(-box_area)[:20]
print("largest boxes:", [(int(order[i]), int(tests[i]), round(float(box_area[i]), 2)) for i in big])
print("scene extent", scene.v0.min(axis=0), scene.v0.max(axis=0))
