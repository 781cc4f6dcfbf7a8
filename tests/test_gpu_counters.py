"""the in-kernel counters the roofline's flops rest on (rt_stats.nodes_visited / tris_tested) against a CPU
restatement of the kernel's walk (oracle: ro_walk_counts) over the SAME committed tree (rt_get_bvh / rt_get_bvh_info) and the
same rays.

How many nodes a ray visits in the per-lane kernel depends on which phase its wave votes for (a node step taken before a
pending leaf's triangles sees the older, larger best distance), so the identity is stated where it is a function of the ray
alone: every ray is traced 256 times in a row -- a closest-hit batch, whichever order it is consumed in, then holds ONE ray, all
lanes of a wave walk in lock step, and the vote is "triangle step whenever a leaf is pending" (node phase iff nNode >= 2 nTri) -- with the packet kernel off
(RT_PACKET_LEVELS=0: primary rays walk per lane like every other ray).  Scene: the competition meshes alone (no analytic
primitives: their hits would set the initial best distance)."""
import copy

import numpy as np
import pytest

from conftest import get_scene, load_golden

pytestmark = pytest.mark.gpu


def _meshes_only(scene):
    s = copy.copy(scene)
    z16, z3 = np.zeros((0, 16), np.float32), np.zeros((0, 3), np.float32)
    s.sph_M = s.sph_Minv = s.sph_N = z16
    s.cone_M = s.cone_Minv = s.cone_N = z16
    s.sph_object = s.cone_object = s.pl_object = np.zeros(0, np.int32)
    s.pl_point = s.pl_normal = z3
    s.name = scene.name + " (meshes only)"
    return s


def test_node_and_triangle_counters_equal_the_cpu_walk(monkeypatch):
    from oracle.binding import Oracle
    from raytracer_amd.renderer import Renderer
    monkeypatch.setenv("RT_PACKET_LEVELS", "0")
    scene = _meshes_only(get_scene("competition", False))
    r = Renderer(0).set_scene(scene)
    nodes, order = r.get_bvh()
    info = r.get_bvh_info()
    # (slots >= triangles: large triangles are cut into pieces before the build and own one slot per piece)
    assert len(order) >= scene.n_triangles and set(order.tolist()) == set(range(scene.n_triangles))
    assert info["root_ref"] == 0 and info["bound"] > 0
    oracle = Oracle(scene)
    g = load_golden("rays_competition")
    rays = np.ascontiguousarray(g["rays"][:1536], np.float32)       # primary + bounce + shadow-like rays of the fixture
    cn, ct, slot = oracle.walk_counts(nodes, order, info, rays)
    # the CPU walk over the kernel's tree finds the reference's closest triangle (the oracle's own closest() over its per-mesh trees)
    _, ohits, _, _ = oracle.trace_batch(rays, depth=0)
    hit = slot >= 0
    assert np.array_equal(hit, ohits["hit"] == 1)
    first = scene.mesh_first_triangle()
    tri_global = np.array([first[o] + t if h else -1 for o, t, h in zip(ohits["object"], ohits["triangle"], hit)])
    assert np.array_equal(order[slot[hit]].astype(np.int64), tri_global[hit])
    REP = 256
    total_n = total_t = 0
    for lo in range(0, len(rays), 256):                                # per group of 256 rays, so that a difference is localised
        part = rays[lo:lo + 256]
        _, hits, _, st = r.trace_batch(np.repeat(part, REP, axis=0), depth=0, want_shadow=False)
        en, et = int(cn[lo:lo + 256].sum()) * REP, int(ct[lo:lo + 256].sum()) * REP
        assert st["rays_closest"] == len(part) * REP
        assert (st["nodes_visited"], st["tris_tested"]) == (en, et), f"rays {lo}..{lo + len(part)}: kernel {st['nodes_visited']} / {st['tris_tested']}, CPU walk {en} / {et}"
        total_n += en
        total_t += et
        assert np.array_equal(hits["hit"][::REP] == 1, hit[lo:lo + 256])
    print(f"counter identity on {len(rays)} rays x {REP}: {total_n // REP} node steps, {total_t // REP} triangle tests "
          f"({total_n / REP / len(rays):.2f} + {total_t / REP / len(rays):.2f} per ray), tree of {len(nodes)} nodes")
    r.close()
