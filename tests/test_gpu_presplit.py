"""Pre-splitting (rt_bvh.cpp: presplit, on by default up to 4 Mi triangles) changes the tree, not the image: a triangle that
owns several slots is the same triangle with the same exact test in each of them, the closest hit and its tie rule do not
depend on the order of the walk, and the accumulators are fixed point.  RT_PRESPLIT is read by rt_commit.

One class of hits WAS a property of the tree ("hits that only a tree decides"): a ray almost parallel to a
triangle's plane gets a plane distance of millions of scene sizes, at that distance `cross(B - P, C - P)` cancels to exactly
zero and the reference's inside test accepts -- if the walk ever tests that triangle.  One such ray exists among the 69 M of
the full-depth frame below (pixel (634, 430), sample 9's second bounce: t = 4.29e6, found with scripts/gpu_r3_pixel_rays.py;
the tree without pre-splitting found it, the pre-split tree did not).  Since the kernels reject triangle hits beyond the
farthest distance the scene allows (rtFarLimit) neither tree reports it, and the frames are bit-identical at every depth."""
import numpy as np
import pytest

from conftest import get_scene

pytestmark = pytest.mark.gpu


def render_both(name, monkeypatch, **kw):
    from raytracer_amd.renderer import Renderer
    scene = get_scene(name, False)
    out, slots = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RT_PRESPLIT", mode)
        r = Renderer(0).set_scene(scene)
        slots[mode] = len(r.get_bvh()[1])
        out[mode] = r.render(1280, 720, collect_stats=1, want=("rgb8", "rgbf", "primid"), **kw)
        r.close()
    assert slots["0"] == scene.n_triangles and slots["1"] >= slots["0"]
    if name == "competition":
        assert slots["1"] > slots["0"]          # (its floor and walls are two triangles each)
    s0, s1 = out["0"]["stats"], out["1"]["stats"]
    print(f"{name} {kw}: slots {slots['0']} -> {slots['1']}, node steps {s0['nodes_visited']} -> {s1['nodes_visited']}, "
          f"triangle tests {s0['tris_tested']} -> {s1['tris_tested']}")
    return out["0"], out["1"]


COUNTS = ("rays_closest", "rays_shadow", "shadow_zero", "shadow_unoccluded", "hits_shaded")


@pytest.mark.parametrize("name,kw", [("competition", dict(spp=16, max_depth=1)), ("bunny", dict(spp=64, max_depth=3)),
                                     ("sample", dict(spp=64, max_depth=3))])
def test_presplit_same_bits(name, kw, monkeypatch):
    a, b = render_both(name, monkeypatch, **kw)
    for k in ("rgb8", "primid"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["rgbf"].view(np.uint32), b["rgbf"].view(np.uint32))
    for k in COUNTS:
        assert a["stats"][k] == b["stats"][k], k


def test_presplit_full_depth_same_bits(monkeypatch):
    a, b = render_both("competition", monkeypatch, spp=16, max_depth=3)
    for k in ("rgb8", "primid"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["rgbf"].view(np.uint32), b["rgbf"].view(np.uint32))
    for k in COUNTS:
        assert a["stats"][k] == b["stats"][k], k
