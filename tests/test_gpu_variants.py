"""Edge cases of the trace loop the shipped scenes do not reach, GPU (through the C ABI) against the pinned CPU oracle
on the same flattened arrays: no light at all, the maximum of 16 lights, bounce depths 4 and 5 (queue fan-out cap),
a scene of analytic primitives only, a scene of triangles only, image sizes that are not multiples of the 16x16 tile.
(Synthetic variations of the reference's scenes: no reference output exists for them, the oracle is the checker.)"""
import copy

import numpy as np
import pytest

from conftest import get_scene, image_metrics

pytestmark = pytest.mark.gpu


def check(scene, w, h, spp=4, depth=3, label=""):
    from oracle.binding import Oracle
    from raytracer_amd.renderer import Renderer
    r = Renderer(0).set_scene(scene)
    try:
        gpu = r.render(w, h, spp=spp, max_depth=depth)
    finally:
        r.close()
    cpu = Oracle(scene).render(w, h, spp=spp, max_depth=depth)
    ids = (gpu["primid"] == cpu["primid"]).mean()
    frac, psnr = image_metrics(gpu["rgb8"], cpu["rgb8"])
    st = gpu["stats"]
    print(f"{label}: primid {ids:.6f} rgb8<=1 {frac:.6f} psnr {psnr:.2f} dB, rays {st['rays_closest']} + {st['rays_shadow']}")
    assert st["queue_overflow"] == 0
    assert st["rays_closest"] == cpu["stats"]["rays_closest"] and st["rays_shadow"] + st["shadow_zero"] == cpu["stats"]["rays_shadow"]
    assert ids >= 0.9999 and frac >= 0.999 and psnr >= 50.0
    return gpu, cpu


def test_no_lights():
    s = copy.copy(get_scene("sample", False))
    s.light_pos = np.zeros((0, 3), np.float32)
    s.light_color = np.zeros((0, 3), np.float32)
    gpu, cpu = check(s, 150, 101, label="no lights")
    assert gpu["stats"]["rays_shadow"] == 0


def test_sixteen_lights():
    s = copy.copy(get_scene("sample", False))
    rng = np.random.default_rng(7)
    s.light_pos = np.concatenate([s.light_pos, rng.uniform([-8, -2, 0], [8, 20, 16], (16 - len(s.light_pos), 3)).astype(np.float32)])
    s.light_color = np.concatenate([s.light_color, rng.uniform(2, 12, (16 - len(s.light_color), 3)).astype(np.float32)])
    assert len(s.light_pos) == 16
    check(s, 131, 77, label="16 lights")


@pytest.mark.parametrize("depth", [4, 5, 15])
def test_deeper_bounces(depth):
    check(get_scene("sample", False), 97, 61, depth=depth, label=f"depth {depth}")


def test_depth_eight_on_the_refractive_scene():
    """bounce depth 8 on the competition scene (ice, crystal and water refract and reflect at once): the levels past
    the queue fan-out cap of 16 rays per sample still fit, because capacity is per chunk, not per sample"""
    check(get_scene("competition", False), 64, 48, depth=8, label="competition depth 8")


def test_analytic_primitives_only():
    s = copy.copy(get_scene("sample", False))
    z3, z2 = np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32)
    s.v0 = s.v1 = s.v2 = s.nA = s.nB = s.nC = z3
    s.tA = s.tB = s.tC = z2
    s.tri_flags = np.zeros(0, np.uint32)
    s.tri_object = np.zeros(0, np.int32)
    gpu, _ = check(s, 160, 90, label="spheres + planes only")
    assert gpu["stats"]["n_triangles"] == 0


def test_triangles_only():
    s = copy.copy(get_scene("bunny", False))
    s.pl_point = s.pl_normal = np.zeros((0, 3), np.float32)
    s.pl_object = np.zeros(0, np.int32)
    s.sph_M = s.sph_Minv = s.sph_N = np.zeros((0, 16), np.float32)
    s.sph_object = np.zeros(0, np.int32)
    check(s, 113, 67, label="bunny alone")


def test_one_pixel_and_one_row():
    s = get_scene("sample", False)
    check(s, 1, 1, label="1x1")
    check(s, 33, 1, spp=16, label="33x1 @16spp")
