"""Random cameras (seeded): the competition and sample scenes seen from poses the reference never used -- grazing views,
views from inside the geometry, the horizon of the infinite planes (far hits, where float precision is poor and the
conservative slab test has to cover what the exact triangle test accepts) -- GPU against the pinned oracle."""
import copy
import math

import numpy as np
import pytest

from conftest import get_scene, image_metrics
from raytracer_amd import _native as N

pytestmark = pytest.mark.gpu


def camera(base, rng, spread):
    c = N.Camera()
    o = np.array(list(base.origin), np.float32) + rng.uniform(-spread, spread, 3).astype(np.float32)
    yaw, pitch = rng.uniform(-math.pi, math.pi), rng.uniform(-0.6, 0.6)
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    rot = np.array([[cy, 0, sy, 0], [sy * sp, cp, -cy * sp, 0], [-sy * cp, sp, cy * cp, 0], [0, 0, 0, 1]], np.float32)
    for k in range(3):
        c.origin[k] = float(o[k])
    c.dz = base.dz
    c.fov_deg = float(rng.uniform(40, 100))
    for k, v in enumerate(rot.T.reshape(-1)):  # column-major like glm
        c.rot[k] = float(v)
    return c


@pytest.mark.parametrize("name,spread", [("competition", 1.5), ("sample", 4.0)])
def test_random_cameras(name, spread):
    from oracle.binding import Oracle
    from raytracer_amd.renderer import Renderer
    scene = get_scene(name, False)
    rng = np.random.default_rng(20261110)
    r = Renderer(0).set_scene(scene)
    o = Oracle(scene)
    try:
        for k in range(6):
            cam = camera(scene.camera, rng, spread)
            gpu = r.render(96, 72, spp=4, camera=cam)
            cpu = o.render(96, 72, spp=4, camera=cam)
            ids = (gpu["primid"] == cpu["primid"]).mean()
            frac, psnr = image_metrics(gpu["rgb8"], cpu["rgb8"])
            hit = (gpu["primid"] != N.PRIM_MISS).mean()
            print(f"{name} camera {k}: {hit:.2f} of the pixels hit, primid {ids:.6f} rgb8<=1 {frac:.6f} psnr {psnr:.2f} dB")
            assert gpu["stats"]["rays_closest"] == cpu["stats"]["rays_closest"] and gpu["stats"]["rays_shadow"] + gpu["stats"]["shadow_zero"] == cpu["stats"]["rays_shadow"]
            assert ids >= 0.9999 and frac >= 0.999 and psnr >= 50.0
    finally:
        r.close()


@pytest.mark.parametrize("distance", [50.0, 2000.0])
def test_camera_far_outside_the_scene(distance):
    """The competition scene seen through a narrow lens from tens and thousands of scene sizes away: every genuine hit is then
    about `distance` away, far beyond the scene's own extent -- the farthest distance a triangle hit may have (rtFarLimit) grows
    with the ray origin's distance from the scene, and the slab test's error bound with it.  And the reference's unit-sphere
    test, which works in the sphere's object space in floats, cancels from thousands of radii away and accepts every other
    ray (from 2000 units the 0.03 sphere is a blob a third of the frame wide): a frame whose camera stands beyond four times
    the scene's reach traces its primary rays without the spheres' box pre-test (RtQueues::prim_far), so it draws the same
    blob.  GPU against the pinned oracle: ids, ray counts and 8-bit colours (before prim_far: ids 0.9996 at 50 units, 0.44
    at 2000)."""
    from oracle.binding import Oracle
    from raytracer_amd.renderer import Renderer
    scene = get_scene("competition", False)
    base = scene.camera
    lo = np.minimum(np.minimum(scene.v0.min(0), scene.v1.min(0)), scene.v2.min(0))
    hi = np.maximum(np.maximum(scene.v0.max(0), scene.v1.max(0)), scene.v2.max(0))
    centre = 0.5 * (lo + hi)
    cam = copy.copy(base)
    # behind the scene on the z axis, identity rotation, lens narrowed to the scene
    for k in range(3):
        cam.origin[k] = float(centre[k])
    cam.origin[2] = float(centre[2] - math.copysign(distance, base.dz))   # (the lens looks along sign(dz) z)
    cam.dz = base.dz
    for k, v in enumerate(np.eye(4, dtype=np.float32).reshape(-1)):
        cam.rot[k] = float(v)
    cam.fov_deg = float(2.0 * math.degrees(math.atan(0.4 * float((hi - lo)[:2].max()) / distance)))
    r = Renderer(0).set_scene(scene)
    try:
        gpu = r.render(96, 72, spp=4, camera=cam)
        cpu = Oracle(scene).render(96, 72, spp=4, camera=cam)
    finally:
        r.close()
    ids = (gpu["primid"] == cpu["primid"]).mean()
    frac, psnr = image_metrics(gpu["rgb8"], cpu["rgb8"])
    tri_hits = ((gpu["primid"] != N.PRIM_MISS) & (gpu["primid"] < N.PRIM_ANALYTIC)).mean()
    print(f"camera {distance:g} away: triangles on {tri_hits:.2f} of the pixels, primid {ids:.6f} rgb8<=1 {frac:.6f} psnr {psnr:.2f} dB")
    assert tri_hits > 0.02        # (the lens does point at the meshes)
    print(f"   closest-hit rays {gpu['stats']['rays_closest']} / {cpu['stats']['rays_closest']} (oracle)")
    assert gpu["stats"]["rays_closest"] == cpu["stats"]["rays_closest"]
    assert ids >= 0.9999 and frac >= 0.999 and psnr >= 50.0
