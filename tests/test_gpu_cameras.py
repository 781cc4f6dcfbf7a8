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
