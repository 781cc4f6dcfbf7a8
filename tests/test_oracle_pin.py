"""Pins the CPU oracle (oracle/rt_oracle.cpp) against the compiled reference.

Every fixture in tests/golden was produced by oracle/_ref/ref_harness_O2 -- the unmodified reference sources built
`g++ -O2 -fopenmp` -- through oracle/make_golden.py.  The oracle restates the reference operation for operation, so
the bar here is BIT-EXACT floats (NaN samples included), not a tolerance.  CPU only.
"""
import json

import numpy as np
import pytest

from conftest import image_metrics, render_params
from raytracer_amd import _native as N


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_perlin_bit_exact(golden):
    from oracle.binding import eval_perlin
    g = golden("perlin")
    assert np.array_equal(bits(eval_perlin(g["points"])), bits(g["value"]))


@pytest.mark.parametrize("name", ["perlinNormal", "perlinWater", "opal", "perlinTerrain", "perlinIceTerrain",
                                  "snowTerrain", "qwilfishTexture"])
def test_procedural_textures_bit_exact(golden, oracle_for, name):
    g = golden("textures")
    out = oracle_for("sample").eval_texture(N.TEX[name], g["uv"])
    assert np.array_equal(bits(out), bits(g[name]))


@pytest.mark.parametrize("name", ["colorAt", "normalAt", "roughnessAt", "ambientOcclusionAt"])
def test_bmp_textures_bit_exact(golden, oracle_for, name):
    g = golden("textures")
    out = oracle_for("sample").eval_texture(N.TEX[name], g["uv_bmp"])
    assert np.array_equal(bits(out), bits(g[name]))


@pytest.mark.parametrize("scene", ["sample", "competition", "bunny", "cones"])
def test_rays_bit_exact(golden, oracle_for, scene):
    g = golden("rays_" + scene)
    o = oracle_for(scene)
    c0, _, _, _ = o.trace_batch(g["rays"], depth=0)
    c3, hits, shadow, _ = o.trace_batch(g["rays"], depth=3)
    hit = g["hit"].astype(bool)
    assert np.array_equal(hits["hit"].astype(bool), hit)
    assert np.array_equal(hits["prim"], g["prim"])
    assert np.array_equal(hits["object"][hit], g["object"][hit])
    assert np.array_equal(hits["triangle"][hit], g["triangle"][hit])
    assert np.array_equal(bits(hits["distance"][hit]), bits(g["distance"][hit]))
    for k in ("intersection", "normal", "normal_shading", "uv", "tangent", "bitangent"):
        assert np.array_equal(bits(hits[k][hit]), bits(g[k][hit])), k
    assert np.array_equal(shadow[hit], g["shadow"][hit])
    assert np.array_equal(bits(c0), bits(g["color_d0"]))
    assert np.array_equal(bits(c3), bits(g["color_d3"]))


@pytest.mark.parametrize("name", ["sample_320x210", "competition_256x192", "bunny_160x120", "sample_160x105_spp16",
                                  "competition_96x72_spp16", "sample_160x105_depth0", "competition_100x70_depth1",
                                  "cones_320x240", "cones_150x110_spp16"])
def test_render_bit_exact(golden, oracle_for, name):
    g = golden("render_" + name)
    p = render_params(g)
    r = oracle_for(p["scene"]).render(p["width"], p["height"], spp=p["spp"], max_depth=p["depth"], seed=p["seed"])
    assert np.array_equal(bits(r["rgbf"]), bits(g["rgbf"]))          # float image incl. NaN samples
    valid = ~g["invalid"]
    assert np.array_equal(r["rgb8"][valid], g["rgb8"][valid])        # truncated 8-bit image
    assert np.array_equal(r["primid"], g["primid"])                  # primitive hit by sample 0
    frac, psnr = image_metrics(r["rgb8"], g["rgb8"], g["invalid"])
    assert frac >= 0.999


def test_ofast_noise_floor_is_recorded(golden):
    """the reference is not reproducible against itself across -O2/-Ofast; keep the numbers visible."""
    g = golden("render_sample_320x210")
    frac, psnr = image_metrics(g["rgb8_ofast"], g["rgb8"], g["invalid"])
    assert frac < 1.0 and psnr < 50.0  # the -Ofast build would FAIL the acceptance bar against the IEEE build
    g = golden("render_competition_256x192")
    frac, psnr = image_metrics(g["rgb8_ofast"], g["rgb8"], g["invalid"])
    assert frac > 0.99


def test_ray_counts_match_survey(oracle_for):
    """rays per primary sample counted the way the survey probe counted them (Ray constructions)"""
    r = oracle_for("sample").render(160, 120)
    st = r["stats"]
    per = (st["rays_closest"] + st["rays_shadow"]) / st["primary_samples"]
    assert 6.0 < per < 10.0
