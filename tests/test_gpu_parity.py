"""GPU parity tests: the HIP wavefront pipeline, called through the C ABI, against (a) the golden fixtures made by
the compiled reference and (b) the CPU oracle on the same seeded inputs.

Acceptance bar (BASELINE.json): primary-hit primitive ids equal on >= 99.99 % of pixels, 8-bit tone-mapped RGB within
1 on >= 99.9 % of pixels, PSNR >= 50 dB.  Arithmetic is FP32 with the reference's operation order; device libm
(pow/exp/acos/asin/atan2) differs from glibc by ulps, so colours carry a tolerance while pure +,-,*,/,sqrt,floor
paths (Perlin, intersection distances) are expected bit-exact.
"""
import numpy as np
import pytest

from conftest import get_scene, image_metrics, load_golden, render_params
from raytracer_amd import _native as N

pytestmark = pytest.mark.gpu

_renderers = {}


def renderer_for(name):
    from raytracer_amd.renderer import Renderer
    if name not in _renderers:
        _renderers[name] = Renderer(0).set_scene(get_scene(name, False))
    return _renderers[name]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_library_loaded_is_in_tree():
    lib = N.hip_lib()
    assert "raytracer_amd/csrc/libraytracer_hip.so" in N.HIP_LIB_PATH and lib is not None


def test_perlin_matches_reference():
    g = load_golden("perlin")
    out = renderer_for("sample").eval_perlin(g["points"])
    assert np.array_equal(bits(out), bits(g["value"]))  # +,-,*,floor only: bit-exact


@pytest.mark.parametrize("name", ["perlinNormal", "perlinWater", "opal", "perlinTerrain", "perlinIceTerrain",
                                  "snowTerrain", "qwilfishTexture"])
def test_procedural_textures(name):
    g = load_golden("textures")
    out = renderer_for("sample").eval_texture(N.TEX[name], g["uv"])
    assert np.allclose(out, g[name], rtol=0, atol=2e-6)
    assert (bits(out) == bits(g[name])).mean() > 0.99


@pytest.mark.parametrize("name", ["colorAt", "normalAt", "roughnessAt", "ambientOcclusionAt"])
def test_bmp_textures(name):
    g = load_golden("textures")
    out = renderer_for("sample").eval_texture(N.TEX[name], g["uv_bmp"])
    assert (np.abs(out - g[name]).max(axis=1) == 0).mean() >= 0.999  # a texel boundary may flip under device fmod


@pytest.mark.parametrize("scene", ["sample", "competition", "bunny", "cones"])
def test_rays_against_reference(scene):
    g = load_golden("rays_" + scene)
    r = renderer_for(scene)
    c0, _, _, _ = r.trace_batch(g["rays"], depth=0, want_hits=False, want_shadow=False)
    c3, hits, shadow, st = r.trace_batch(g["rays"], depth=3)
    n = len(g["rays"])
    same = hits["prim"] == g["prim"]
    both = same & (g["hit"] == 1)
    dist_bits = float((bits(hits["distance"][both]) == bits(g["distance"][both])).mean())
    fields = {k: float(np.isclose(hits[k][both], g[k][both], rtol=1e-5, atol=1e-5, equal_nan=True).all(axis=1).mean())
              for k in ("intersection", "normal", "normal_shading", "uv", "tangent", "bitangent")}
    sh = float((shadow[both] == g["shadow"][both]).all(axis=1).mean())
    cols = []
    for got, ref in ((c0, g["color_d0"]), (c3, g["color_d3"])):
        valid = ~np.isnan(ref).any(axis=1)
        cols.append(float(((np.abs(got - ref).max(axis=1) <= 1e-5) & valid).sum() / n))
        cols.append(float(np.abs(got - ref)[valid].max()))
    # achieved values, for profiles/r02_parity_margins.md (pytest -s)
    print(f"rays_{scene}: prim ids {same.mean():.5f} ({int((~same).sum())} of {n} differ), distance bit-equal {dist_bits:.5f}, "
          f"hit fields {min(fields.values()):.5f} (worst of {', '.join(fields)}), shadow flags {sh:.5f}, "
          f"colour depth 0 within 1e-5: {cols[0]:.5f} (max |d| {cols[1]:.2e}), depth 3: {cols[2]:.5f} (max |d| {cols[3]:.2e}), "
          f"NaN-poisoned in the reference: {int(np.isnan(g['color_d3']).any(axis=1).sum())}")
    # thresholds = what is achieved on every fixture (r02 margins table) minus a small margin; the rays that differ are
    # grazing / exact-edge cases where the hit test's outcome hangs on the last bit of a device libm / division result
    # (round 2, MI355X / ROCm 7.2: ids, distances and Hit fields agree on EVERY fixture ray, colours to 2.4e-7, one shadow
    # flag of 13 000 differs: profiles/r02_parity_margins.md)
    assert same.mean() >= 0.9999, f"primitive ids agree on {same.mean():.5f}"
    assert np.array_equal(bits(hits["distance"][both]), bits(g["distance"][both])) or \
        np.allclose(hits["distance"][both], g["distance"][both], rtol=2e-6)
    assert np.array_equal(hits["object"][both], g["object"][both])
    assert np.array_equal(hits["triangle"][both], g["triangle"][both])
    for k, v in fields.items():
        assert v >= 0.9999, k
    assert sh >= 0.9995
    assert cols[0] >= 0.9999 and cols[2] >= 0.9999  # (NaN-poisoned reference colours count as mismatches: none in the fixtures)


@pytest.mark.parametrize("name", ["sample_320x210", "competition_256x192", "bunny_160x120", "sample_160x105_spp16",
                                  "competition_96x72_spp16", "sample_160x105_depth0", "competition_100x70_depth1",
                                  "cones_320x240", "cones_150x110_spp16"])
def test_render_against_reference(name):
    g = load_golden("render_" + name)
    p = render_params(g)
    out = renderer_for(p["scene"]).render(p["width"], p["height"], spp=p["spp"], max_depth=p["depth"], seed=p["seed"])
    ids = (out["primid"] == g["primid"]).mean()
    frac, psnr = image_metrics(out["rgb8"], g["rgb8"], g["invalid"])
    print(f"{name}: primid {ids:.6f} rgb8<=1 {frac:.6f} psnr {psnr:.2f} dB")
    assert ids >= 0.9999
    assert frac >= 0.999
    assert psnr >= 50.0
    assert out["stats"]["queue_overflow"] == 0


@pytest.mark.parametrize("name", ["full_competition_1024x768", "full_sample_1920x1080_spp16"])
def test_baseline_configs_at_stated_size_against_reference(name):
    """BASELINE.json configs[0] (main.cpp as shipped, minus the missing pietre.obj: 1024x768, 4 spp, main.cpp:543-545,
    625) and configs[1] (sample scene 1920x1080, 16 spp) at their stated sizes, against goldens rendered by the COMPILED
    REFERENCE (oracle/make_golden.py --only full: 355 s and 189 s of reference time on 8 cores), at the acceptance bar."""
    g = load_golden("render_" + name)
    p = render_params(g)
    w, h = p["width"], p["height"]
    invalid = np.unpackbits(g["invalid"])[:w * h].reshape(h, w).astype(bool)
    out = renderer_for(p["scene"]).render(w, h, spp=p["spp"], max_depth=p["depth"], seed=p["seed"], want=("rgb8", "primid"))
    ids = (out["primid"] == g["primid"]).mean()
    frac, psnr = image_metrics(out["rgb8"], g["rgb8"], invalid)
    worse = int((np.abs(out["rgb8"].astype(int) - g["rgb8"].astype(int)).max(axis=2) > 1).sum())
    print(f"{name}: primid {ids:.6f} rgb8<=1 {frac:.6f} ({worse} px differ by > 1, {int(invalid.sum())} invalid in the reference) psnr {psnr:.2f} dB, "
          f"GPU {out['stats']['ms_total']:.1f} ms vs reference {p['reference_render_s']:.1f} s on {p['reference_threads']} threads")
    assert ids >= 0.9999
    assert frac >= 0.999
    assert psnr >= 50.0
    assert out["stats"]["queue_overflow"] == 0


@pytest.mark.parametrize("scene,w,h", [("sample", 200, 150), ("competition", 128, 96)])
def test_ray_counts_match_oracle(scene, w, h, oracle_for):
    """a ray = one traced closest-hit or shadow query; counts are a property of the algorithm"""
    gpu = renderer_for(scene).render(w, h)["stats"]
    cpu = oracle_for(scene, lazy=False).render(w, h)["stats"]
    assert gpu["primary_samples"] == cpu["primary_samples"] == w * h * 4
    # the reference's shadow-ray count = what the GPU traced + the rays it did not emit because their pending radiance
    # is exactly zero (rt_stats.shadow_zero; the reference casts them all, main.cpp:100)
    gpu = dict(gpu, rays_shadow_reference=gpu["rays_shadow"] + gpu["shadow_zero"])
    cpu = dict(cpu, rays_shadow_reference=cpu["rays_shadow"])
    print(f"{scene}: shadow rays traced {gpu['rays_shadow']}, not emitted (zero radiance) {gpu['shadow_zero']}, reference {cpu['rays_shadow']}")
    for k in ("rays_closest", "rays_shadow_reference", "shadow_culled", "hits_shaded"):
        assert abs(gpu[k] - cpu[k]) <= 0.002 * cpu[k] + 4, (k, gpu[k], cpu[k])
    assert gpu["shadow_zero"] > 0  # both scenes hold materials with refraction 1


def test_render_is_deterministic_and_chunk_independent():
    r = renderer_for("competition")
    a = r.render(160, 112, spp=4)
    b = r.render(160, 112, spp=4)
    c = r.render(160, 112, spp=4, chunk_samples=4096)  # many small chunks
    for k in ("rgb8", "rgbf", "primid"):
        assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a[k], c[k]), k


def test_interleaved_tiles_assemble_to_the_single_gpu_image():
    """multi-GPU data path on one GPU: every rank's tiles rendered in turn, buffers laid out as a gather would,
    assembled with rt_assemble_tiles_device; must equal rt_render bit for bit (fixed-point accumulation)"""
    import torch
    r = renderer_for("sample")
    w, h, world = 150, 100, 3  # ragged tiles, world does not divide the tile count
    full = r.render(w, h, spp=4)
    per = max(r.tile_counts(w, h, k, world)[0] for k in range(world))
    dev = torch.device("cuda:0")
    t8 = torch.zeros(world, per, 256, 3, dtype=torch.uint8, device=dev)
    tf = torch.zeros(world, per, 256, 3, dtype=torch.float32, device=dev)
    tp = torch.zeros(world, per, 256, dtype=torch.int32, device=dev)
    for k in range(world):
        r.render_tiles_device(w, h, k, world, t8[k].data_ptr(), tf[k].data_ptr(), tp[k].data_ptr(), spp=4)
    o8 = torch.zeros(h, w, 3, dtype=torch.uint8, device=dev)
    of = torch.zeros(h, w, 3, dtype=torch.float32, device=dev)
    op = torch.zeros(h, w, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    r.assemble_tiles_device(w, h, world, per, t8.data_ptr(), tf.data_ptr(), tp.data_ptr(), o8.data_ptr(), of.data_ptr(), op.data_ptr())
    assert np.array_equal(o8.cpu().numpy(), full["rgb8"])
    assert np.array_equal(of.cpu().numpy(), full["rgbf"])
    assert np.array_equal(op.cpu().numpy().view(np.uint32), full["primid"])


def test_empty_and_error_paths():
    from raytracer_amd.renderer import Renderer, RtError
    r = renderer_for("bunny")
    col, hits, sh, st = r.trace_batch(np.zeros((0, 6), np.float32))
    assert col.shape == (0, 3)
    with pytest.raises(RtError):
        r.render(64, 64, spp=5, jitter_mode=N.JITTER_TABLE)  # the reference table only defines 4 samples
    with pytest.raises(RtError):
        r.render(0, 64)
    fresh = Renderer(0)
    with pytest.raises(RtError):
        fresh.render(16, 16, camera=N.Camera())  # nothing committed
    fresh.close()
    # a ray that leaves the scene: black, miss record
    col, hits, sh, st = renderer_for("competition").trace_batch(np.array([[0, 50, 0, 0, 1, 0]], np.float32))
    assert hits["hit"][0] == 0 and hits["prim"][0] == N.PRIM_MISS and np.all(col == 0)


def test_stress_lattice_against_oracle(oracle_for):
    """SURVEY.md 8(d) C5 in small: the competition meshes replicated on a lattice (here 9 replicas, 614 k triangles;
    synthetic, no reference counterpart, so the checker is the pinned oracle).  Exercises a deeper tree: stack rows
    beyond the LDS part, and -- when the worst-case stack exceeds 64 -- the path without packet kernels."""
    name = "stress:9"
    r = renderer_for(name)
    w, h = 128, 96
    gpu = r.render(w, h, spp=4)
    cpu = oracle_for(name, lazy=False).render(w, h, spp=4)
    ids = (gpu["primid"] == cpu["primid"]).mean()
    frac, psnr = image_metrics(gpu["rgb8"], cpu["rgb8"])
    st = gpu["stats"]
    print(f"{name}: {st['n_triangles']} triangles, {st['bvh_nodes']} nodes, depth {st['bvh_depth']}, max stack {st['bvh_max_stack']}; "
          f"primid {ids:.6f} rgb8<=1 {frac:.6f} psnr {psnr:.2f} dB")
    assert st["n_triangles"] == 9 * 68202
    assert ids >= 0.9999 and frac >= 0.999 and psnr >= 50.0
    assert st["queue_overflow"] == 0


def test_cpp_main_renders_what_the_c_abi_renders(tmp_path):
    """raytracer_amd/host/raytracer_hip -- the reference's main() restated on the kept C++ surface (scene function ->
    globals -> render() -> Image -> PPM) -- must write exactly the image rt_render returns for the same parameters"""
    import os
    import subprocess
    exe = os.path.join(N.PKG_DIR, "host", "raytracer_hip")
    out = str(tmp_path / "cones.ppm")
    subprocess.run([exe, "--scene", "cones", "--width", "96", "--height", "64", "--spp", "4", "--depth", "3",
                    "--assets", N.ASSET_DIR, out], check=True, timeout=120)
    raw = open(out, "rb").read()
    head = raw.split(b"\n", 3)
    assert head[0] == b"P6" and head[1] == b"96 64" and head[2] == b"255"
    img = np.frombuffer(head[3], np.uint8).reshape(64, 96, 3)
    ref = renderer_for("cones").render(96, 64, spp=4, max_depth=3)
    assert np.array_equal(img, ref["rgb8"])
    # the reference's own ASCII P3 layout on request
    out3 = str(tmp_path / "cones_p3.ppm")
    subprocess.run([exe, "--scene", "cones", "--width", "32", "--height", "16", "--p3", "--assets", N.ASSET_DIR, out3], check=True, timeout=120)
    tok = open(out3).read().split()
    assert tok[:4] == ["P3", "32", "16", "255"] and len(tok) == 4 + 32 * 16 * 3
