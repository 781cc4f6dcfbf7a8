"""Full-size checks (BASELINE.json configs[3]: competition scene, 3840x2160): the oracle cannot render 530 M samples,
so parity at this size is (a) the oracle on a SAMPLE of the frame's tiles -- tile t of 2025 interleaved "ranks", i.e. 16
tiles spread over the whole 4K image, same jitter hash -- and (b) size-independent properties of the whole frame at
64 spp: bit-identical results across runs, chunk sizes and tile partitions (fixed-point accumulation), ray-count
bookkeeping, no queue overflow."""
import numpy as np
import pytest

from conftest import get_scene, image_metrics

pytestmark = pytest.mark.gpu

W, H = 3840, 2160
_r = {}


def renderer():
    from raytracer_amd.renderer import Renderer
    if "r" not in _r:
        _r["r"] = Renderer(0).set_scene(get_scene("competition", False))
    return _r["r"]


def render_rank(rank, world, spp, **kw):
    """this rank's tiles as (rgb8 [n,256,3], primid [n,256], stats) device tensors; the float image goes to _r["f"]"""
    import torch
    r = renderer()
    n = r.tile_counts(W, H, rank, world)[0]
    t8 = torch.zeros(n, 256, 3, dtype=torch.uint8, device="cuda:0")
    tp = torch.zeros(n, 256, dtype=torch.int32, device="cuda:0")
    tf = torch.zeros(n, 256, 3, dtype=torch.float32, device="cuda:0")
    st = r.render_tiles_device(W, H, rank, world, t8.data_ptr(), tf.data_ptr(), tp.data_ptr(), spp=spp, **kw)
    torch.cuda.synchronize()
    _r["f"] = tf.view(torch.int32)  # float bits
    return t8, tp, st


def test_sampled_tiles_of_the_4k_frame_match_the_oracle(oracle_for):
    world, rank, spp = 2025, 1017, 16
    t8, tp, st = render_rank(rank, world, spp)
    ref = oracle_for("competition", lazy=False).render(W, H, spp=spp, max_depth=3, tile_rank=rank, tile_world=world)
    tx = (W + 15) // 16
    tiles = list(range(rank, tx * ((H + 15) // 16), world))
    assert len(tiles) == t8.shape[0] >= 16
    g8, gp = t8.cpu().numpy(), tp.cpu().numpy().view(np.uint32)
    ref8 = np.zeros_like(g8)
    refp = np.zeros_like(gp)
    for k, t in enumerate(tiles):
        y0, x0 = (t // tx) * 16, (t % tx) * 16
        ref8[k] = ref["rgb8"][y0:y0 + 16, x0:x0 + 16].reshape(256, 3)   # 3840 and 2160 are multiples of 16... 2160 = 135 * 16
        refp[k] = ref["primid"][y0:y0 + 16, x0:x0 + 16].reshape(256)
    ids = float((gp == refp).mean())
    frac, psnr = image_metrics(g8.reshape(-1, 16, 3), ref8.reshape(-1, 16, 3))
    print(f"4K sample of {len(tiles)} tiles: primid {ids:.5f} rgb8<=1 {frac:.5f} psnr {psnr:.2f} dB")
    assert ids >= 0.9999 and frac >= 0.999 and psnr >= 50.0
    assert st["queue_overflow"] == 0 and st["primary_samples"] == len(tiles) * 256 * spp


def test_full_frame_64spp_is_deterministic_chunk_and_partition_independent():
    import torch
    a8, ap, sa = render_rank(0, 1, 64)
    af = _r["f"]
    b8, bp, sb = render_rank(0, 1, 64, chunk_samples=3 << 20)   # different chunking
    assert torch.equal(a8, b8) and torch.equal(ap, bp)
    assert torch.equal(af, _r["f"])  # the float image too, bit for bit (work is fetched dynamically: no two runs schedule alike)
    assert sa["primary_samples"] == sb["primary_samples"] == W * H * 64
    for k in ("rays_closest", "rays_shadow", "shadow_culled", "hits_shaded"):
        assert sa[k] == sb[k], k
    assert sa["queue_overflow"] == 0 and sa["hits_shaded"] <= sa["rays_closest"]
    assert sa["rays_closest"] >= sa["primary_samples"]
    # two-rank partition: every tile of the single-rank frame reappears bit for bit in its owner's buffer
    del b8, bp
    _r.pop("f")
    world = 2
    tot = {k: 0 for k in ("rays_closest", "rays_shadow", "primary_samples")}
    for rank in range(world):
        r8, rp, sr = render_rank(rank, world, 64)
        assert torch.equal(r8, a8[rank::world]) and torch.equal(rp, ap[rank::world])
        assert torch.equal(_r["f"], af[rank::world])
        for k in tot:
            tot[k] += sr[k]
        del r8, rp
    for k in tot:
        assert tot[k] == sa[k], k


def test_bunny_full_frame_4k_64spp_is_deterministic_across_accumulation_paths():
    """BASELINE.json configs[2] (C3: the bunny scene) at its STATED size, 3840x2160 @64 spp.  The first frame after rt_commit adds
    its level-0 shadow results with atomics; the scene's shadow rays are mostly level-0 rays, so the following frames STORE them
    and fold them afterwards (RtQueues::occ0, k_fold0): the same fixed-point integers, hence the same image bit for bit -- also
    for another chunking.  Plus the ray bookkeeping of the whole frame."""
    import torch
    from raytracer_amd.renderer import Renderer
    r = Renderer(0).set_scene(get_scene("bunny", False))
    n = r.tile_counts(W, H, 0, 1)[0]
    frames = []
    for kw in ({}, {}, {"chunk_samples": 24 << 20}):
        t8 = torch.zeros(n, 256, 3, dtype=torch.uint8, device="cuda:0")
        tp = torch.zeros(n, 256, dtype=torch.int32, device="cuda:0")
        tf = torch.zeros(n, 256, 3, dtype=torch.float32, device="cuda:0")
        st = r.render_tiles_device(W, H, 0, 1, t8.data_ptr(), tf.data_ptr(), tp.data_ptr(), spp=64, **kw)
        torch.cuda.synchronize()
        frames.append((t8, tp, tf.view(torch.int32), st))
        print(f"bunny {W}x{H}@64 {kw or 'library chunking'}: {st['ms_total']:.1f} ms, {st['chunks']} chunks, {st['rays_closest']} + {st['rays_shadow']} rays")
    a = frames[0]
    for b in frames[1:]:
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        for k in ("primary_samples", "rays_closest", "rays_shadow", "shadow_culled", "shadow_zero", "hits_shaded"):
            assert a[3][k] == b[3][k], k
        assert b[3]["queue_overflow"] == 0
    sa = a[3]
    assert sa["primary_samples"] == W * H * 64 and sa["rays_closest"] >= sa["primary_samples"] and sa["hits_shaded"] <= sa["rays_closest"]
    r.close()
