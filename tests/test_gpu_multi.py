"""rt_multi (include/rt_hip.h): the multi-GPU render loop INSIDE the C ABI -- one context and one host thread per GPU,
interleaved 16x16 tiles, one RCCL send/recv exchange at frame end, assembly on rank 0.  The driver's box has one GPU, so:
(a) one rank with its tiles pushed through ncclSend/ncclRecv to itself (the RCCL calls, stream ordering and the assembly
of a gathered buffer), (b) two and three ranks on the same card (two contexts, two host threads, interleaved tiles, the
exchange by device copies because RCCL wants one GPU per rank) -- all bit-identical to rt_render."""
import numpy as np
import pytest

from conftest import get_scene

pytestmark = pytest.mark.gpu


def _single(scene, w, h, spp, depth):
    from raytracer_amd.renderer import Renderer
    r = Renderer(0).set_scene(scene)
    out = r.render(w, h, spp=spp, max_depth=depth)
    r.close()
    return out


def _same(a, b):
    for k in ("rgb8", "primid"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["rgbf"].view(np.uint32), b["rgbf"].view(np.uint32))  # float image by its bits


def test_one_rank_through_rccl_send_recv_equals_rt_render():
    from raytracer_amd.renderer import MultiRenderer
    scene = get_scene("sample", False)
    ref = _single(scene, 150, 100, 4, 3)
    m = MultiRenderer([0], rehearse_rccl=True).set_scene(scene)
    assert m.transport == "rccl" and m.n == 1
    for _ in range(2):  # buffers and communicator are reused across frames
        out = m.render(150, 100, spp=4, max_depth=3)
        _same(out, ref)
    assert out["stats"][0]["rays_closest"] == ref["stats"]["rays_closest"]
    m.close()


@pytest.mark.parametrize("ranks", [2, 3])
def test_ranks_sharing_one_gpu_equal_rt_render(ranks):
    from raytracer_amd.renderer import MultiRenderer
    scene = get_scene("competition", False)
    w, h = 200, 120  # ragged edge tiles; 13 x 8 = 104 tiles, not a multiple of 3
    ref = _single(scene, w, h, 4, 3)
    m = MultiRenderer([0] * ranks).set_scene(scene)
    assert m.transport == "copy" and m.n == ranks
    out = m.render(w, h, spp=4, max_depth=3)
    _same(out, ref)
    st = out["stats"]
    assert sum(s["primary_samples"] for s in st) == ref["stats"]["primary_samples"] == w * h * 4
    for k in ("rays_closest", "rays_shadow", "hits_shaded"):
        assert sum(s[k] for s in st) == ref["stats"][k], k
    out8 = m.render(w, h, spp=4, max_depth=3, want=("rgb8",))  # without the float image / prim ids
    assert np.array_equal(out8["rgb8"], ref["rgb8"])
    m.close()


def test_timings_and_the_timed_multi_abi_mode_of_bench():
    """rt_multi_last_timings tells where a frame's wall time went (per-rank share, exchange, assembly, read-back), and
    bench.py --multi-abi times the path a C++ host links -- rehearsed with two ranks on the one card."""
    import json, os, subprocess, sys
    from raytracer_amd.renderer import MultiRenderer
    scene = get_scene("competition", False)
    m = MultiRenderer([0, 0]).set_scene(scene)
    for _ in range(2):  # (the worker threads of the rt_multi outlive a frame)
        out = m.render(200, 120, spp=4, max_depth=3, want=("rgb8",))
    t = out["timings"]
    assert t["ms_share_max"] >= t["ms_share_min"] > 0 and t["ms_render_wall"] >= t["ms_share_max"] * 0.5
    assert t["ms_exchange"] >= 0 and t["ms_assemble"] > 0 and t["ms_readback"] > 0
    m.close()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--multi-abi", "--multi-devices", "0,0", "--gpus", "2", "--steps", "2",
                          "--warmup", "1", "--scene", "sample", "--width", "320", "--height", "208", "--spp", "4"],
                         cwd=repo, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    d = json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and len(d["multi"]["share_ms"]) == 2 and "copy" in d["multi"]["path"]
