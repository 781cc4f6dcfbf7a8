"""Frame planning from the observed fan-out (rt_frame.inl: planFrame): after the first frame of a scene the queues are sized
for 1.5 x the fan-out seen so far instead of the worst case, so the same workspace holds fewer, larger chunks; a frame that
overflows its queues after all is rendered again with worst-case queues.  Either way the images are those of fixed small
chunks -- and of any other cut of the frame into pipelined chunks."""
import os

import numpy as np
import pytest

from conftest import get_scene


@pytest.mark.gpu
def test_adaptive_chunks_and_overflow_repeat_change_no_bit():
    from raytracer_amd.renderer import Renderer
    r = Renderer(0).set_scene(get_scene("competition", False))
    w, h, spp = 1920, 1080, 16
    want = ("rgb8", "primid", "rgbf")
    ref = r.render(w, h, spp=spp, chunk_samples=1 << 22, collect_stats=1, want=want)   # fixed chunks, worst-case queues
    assert ref["stats"]["chunks"] >= 8 and ref["stats"]["chunk_retries"] == 0
    a = r.render(w, h, spp=spp, collect_stats=1, want=want)                              # sized from what `ref` saw
    assert a["stats"]["chunks"] < ref["stats"]["chunks"] and a["stats"]["chunk_retries"] == 0
    for k in want:
        assert np.array_equal(a[k].view(np.uint32) if k == "rgbf" else a[k], ref[k].view(np.uint32) if k == "rgbf" else ref[k]), k
    for k in ("rays_closest", "rays_shadow", "shadow_zero", "shadow_culled", "hits_shaded"):
        assert a["stats"][k] == ref["stats"][k], k

    # A fresh context (queues only ever grow inside one): a small frame gives it a fan-out, then the full view is rendered
    # with queues sized for a twentieth of that -- the frame overflows and is repeated with worst-case queues.
    r2 = Renderer(0).set_scene(get_scene("competition", False))
    r2.render(64, 48, spp=4, want=())
    os.environ["RT_CHUNK_MARGIN"] = "0.05"
    try:
        b = r2.render(w, h, spp=spp, collect_stats=1, want=want)
    finally:
        del os.environ["RT_CHUNK_MARGIN"]
    assert b["stats"]["chunk_retries"] >= 1 and b["stats"]["queue_overflow"] == 0
    for k in want:
        assert np.array_equal(b[k].view(np.uint32) if k == "rgbf" else b[k], ref[k].view(np.uint32) if k == "rgbf" else ref[k]), k
    for k in ("rays_closest", "rays_shadow", "shadow_zero", "shadow_culled", "hits_shaded"):
        assert b["stats"][k] == ref["stats"][k], k
    c = r2.render(w, h, spp=spp, collect_stats=1, want=("rgb8",))                       # and the next frame is back to normal
    assert c["stats"]["chunk_retries"] == 0 and np.array_equal(c["rgb8"], ref["rgb8"])


@pytest.mark.gpu
def test_single_frame_flag_changes_no_bit():
    """RT_RENDER_SINGLE_FRAME (what raytracer_hip sets): worst-case queues for the first frame, the workspace does not grow
    for the second -- same image."""
    from raytracer_amd.renderer import Renderer
    w, h, spp = 1920, 1080, 16
    ref = Renderer(0).set_scene(get_scene("competition", False)).render(w, h, spp=spp, chunk_samples=1 << 22, want=("rgb8", "primid"))
    r = Renderer(0).set_scene(get_scene("competition", False))
    a = r.render(w, h, spp=spp, want=("rgb8", "primid"), single_frame=True)
    b = r.render(w, h, spp=spp, want=("rgb8", "primid"), single_frame=True)
    assert a["stats"]["chunks"] >= 2 and a["stats"]["chunk_retries"] == 0
    for x in (a, b):
        assert np.array_equal(x["rgb8"], ref["rgb8"]) and np.array_equal(x["primid"], ref["primid"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bunny", "sample"])
def test_level0_shadow_results_stored_or_added_same_bits(name):
    """RtQueues::occ0: where most shadow rays are level-0 rays, k_any stores the radiance of an unoccluded level-0 shadow ray
    at [light][sample] and k_fold0 adds it to the accumulator, instead of three 64-bit atomics per ray.  Same integers."""
    from raytracer_amd.renderer import Renderer
    r = Renderer(0).set_scene(get_scene(name, False))
    w, h, spp = 1280, 720, 16
    out = {}
    for mode in ("0", "1"):
        os.environ["RT_L0_STORE"] = mode
        try:
            r.render(w, h, spp=spp, want=())  # (the first frame after commit also measures the scene's fan-out)
            out[mode] = r.render(w, h, spp=spp, collect_stats=1, want=("rgb8", "rgbf", "primid"))
        finally:
            del os.environ["RT_L0_STORE"]
    for k in ("rgb8", "primid"):
        assert np.array_equal(out["0"][k], out["1"][k]), k
    assert np.array_equal(out["0"]["rgbf"].view(np.uint32), out["1"]["rgbf"].view(np.uint32))
    for k in ("rays_closest", "rays_shadow", "shadow_zero", "shadow_unoccluded", "hits_shaded"):
        assert out["0"]["stats"][k] == out["1"]["stats"][k], k


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["competition", "bunny"])
def test_two_lanes_same_bits(name):
    """Two pipelines side by side (what scenes of >= 8 Mi triangles get by default; RT_LANES forces it): chunk c runs in lane
    c % 2, each lane with queues, counters and a stream of its own -- the image and the ray counts are those of one lane."""
    from raytracer_amd.renderer import Renderer
    r = Renderer(0).set_scene(get_scene(name, False))
    w, h, spp = 1280, 720, 16
    out = {}
    for lanes in ("1", "2"):
        os.environ["RT_LANES"] = lanes
        try:
            r.render(w, h, spp=spp, chunk_samples=1 << 20, want=())
            out[lanes] = r.render(w, h, spp=spp, chunk_samples=1 << 20, collect_stats=1, want=("rgb8", "rgbf", "primid"))
        finally:
            del os.environ["RT_LANES"]
    assert out["2"]["stats"]["chunks"] >= 8
    for k in ("rgb8", "primid"):
        assert np.array_equal(out["1"][k], out["2"][k]), k
    assert np.array_equal(out["1"]["rgbf"].view(np.uint32), out["2"]["rgbf"].view(np.uint32))
    for k in ("rays_closest", "rays_shadow", "shadow_zero", "shadow_unoccluded", "hits_shaded", "nodes_visited", "tris_tested"):
        if k in ("nodes_visited", "tris_tested"):
            continue  # (how many nodes a ray visits depends on its wave's phase votes)
        assert out["1"]["stats"][k] == out["2"]["stats"][k], k
