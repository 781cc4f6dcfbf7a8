"""rt_render_tiles_static (include/rt_hip.h): the render loop on the library's static scratch, for callers that cannot
allocate.  Called through the C ABI and compared with rt_render_tiles_device bit for bit, on two calls in a row, the
second with a smaller input (the scratch keeps the first call's contents: counters must be cleared, queues rewritten)."""
import ctypes as C

import numpy as np
import pytest

from conftest import get_scene
from raytracer_amd import _native as N

pytestmark = pytest.mark.gpu


def _both(r, scene, w, h, spp, depth, rank=0, world=1):
    import torch
    lib = N.hip_lib()
    dev = torch.device("cuda:0")
    n_local = r.tile_counts(w, h, rank, world)[0]
    bufs = {}
    for tag in ("dev", "static"):
        bufs[tag] = (torch.full((n_local, 256, 3), 7, dtype=torch.uint8, device=dev),
                     torch.full((n_local, 256, 3), -1.0, dtype=torch.float32, device=dev),
                     torch.full((n_local, 256), 5, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    t8, tf, tp = bufs["dev"]
    st = r.render_tiles_device(w, h, rank, world, t8.data_ptr(), tf.data_ptr(), tp.data_ptr(), spp=spp, max_depth=depth)
    s8, sf, sp = bufs["static"]
    p = N.RenderParams(width=w, height=h, spp=spp, max_depth=depth, seed=0, jitter_mode=N.JITTER_AUTO, tile_rank=rank,
                       tile_world=world, chunk_samples=0, collect_stats=0)
    stream = torch.cuda.Stream(device=dev)
    rc = lib.rt_render_tiles_static(r._ctx, C.byref(scene.camera), C.byref(p), C.c_void_p(s8.data_ptr()),
                                    C.c_void_p(sf.data_ptr()), C.c_void_p(sp.data_ptr()), C.c_void_p(stream.cuda_stream))
    assert rc == 0, lib.rt_last_error().decode()
    stream.synchronize()  # the entry point only enqueues
    for a, b, name in ((t8, s8, "rgb8"), (tf, sf, "rgbf"), (tp, sp, "primid")):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), name  # float image compared by its bits
    return st


def test_static_scratch_entry_point_matches_the_allocating_one():
    from raytracer_amd.renderer import Renderer
    scene = get_scene("competition", False)
    r = Renderer(0).set_scene(scene)
    # first call: several chunks of the static scratch (ragged tiles, 4 lights, depth 3)
    st = _both(r, scene, 200, 136, 4, 3)
    assert st["queue_overflow"] == 0 and st["rays_shadow"] > 0
    # second call in a row, smaller input, other tile partition, more samples per tile
    _both(r, scene, 72, 40, 16, 2, rank=1, world=2)
    # a third with the first size again (what the second left in the scratch must not matter)
    _both(r, scene, 200, 136, 4, 3)
    r.close()


def test_static_scratch_rejects_what_does_not_fit():
    from raytracer_amd.renderer import Renderer
    scene = get_scene("competition", False)
    r = Renderer(0).set_scene(scene)
    lib = N.hip_lib()
    p = N.RenderParams(width=64, height=64, spp=4096, max_depth=3, seed=0, jitter_mode=N.JITTER_HASH, tile_rank=0,
                       tile_world=1, chunk_samples=0, collect_stats=0)
    rc = lib.rt_render_tiles_static(r._ctx, C.byref(scene.camera), C.byref(p), None, None, None, None)
    assert rc != 0 and b"static scratch" in lib.rt_last_error()
    r.close()
