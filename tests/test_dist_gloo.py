"""Multi-GPU data path on CPU: 2 ranks over gloo.  Each rank "renders" its interleaved tiles (with the CPU oracle
standing in for the GPU), packs them tile-major exactly as rt_render_tiles_device does, the product's gather
(raytracer_amd.dist.gather_tiles, the same call bench.py makes over RCCL) moves them to rank 0, and rank 0's
de-interleaved image must equal the single-rank image bit for bit.  The GPU counterpart (rt_assemble_tiles_device on
the gathered layout) is tests/test_gpu_parity.py::test_interleaved_tiles_assemble_to_the_single_gpu_image."""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, WORLD = 150, 100, 2  # ragged edge tiles, odd tile count


def to_tile_major(img, rank, world, per):
    """row-major image -> this rank's tile-major buffer [per, 256, ...] (layout of include/rt_hip.h)"""
    h, w = img.shape[:2]
    tx = (w + 15) // 16
    ty = (h + 15) // 16
    out = np.zeros((per, 256) + img.shape[2:], img.dtype)
    for k, t in enumerate(range(rank, tx * ty, world)):
        y0, x0 = (t // tx) * 16, (t % tx) * 16
        tile = np.zeros((16, 16) + img.shape[2:], img.dtype)
        blk = img[y0:y0 + 16, x0:x0 + 16]
        tile[:blk.shape[0], :blk.shape[1]] = blk
        out[k] = tile.reshape((256,) + img.shape[2:])
    return out


def from_tile_major(g, w, h, world):
    """gathered [world, per, 256, ...] -> row-major image: host mirror of k_assemble, used only as the checker"""
    tx = (w + 15) // 16
    ty = (h + 15) // 16
    img = np.zeros((h, w) + g.shape[3:], g.dtype)
    for t in range(tx * ty):
        y0, x0 = (t // tx) * 16, (t % tx) * 16
        tile = g[t % world, t // world].reshape((16, 16) + g.shape[3:])
        img[y0:y0 + 16, x0:x0 + 16] = tile[:min(16, h - y0), :min(16, w - x0)]
    return img


def _worker(rank, port, ret):
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    from oracle.binding import Oracle
    from raytracer_amd.dist import gather_tiles, tile_counts, tiles_per_rank
    from raytracer_amd.scene import FlatScene
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=WORLD)
    try:
        n_local, n_total = tile_counts(W, H, rank, WORLD)
        counts = [torch.zeros(1, dtype=torch.int64) for _ in range(WORLD)]
        dist.all_gather(counts, torch.tensor([n_local]))
        assert sum(int(c) for c in counts) == n_total == ((W + 15) // 16) * ((H + 15) // 16)
        per = tiles_per_rank(W, H, WORLD)
        assert per >= n_local
        oracle = Oracle(FlatScene.builtin("sample", lazy=True))
        mine = oracle.render(W, H, spp=4, max_depth=2, tile_rank=rank, tile_world=WORLD, threads=2)
        t8 = torch.from_numpy(to_tile_major(mine["rgb8"], rank, WORLD, per))
        tp = torch.from_numpy(to_tile_major(mine["primid"].view(np.int32), rank, WORLD, per))
        g = gather_tiles([t8, tp], rank, WORLD, dist)
        if rank == 0:
            full = oracle.render(W, H, spp=4, max_depth=2, threads=2)
            img8 = from_tile_major(g[0].numpy(), W, H, WORLD)
            imgp = from_tile_major(g[1].numpy(), W, H, WORLD).view(np.uint32)
            ret["rgb8_equal"] = bool(np.array_equal(img8, full["rgb8"]))
            ret["primid_equal"] = bool(np.array_equal(imgp, full["primid"]))
        else:
            assert g is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_assembles_the_single_rank_image():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(port, ret), nprocs=WORLD, join=True)
    assert ret.get("rgb8_equal") and ret.get("primid_equal"), dict(ret)


def test_tile_partition_covers_every_tile_once():
    sys.path.insert(0, REPO)
    from raytracer_amd.dist import tile_counts
    for w, h in ((16, 16), (17, 33), (3840, 2160), (150, 100)):
        for world in (1, 2, 3, 8):
            total = tile_counts(w, h, 0, world)[1]
            assert sum(tile_counts(w, h, r, world)[0] for r in range(world)) == total
            assert tile_counts(w, h, 0, world)[0] == max(tile_counts(w, h, r, world)[0] for r in range(world))
