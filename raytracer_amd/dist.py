"""Multi-GPU frames: interleaved 16x16 tiles over ranks, scene + BVH replicated, ONE exchange at frame end.

Tile t of the image belongs to rank t % world (rt_tile_counts / rt_render_tiles_device in include/rt_hip.h); a rank
renders its tiles into tile-major device buffers; `gather_tiles` moves them to rank 0 with torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) and rank 0 de-interleaves them with
rt_assemble_tiles_device.  There is no collective during rendering.  torch is plumbing here (buffers, streams,
process group), never a compute path.
"""
import ctypes as C

from . import _native as N

TILE = 16
TILE_PIXELS = TILE * TILE


def tile_counts(width, height, rank, world):
    """(tiles owned by `rank`, tiles of the image) -- host arithmetic of the C ABI, needs no GPU"""
    a, b = C.c_int64(), C.c_int64()
    if N.hip_lib().rt_tile_counts(width, height, rank, world, C.byref(a), C.byref(b)):
        raise RuntimeError(N.hip_lib().rt_last_error().decode())
    return a.value, b.value


def tiles_per_rank(width, height, world):
    """tiles every rank's buffer is padded to (rank 0 owns the most), so that one gather of equal sizes suffices"""
    return tile_counts(width, height, 0, world)[0]


def gather_tiles(local, rank, world, dist=None, force_collective=False):
    """Gathers each rank-local tile-major tensor of `local` to rank 0.

    Returns, on rank 0, a list of tensors shaped [world, *local.shape] (the ranks' buffers back to back in rank order:
    the layout rt_assemble_tiles_device expects); None elsewhere.  world == 1 needs no process group."""
    import torch
    if world == 1 and not force_collective:
        return [t.unsqueeze(0) for t in local]
    out = []
    for t in local:
        g = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device) if rank == 0 else None
        dist.gather(t, list(g.unbind(0)) if rank == 0 else None, dst=0)
        out.append(g)
    return out if rank == 0 else None


class DistributedFrame:
    """One rank's share of a frame on its GPU: render own tiles -> gather -> (rank 0) assemble the image."""

    def __init__(self, renderer, width, height, rank, world, device, dist=None, force_collective=False, stage_host=False):
        import torch
        self.r, self.W, self.H, self.rank, self.world, self.dist = renderer, width, height, rank, world, dist
        self.force_collective = force_collective  # run the gather even with one rank (rehearsal of the RCCL path)
        # rehearsal of N ranks on ONE card (RCCL wants a GPU per rank): the gather goes through host tensors and gloo
        self.stage_host = stage_host
        self.per = tiles_per_rank(width, height, world)
        self.device = device
        self.t8 = torch.zeros(self.per, TILE_PIXELS, 3, dtype=torch.uint8, device=device)
        self.tp = torch.zeros(self.per, TILE_PIXELS, dtype=torch.int32, device=device)
        self.collective = world > 1 or force_collective
        if rank == 0:
            self.o8 = torch.zeros(height, width, 3, dtype=torch.uint8, device=device)
            self.op = torch.zeros(height, width, dtype=torch.int32, device=device)
            # receive buffers of the gather, allocated once ([world, ...]: the layout rt_assemble_tiles_device reads)
            self.g8 = torch.empty((world,) + tuple(self.t8.shape), dtype=torch.uint8, device=device) if self.collective else None
            self.gp = torch.empty((world,) + tuple(self.tp.shape), dtype=torch.int32, device=device) if self.collective else None

    def render(self, stream=None, time_parts=False, **kw):
        """returns this rank's rt_stats dict; on rank 0 self.o8 / self.op hold the assembled image afterwards.

        Everything is ordered on ONE stream: torch's current stream of this rank's device unless `stream` (a raw
        hipStream_t) is given -- the render and the assembly are enqueued on it through the C ABI, and the nccl backend
        orders its gather against torch's current stream, so the caller must pass that very stream if it passes one.
        (0 would mean the context's private non-blocking stream, which the gather is NOT ordered against.)
        time_parts: synchronise after the gather and after the assembly and leave {"share_ms", "gather_ms", "assemble_ms"}
        (this rank's GPU time for its tiles; host wall time of the two steps behind it) in self.timings -- so that a
        scaling loss can be assigned to the partition, the cost of a share, or the exchange."""
        import time
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0
            if self.device.type == "cuda" and stream == 0:
                # torch's legacy default stream has handle 0, which the ABI reads as "context stream": render on the
                # context stream then, and fence both ways around the gather with device synchronisation
                torch.cuda.synchronize(self.device)
        st = self.r.render_tiles_device(self.W, self.H, self.rank, self.world, self.t8.data_ptr(), 0, self.tp.data_ptr(),
                                        stream=stream, **kw)   # (complete when the call returns)
        t_render = time.perf_counter()
        if not self.collective:
            g8, gp = self.t8, self.tp
        elif self.stage_host:
            torch.cuda.synchronize(self.device)
            for loc, dst in ((self.t8, getattr(self, "g8", None)), (self.tp, getattr(self, "gp", None))):
                h = loc.cpu()
                hg = torch.empty((self.world,) + tuple(h.shape), dtype=h.dtype) if self.rank == 0 else None
                self.dist.gather(h, list(hg.unbind(0)) if self.rank == 0 else None, dst=0)
                if self.rank == 0:
                    dst.copy_(hg)
            g8, gp = (self.g8, self.gp) if self.rank == 0 else (None, None)
            torch.cuda.synchronize(self.device)
        else:
            self.dist.gather(self.t8, list(self.g8.unbind(0)) if self.rank == 0 else None, dst=0)
            self.dist.gather(self.tp, list(self.gp.unbind(0)) if self.rank == 0 else None, dst=0)
            g8, gp = (self.g8, self.gp) if self.rank == 0 else (None, None)
            if stream == 0 and self.device.type == "cuda":
                torch.cuda.synchronize(self.device)  # the assembly below runs on the context stream: wait for the gather
        if time_parts and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        t_gather = time.perf_counter()
        if self.rank == 0:
            self.r.assemble_tiles_device(self.W, self.H, self.world, self.per, g8.data_ptr(), 0, gp.data_ptr(),
                                         self.o8.data_ptr(), 0, self.op.data_ptr(), stream=stream)
        if time_parts:
            self.timings = {"share_ms": st["ms_total"], "gather_ms": (t_gather - t_render) * 1e3,
                            "assemble_ms": (time.perf_counter() - t_gather) * 1e3}
        return st
