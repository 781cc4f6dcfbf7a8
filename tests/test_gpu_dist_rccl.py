"""The launch the driver uses for N > 1 (torch.distributed.run, backend "nccl" = RCCL), rehearsed with ONE rank on the
1-GPU box: process group on the GPU, barrier, the gather of the tile-major uint8 / int32 buffers into views of one
tensor, all_reduce of the timings, the JSON line.  (More ranks on one card are refused by RCCL; the partition and
assembly logic for world_size 2 is covered on CPU with gloo in test_dist_gloo.py.)"""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """a TCP port nobody listens on right now (a fixed one collides with a run that is still winding down)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


@pytest.mark.gpu
def test_bench_under_torchrun_with_rccl_gather():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--scene", "sample", "--width", "320", "--height", "208", "--spp", "4", "--no-cpu", "--no-roofline", "--force-dist"]
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0
    assert d["config"]["tiles"] == 20 * 13
    # where the frame's time went, per rank: share, gather, assembly
    assert len(d["multi"]["share_ms"]) == 1 and d["multi"]["share_ms"][0] > 0 and d["multi"]["gather_ms_rank0"] >= 0

    # the same frame without the collective gives the same ray count (the gather changes nothing but where the tiles end up)
    plain = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--scene", "sample",
                            "--width", "320", "--height", "208", "--spp", "4", "--no-cpu", "--no-roofline"],
                           cwd=REPO, env=env, capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0, plain.stderr[-2000:]
    p = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][-1])
    assert p["rays_per_primary"] == d["rays_per_primary"]


@pytest.mark.gpu
def test_bench_two_ranks_control_flow_on_one_card():
    """bench.py as the driver launches it for N = 2, roofline leg included, with both ranks on cuda:0 and the gather
    through gloo (--rehearse-gloo): every rank must reach every collective (a rank-0-only frame would hang here), the line
    must report n_gpus 2 and the two shares must add up to the single-process frame's rays."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "2", "--warmup", "1", "--scene", "sample", "--width", "320", "--height", "208", "--spp", "4", "--no-cpu"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(REPO, "bench.py"), "--gpus", "2", "--rehearse-gloo"] + common
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "tiles%2" and "roofline" in d and "kernel_ms" in d
    plain = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--no-roofline"] + common,
                           cwd=REPO, env=env, capture_output=True, text=True, timeout=240)
    assert plain.returncode == 0, plain.stderr[-2000:]
    p = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][-1])
    assert d["rays"]["traced_per_frame"] == p["rays"]["traced_per_frame"]
    assert d["rays"]["reference_per_frame"] == p["rays"]["reference_per_frame"]
