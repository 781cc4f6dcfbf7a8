#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native trace loop.

Workload (BASELINE.json metric, configs[3]): the competition scene (main.cpp:409-536 minus the missing pietre.obj,
68 202 triangles + 2 planes + 2 spheres, 4 lights) at 3840x2160, 64 spp, bounce depth 3.  One "step" = one frame.
Scene, BVH and queues are resident in HBM before the timed region; the timed region covers primary-ray generation,
every trace/shade level, the resolve and (N > 1) the RCCL framebuffer gather + assembly on rank 0.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver with torch.distributed.run (one process per GPU): the frame's 16x16 tiles are
interleaved over the ranks (tile t -> rank t % N), scene + BVH replicated, one gather at frame end ("strong"
scaling: the frame is fixed).  Rank 0 prints ONE JSON line.

`roofline` is reported for the dominant kernel (k_closest or k_any, whichever took longer): algorithmic bytes
(64 B per 4-wide BVH node or triangle record fetched + the ray's queue record, counted in-kernel) over its summed launch time
measured with HIP events on the launch stream, against 8 TB/s HBM.  The working set of this scene is cache resident,
so the fraction is expected to be small; `roofline_valu` prices the same kernel against the FP32 vector peak.
`cpu_baseline` is the compiled reference (oracle/_ref, `g++ -Ofast -fopenmp`) on a bounded sample of the same scene.
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TFLOPS = 157.3     # same guide: peak FP32 vector


def cpu_baseline(scene_name, sample_w, sample_h):
    """reference (-Ofast -fopenmp) on a bounded sample of the workload; rays counted by the parity-checked port"""
    from oracle.binding import Oracle, REF_OFAST
    from raytracer_amd import FlatScene
    scene = FlatScene.builtin(scene_name)
    ref = Oracle(scene).render(sample_w, sample_h, spp=4, max_depth=3)
    rays = ref["stats"]["rays_closest"] + ref["stats"]["rays_shadow"]
    port_s = ref["stats"]["ms_total"] / 1e3
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    sample = f"{scene_name} {sample_w}x{sample_h} @4spp depth 3 ({rays} rays)"
    if os.path.exists(REF_OFAST):
        out = subprocess.run([REF_OFAST, "render", scene_name, str(sample_w), str(sample_h), "4", "3", "0", "-"],
                             cwd=os.path.join(REPO, "assets"), capture_output=True, text=True, check=True).stdout
        info = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        return {"value": rays / info["render_s"] / 1e6, "unit": "Mrays/s", "cores": info["threads"], "kind": "reference",
                "sample": sample + ", reference built g++ -Ofast -fopenmp", "seconds": info["render_s"],
                "primary_samples_per_s": sample_w * sample_h * 4 / info["render_s"],
                "port_mrays_per_s": rays / port_s / 1e6}
    return {"value": rays / port_s / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": sample + ", oracle/rt_oracle.cpp -O2 -fopenmp", "seconds": port_s,
            "primary_samples_per_s": sample_w * sample_h * 4 / port_s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="competition")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--chunk", type=int, default=0, help="primary samples per wavefront chunk (0 = library default)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-roofline", action="store_true", help="skip the two instrumented passes")
    ap.add_argument("--cpu-sample", default="480x360")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the gather even with one rank (rehearses the RCCL path on a 1-GPU box)")
    a = ap.parse_args()

    import numpy as np
    import torch
    from raytracer_amd import FlatScene
    from raytracer_amd.renderer import Renderer

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
    dist = None
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    scene = FlatScene.builtin(a.scene)
    r = Renderer(local_rank).set_scene(scene)
    W, H = a.width, a.height
    from raytracer_amd.dist import DistributedFrame, tile_counts
    n_tiles = tile_counts(W, H, 0, 1)[1]
    df = DistributedFrame(r, W, H, rank, world, dev, dist, force_collective=a.force_dist)   # tile-major device buffers owned by torch
    stream = torch.cuda.current_stream().cuda_stream

    def frame(collect=0):
        # own tiles -> RCCL gather of the tile-major buffers over xGMI -> de-interleave on rank 0
        return df.render(stream=stream, spp=a.spp, max_depth=a.depth, chunk_samples=a.chunk, collect_stats=collect)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        frame()
    sync()
    t0 = time.perf_counter()
    rays = 0
    prim = 0
    for _ in range(a.steps):
        st = frame()
        rays += st["rays_closest"] + st["rays_shadow"]
        prim += st["primary_samples"]
    sync()
    dt = time.perf_counter() - t0
    tot = torch.tensor([dt, float(rays), float(prim)], dtype=torch.float64, device=dev)
    if use_dist:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(mx[0])
    rays_all, prim_all = float(tot[1]), float(tot[2])

    result = None
    if rank == 0:
        result = {
            "metric": "Mrays/s", "value": rays_all / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / max(a.steps, 1) * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "reference scene assets (meshes/*.obj), synthetic camera samples",
            "config": {"workload": f"{a.scene} scene {W}x{H} @{a.spp}spp depth {a.depth}", "triangles": scene.n_triangles,
                       "tiles": n_tiles, "parallelism": f"tiles%{world}", "chunk_samples": a.chunk or (1 << 26)},
            "primary_samples_per_s": prim_all / dt, "rays_per_primary": rays_all / max(prim_all, 1.0),
            "ms_per_frame": dt / max(a.steps, 1) * 1e3, "bvh": {"nodes": st["bvh_nodes"], "depth": st["bvh_depth"]},
        }
    if rank == 0 and not a.no_roofline:
        timed = frame(collect=2)       # per-kernel-class HIP-event timing on the launch stream
        counted = frame(collect=1)     # in-kernel record counters (stats build of the same kernels, counts only)
        pc, pa = counted["prof_packet_closest"], counted["prof_packet_any"]
        # 64-byte records fetched: per-lane kernels count one per lane step, packet kernels one 128-byte node (two
        # records) or one triangle record per WAVE step
        kinds = {
            "k_closest": (timed["ms_closest"], timed["launches_closest"], counted["nodes_visited"] + 2 * pc[1],
                          counted["tris_tested"] + pc[3], counted["rays_closest"], 40.0),
            "k_any": (timed["ms_any"], timed["launches_any"], counted["nodes_visited_any"] + 2 * pa[1],
                      counted["tris_tested_any"] + pa[3], counted["rays_shadow"], 48.0),
        }
        name = max(kinds, key=lambda k: kinds[k][0])
        ms, launches, nodes, tris, nrays, ray_bytes = kinds[name]
        launches = max(launches, 1)
        abytes = 64.0 * nodes + 64.0 * tris + ray_bytes * nrays   # algorithmic bytes per ray
        flops = 80.0 * nodes + 80.0 * tris                         # four slab tests per node, SURVEY.md 8(d)
        ach = abytes / (ms / 1e3) / 1e9 if ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(REPO, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same workload
            t = json.load(open(tpath))
            if t.get("workload") == result["config"]["workload"] and name in t.get("kernels", {}):
                traffic = t["kernels"][name]["hbm_bytes_per_launch"]
        result["roofline"] = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                              "bytes_per_launch": abytes / launches, "avg_launch_ms": ms / launches, "launches": launches,
                              "records_per_ray": (nodes + tris) / max(nrays, 1),
                              "note": "scene records are cache resident: achieved counts algorithmic record fetches, "
                                      "traffic the HBM side (queues); see roofline_valu for the binding ceiling"}
        tf = flops / (ms / 1e3) / 1e12 if ms > 0 else 0.0
        result["roofline_valu"] = {"kernel": name, "bound": "valu_fp32", "achieved": tf, "peak": VALU_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": tf / VALU_PEAK_TFLOPS,
                                   "note": "80 flop per 4-wide node (4 slab tests) + 80 per triangle test (SURVEY.md 8(d))"}
        pr = counted["prof_any" if name == "k_any" else "prof_closest"]  # counted in-kernel (cheap statistics build)
        if pr[2] and pr[5] and pr[8]:
            result["roofline_valu"]["lanes"] = {
                "lanes_per_node_phase": pr[3] / pr[2], "lanes_per_triangle_phase": pr[6] / pr[5],
                "node_phases": pr[2], "triangle_phases": pr[5], "rays_per_refill": pr[9] / pr[8],
                "note": "of 64 lanes; a phase issues its instructions for the whole wave"}
        vpath = os.path.join(REPO, "profiles", "r01_pmc_valu.json")
        if os.path.exists(vpath):  # committed rocprofv3 --pmc SQ_INSTS_VALU pass of this workload + the opcode micro-benchmark
            v = json.load(open(vpath)).get("kernels", {}).get(name)
            if v and v.get("SQ_INSTS_VALU"):
                per_ray = v["SQ_INSTS_VALU"] / max(nrays, 1)
                result["roofline_valu"]["issue"] = {
                    "valu_wave_insts_per_ray": per_ray,
                    "simd_cycles_per_valu_inst": (ms / 1e3) * 2.4e9 * 1024 / (per_ray * nrays),
                    "note": "instruction count per ray from profiles/r01_pmc_valu.json x this run's time, 1024 SIMDs at 2.4 GHz; "
                            "the node step's opcode mix costs ~3.7 cycles per instruction (profiles/r01_valu_issue.md): the "
                            "closer this figure is to 3.7, the closer the kernel is to the SIMDs' issue limit"}
        result["kernel_ms"] = {k: timed[k] for k in ("ms_gen", "ms_closest", "ms_shade", "ms_any", "ms_resolve", "ms_total")}
    if rank == 0 and not a.no_cpu and world == 1:
        try:
            sw, sh = (int(x) for x in a.cpu_sample.split("x"))
            result["cpu_baseline"] = cpu_baseline(a.scene, sw, sh)
        except Exception as e:  # the baseline is a reported number, not a dependency of the measurement
            result["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "unavailable", "sample": repr(e)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
