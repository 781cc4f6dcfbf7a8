#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native trace loop.

Workload (BASELINE.json metric, configs[3]): the competition scene (main.cpp:409-536 minus the missing pietre.obj,
68 202 triangles + 2 planes + 2 spheres, 4 lights) at 3840x2160, 64 spp, bounce depth 3.  One "step" = one frame.
`value` = closest-hit + shadow queries actually traced per second; the reference-defined ray count of the same frame
(which includes shadow rays that this implementation proves irrelevant and does not trace) is reported under `rays`.
Scene, BVH and queues are resident in HBM before the timed region; the timed region covers primary-ray generation,
every trace/shade level, the resolve and (N > 1) the RCCL framebuffer gather + assembly on rank 0.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver with torch.distributed.run (one process per GPU): the frame's 16x16 tiles are
interleaved over the ranks (tile t -> rank t % N), scene + BVH replicated, one gather at frame end ("strong"
scaling: the frame is fixed).  Rank 0 prints ONE JSON line.

`roofline` is reported for the dominant kernel (k_closest or k_any, whichever took longer).  Nothing on this path is a
dense contraction and the scene's records are cache resident, so the binding ceiling is the FP32 vector unit
(157.3 TFLOP/s): flops = 80 per 4-wide node step + 80 per triangle test, counted in-kernel in this run, over the kernel's
summed launch time measured with HIP events on the launch stream.  Its HBM side (queue records + accumulator updates)
sits next to it under `roofline.hbm`; `roofline.traffic` is only filled from a committed rocprofv3 --pmc pass when that
pass was collected on the same kernel sources (hash).
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
                "build": "g++ -std=c++17 -Ofast -fopenmp oracle/ref_harness.cpp (includes /root/reference/main.cpp; oracle/Makefile)",
                "sample": sample + ", reference built g++ -Ofast -fopenmp", "seconds": info["render_s"],
                "primary_samples_per_s": sample_w * sample_h * 4 / info["render_s"],
                "port_mrays_per_s": rays / port_s / 1e6}
    return {"value": rays / port_s / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "build": "g++ -O2 -fopenmp oracle/rt_oracle.cpp (oracle/Makefile)",
            "sample": sample + ", oracle/rt_oracle.cpp -O2 -fopenmp", "seconds": port_s,
            "primary_samples_per_s": sample_w * sample_h * 4 / port_s}


def multi_abi(a):
    """--multi-abi: the frame over N GPUs of ONE process through rt_multi_* (one host thread per GPU inside the library, RCCL
    send/recv at frame end).  `value` covers rt_multi_render as a C++ host calls it: it ends with the images in HOST buffers."""
    from raytracer_amd import FlatScene
    from raytracer_amd.renderer import MultiRenderer
    devices = [int(x) for x in a.multi_devices.split(",")] if a.multi_devices else list(range(a.gpus))
    scene = FlatScene.builtin(a.scene)
    m = MultiRenderer(devices, rehearse_rccl=len(devices) == 1).set_scene(scene)
    W, H = a.width, a.height
    kw = dict(spp=a.spp, max_depth=a.depth, chunk_samples=a.chunk, want=("rgb8", "primid"))
    for _ in range(a.warmup):
        m.render(W, H, **kw)
    t0 = time.perf_counter()
    rays = zero = prim = 0
    for _ in range(a.steps):
        out = m.render(W, H, **kw)
        for st in out["stats"]:
            rays += st["rays_closest"] + st["rays_shadow"]
            zero += st["shadow_zero"]
            prim += st["primary_samples"]
    dt = time.perf_counter() - t0
    result = {
        "metric": "Mrays/s", "value": rays / dt / 1e6, "unit": "Mrays/s", "n_gpus": len(devices), "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / max(a.steps, 1) * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "reference scene assets (meshes/*.obj), synthetic camera samples",
        "config": {"workload": f"{a.scene} scene {W}x{H} @{a.spp}spp depth {a.depth}", "triangles": scene.n_triangles,
                   "parallelism": f"tiles%{len(devices)}", "devices": devices},
        "primary_samples_per_s": prim / dt, "rays_per_primary": rays / max(prim, 1),
        "multi": dict(out["timings"], path="rt_multi_* (C ABI, one process, one host thread per GPU), transport " + m.transport,
                      share_ms=[st["ms_total"] for st in out["stats"]], chunks_per_share=[st["chunks"] for st in out["stats"]],
                      note="timings of the last frame; ms_per_step includes the read-back of the images to host buffers (ms_readback)"),
    }
    print(json.dumps(result), flush=True)
    m.close()


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
    ap.add_argument("--sizing-frame", action="store_true",
                    help="render one frame at 1/8 of the width and height first (untimed): the library then knows the scene's fan-out "
                         "and plans the full frames like any later frame.  For profiler passes that sum over the process "
                         "(--steps 1 --warmup 0 --sizing-frame: 1.6 %% of the samples ride along instead of a worst-case first frame)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the gather even with one rank (rehearses the RCCL path on a 1-GPU box)")
    ap.add_argument("--multi-abi", action="store_true",
                    help="ONE process, N GPUs through the C ABI's rt_multi_* (what a C++ host links) instead of one process per GPU: "
                         "--gpus N uses devices 0..N-1 (or --multi-devices)")
    ap.add_argument("--multi-devices", default="", help="with --multi-abi: comma-separated device ids; a repeated id rehearses "
                                                         "several ranks on one card (exchange by device copies, not RCCL)")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N ranks on ONE card: every rank renders on cuda:0 and the gather goes through host tensors and gloo "
                         "(RCCL wants one GPU per rank).  Checks the control flow of the N > 1 path on a 1-GPU box, not its speed")
    a = ap.parse_args()

    import numpy as np
    import torch
    from raytracer_amd import FlatScene
    from raytracer_amd.renderer import Renderer

    if a.multi_abi:
        return multi_abi(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
    dist = None
    if a.rehearse_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        import torch.distributed as dist
        if a.rehearse_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    scene = FlatScene.builtin(a.scene)
    r = Renderer(local_rank).set_scene(scene)
    W, H = a.width, a.height
    from raytracer_amd.dist import DistributedFrame, tile_counts
    n_tiles = tile_counts(W, H, 0, 1)[1]
    df = DistributedFrame(r, W, H, rank, world, dev, dist, force_collective=a.force_dist,
                          stage_host=a.rehearse_gloo)   # tile-major device buffers owned by torch
    # one explicit stream for render, gather and assembly: torch's legacy default stream has handle 0, which the C ABI reads
    # as "the context's own stream" -- the RCCL gather (ordered against torch's CURRENT stream) would then not be ordered
    # against the render / assembly kernels
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    stream = side.cuda_stream

    def frame(collect=0):
        # own tiles -> RCCL gather of the tile-major buffers over xGMI -> de-interleave on rank 0
        return df.render(stream=stream, spp=a.spp, max_depth=a.depth, chunk_samples=a.chunk, collect_stats=collect)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if a.sizing_frame:
        # (with the counting build of the kernels: their launches carry other names in a profiler's kernel table, which then
        # holds full-size launches only under the plain names)
        r.render(max(W // 8, 16), max(H // 8, 16), spp=a.spp, max_depth=a.depth, collect_stats=1, want=())
    for _ in range(a.warmup):
        frame()
    sync()
    t0 = time.perf_counter()
    rays = 0
    prim = 0
    zero = 0
    for _ in range(a.steps):
        st = frame()
        rays += st["rays_closest"] + st["rays_shadow"]   # queries actually traced
        zero += st["shadow_zero"]                         # shadow rays of the reference that are provably irrelevant (not traced)
        prim += st["primary_samples"]
    sync()
    dt = time.perf_counter() - t0
    tot = torch.tensor([dt, float(rays), float(prim), float(zero)], dtype=torch.float64, device="cpu" if a.rehearse_gloo else dev)
    if use_dist:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(mx[0])
    rays_all, prim_all, zero_all = float(tot[1]), float(tot[2]), float(tot[3])

    # one more frame with a synchronisation behind the gather and behind the assembly: where a frame's time goes per rank
    multi = None
    if use_dist:
        st_parts = df.render(stream=stream, spp=a.spp, max_depth=a.depth, chunk_samples=a.chunk, time_parts=True)
        mine = [df.timings["share_ms"], df.timings["gather_ms"], df.timings["assemble_ms"], float(st_parts["chunks"])]
        parts = [None] * world
        dist.all_gather_object(parts, mine)
        multi = {"path": "one process per GPU (torch.distributed, backend %s)" % ("gloo, host-staged" if a.rehearse_gloo else "nccl = RCCL"),
                 "share_ms": [x[0] for x in parts], "gather_ms_rank0": parts[0][1], "assemble_ms_rank0": parts[0][2],
                 "chunks_per_share": [int(x[3]) for x in parts],
                 "note": "share_ms: GPU time of each rank's tiles (HIP events); gather: host wall time from the end of rank 0's render to "
                         "the end of the gather on rank 0, i.e. it includes waiting for the slowest rank"}
    result = None
    if rank == 0:
        result = {
            "metric": "Mrays/s", "value": rays_all / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / max(a.steps, 1) * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "reference scene assets (meshes/*.obj), synthetic camera samples",
            "config": {"workload": f"{a.scene} scene {W}x{H} @{a.spp}spp depth {a.depth}", "triangles": scene.n_triangles,
                       "tiles": n_tiles, "parallelism": f"tiles%{world}",
                       "chunk_samples": a.chunk or "as many as the workspace holds (fan-out observed on the scene; DESIGN.md 2)",
                       "workspace_gb": float(os.environ.get("RT_WORKSPACE_GB", "30")),
                       "chunks_per_frame": st["chunks"], "chunk_retries_last_frame": st["chunk_retries"]},
            "primary_samples_per_s": prim_all / dt, "rays_per_primary": rays_all / max(prim_all, 1.0),
            # `value` counts the queries that were TRACED.  The reference casts more: shadow rays whose pending radiance is
            # exactly zero (main.cpp:172 with refraction 1) are not emitted here (bit-identical image; rt_stats.shadow_zero)
            "rays": {"traced_per_frame": rays_all / max(a.steps, 1), "reference_per_frame": (rays_all + zero_all) / max(a.steps, 1),
                     "shadow_rays_not_traced_per_frame": zero_all / max(a.steps, 1),
                     "reference_defined_mrays_per_s": (rays_all + zero_all) / dt / 1e6},
            "ms_per_frame": dt / max(a.steps, 1) * 1e3, "bvh": {"nodes": st["bvh_nodes"], "depth": st["bvh_depth"]},
        }
        if multi:
            result["multi"] = multi
    if not a.no_roofline:
        # EVERY rank renders the two instrumented frames: a frame ends in a collective (the gather), so rank 0 alone
        # would wait for peers that have moved on.  The figures reported are rank 0's (its share of the tiles).
        lanes_env = os.environ.get("RT_LANES")
        os.environ["RT_LANES"] = "1"   # one pipeline: the kernels run one after the other, a launch's duration is its own
        timed = frame(collect=2)       # per-kernel-class HIP-event timing on the launch stream
        counted = frame(collect=1)     # in-kernel record counters (stats build of the same kernels, counts only)
        if lanes_env is None:
            del os.environ["RT_LANES"]
        else:
            os.environ["RT_LANES"] = lanes_env
    if rank == 0 and not a.no_roofline:
        from raytracer_amd._native import kernel_source_hash
        pc, pa = counted["prof_packet_closest"], counted["prof_packet_any"]
        # records fetched, counted in-kernel in THIS run: (ms, launches, node steps, triangle steps, rays)
        kinds = {
            "k_closest": (timed["ms_closest"], timed["launches_closest"], counted["nodes_visited"], counted["tris_tested"], counted["rays_closest"]),
            "k_any": (timed["ms_any"], timed["launches_any"], counted["nodes_visited_any"], counted["tris_tested_any"], counted["rays_shadow"]),
            # the fused traversal launch (one per iteration): shadow rays + bounce rays per lane, primary rays as packets
            # (a packet step is counted once per lane that takes part in it: prof_packet[2] / [4])
            # (T iterations walk T ray queues and T shadow queues in T + 1 launches: the first has no shadow rays yet, the last no bounce rays any more)
            "k_trace": (timed["ms_trace"], max(timed["launches_closest"], timed["launches_any"]) + 1,
                        counted["nodes_visited"] + counted["nodes_visited_any"] + pc[2],
                        counted["tris_tested"] + counted["tris_tested_any"] + pc[4], counted["rays_closest"] + counted["rays_shadow"]),
        }
        name = max(kinds, key=lambda k: kinds[k][0])
        ms, launches, nodes, tris, nrays = kinds[name]
        launches = max(launches, 1)
        # 20 flop per slab test x 4 child boxes of a 4-wide node, 80 per reference-formulation triangle test
        flops = 80.0 * nodes + 80.0 * tris
        tf = flops / (ms / 1e3) / 1e12 if ms > 0 else 0.0
        # the HBM side of the same kernel: its records are cache resident (5.4 MB scene), what streams is the queue --
        # k_any: 16 B shadow record per ray + (16 B pending colour + 3 x 8 B accumulator update) per ray that reaches its
        # light; k_closest: 32 B ray record + 8 B hit record per ray
        if name == "k_any":
            qbytes = 16.0 * nrays + 40.0 * counted["shadow_unoccluded"]
        elif name == "k_trace":
            qbytes = 16.0 * counted["rays_shadow"] + 40.0 * counted["shadow_unoccluded"] + 40.0 * counted["rays_closest"]
        else:
            qbytes = 40.0 * nrays
        khash = kernel_source_hash()
        traffic, traffic_src = None, "no rocprofv3 --pmc pass of these kernel sources is committed"
        tpath = os.path.join(REPO, "profiles", "r03_pmc_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from the committed FETCH_SIZE / WRITE_SIZE passes of this workload
            t = json.load(open(tpath))
            if t.get("kernel_source_hash") != khash:
                traffic_src = "profiles/r03_pmc_traffic.json was collected on other kernel sources (hash mismatch): not used"
            elif t.get("workload") == result["config"]["workload"] and name in t.get("kernels", {}):
                # per launch of THIS run (the profiled frame may have been cut into more chunks, i.e. more launches)
                traffic = t["kernels"][name]["hbm_bytes_per_frame"] / launches
                traffic_src = "profiles/r03_pmc_traffic.json (same kernel sources, hash %s)" % khash
        rec_bytes = 64.0 * (nodes + tris)   # what the walks ask the memory system for: one 64-byte record per node step / triangle test
        result["roofline"] = {
            "kernel": name, "bound": "valu_fp32", "achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tf / VALU_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
            "flops_per_launch": flops / launches, "avg_launch_ms": ms / launches, "launches": launches,
            "records_per_ray": (nodes + tris) / max(nrays, 1), "kernel_source_hash": khash,
            "hbm": {"algorithmic_bytes_per_launch": qbytes / launches, "achieved_GBs": qbytes / (ms / 1e3) / 1e9 if ms > 0 else 0.0,
                    "peak_GBs": HBM_PEAK_GBS, "frac": qbytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0,
                    "note": "queue records + accumulator updates only; BVH / triangle records are L2 / Infinity Cache resident"},
            "note": "no dense contraction on this path (SURVEY.md 8d): the ceiling is the FP32 vector unit; flops = 80 per 4-wide "
                    "node step + 80 per triangle test, both counted in-kernel in this run"}
        if a.scene.startswith("stress"):
            # the 50 M-triangle tree (3.9 GB of records) is the configuration whose record fetches leave the
            # caches.  Bound = HBM: `achieved` = record bytes the walks fetch (64 B each, counted in-kernel in this run) per
            # second of kernel time; `traffic` = what the L2 really fetched from the memory side for this kernel, from the
            # committed rocprofv3 FETCH_SIZE pass of this scene and camera (profiles/r03_c5_pmc.json), per launch.
            fl = result["roofline"]
            gbs = rec_bytes / (ms / 1e3) / 1e9 if ms > 0 else 0.0
            fl["valu_fp32"] = {"achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / VALU_PEAK_TFLOPS}
            fl.update({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                       "record_bytes_per_launch": rec_bytes / launches,
                       "note": "achieved = 64-byte records fetched by the walks (node steps + triangle tests, counted in-kernel in this run) / kernel "
                               "time: an upper bound on what reaches HBM (L1 / L2 / Infinity Cache hits included); `traffic` and `l2` say "
                               "how much of it did"})
            ppath = os.path.join(REPO, "profiles", "r03_c5_pmc.json")
            if os.path.exists(ppath):
                pj = json.load(open(ppath))
                cam = pj.get("cameras", {}).get(a.scene.split(":")[0])
                if cam:
                    per_ray = cam["k_trace_hbm_read_bytes_x2"] / max(cam["rays"], 1)   # the profiled frame may differ in size: per ray
                    fl["traffic"] = per_ray * nrays / launches
                    fl["traffic_source"] = ("profiles/r03_c5_pmc.json (%s; HBM read bytes per traced ray x this run's rays; kernel sources %s)"
                                            % (cam["workload"], "match" if pj.get("kernel_source_hash") == khash else "DIFFER from this build"))
                    fl["hbm_measured"] = {"read_GBs": cam["hbm_read_GBs_x2"], "frac_of_peak": cam["hbm_frac_of_peak_x2"],
                                          "read_GBs_raw_counter": cam["hbm_read_GBs_raw"]}
                    fl["l2"] = {"hit_rate": cam["l2_hit_rate"], "l1_hit_rate": cam["l1_hit_rate"], "source": "profiles/r03_c5_pmc.json"}
        pr = counted["prof_any" if name == "k_any" else "prof_closest"]  # counted in-kernel (cheap statistics build; k_trace: its closest-hit phase)
        if pr[2] and pr[5] and pr[8]:
            result["roofline"]["lanes"] = {
                "lanes_per_node_phase": pr[3] / pr[2], "lanes_per_triangle_phase": pr[6] / pr[5],
                "node_phases": pr[2], "triangle_phases": pr[5], "rays_per_refill": pr[9] / pr[8],
                "note": "of 64 lanes; a phase issues its instructions for the whole wave"}
        cpath = os.path.join(REPO, "profiles", "r03_pmc_cache.json")
        if os.path.exists(cpath):  # committed L1 / L2 hit-rate passes of this workload: only with matching kernel sources
            cj = json.load(open(cpath))
            c = cj.get("kernels", {}).get(name)
            if cj.get("kernel_source_hash") == khash and c and cj.get("workload") == result["config"]["workload"]:
                result["roofline"]["l2"] = dict(c, source="profiles/r03_pmc_cache.json (same kernel sources)")
        vpath = os.path.join(REPO, "profiles", "r03_pmc_valu.json")
        if os.path.exists(vpath):  # committed rocprofv3 --pmc SQ pass of this workload: only with matching kernel sources
            vj = json.load(open(vpath))
            v = vj.get("kernels", {}).get(name)
            if vj.get("kernel_source_hash") == khash and v and v.get("SQ_INSTS_VALU") and vj.get("workload", "").startswith(result["config"]["workload"]):
                result["roofline"]["issue"] = {
                    "valu_wave_insts_per_ray": v["SQ_INSTS_VALU"] / max(nrays, 1),
                    "valu_issue_frac_of_2cyc_peak": v.get("valu_issue_frac"), "sclk_ghz_profiled": v.get("sclk_ghz"),
                    "source": "profiles/r03_pmc_valu.json (same kernel sources)"}
        result["kernel_ms"] = {k: timed[k] for k in ("ms_gen", "ms_trace", "ms_packet", "ms_closest", "ms_shade", "ms_any", "ms_resolve", "ms_total")}
    if rank == 0 and not a.no_cpu and world == 1:
        try:
            sw, sh = (int(x) for x in a.cpu_sample.split("x"))
            result["cpu_baseline"] = cpu_baseline(a.scene, sw, sh)
        except Exception as e:  # the baseline is a reported number, not a dependency of the measurement
            result["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "unavailable", "sample": repr(e)}
        if result["cpu_baseline"].get("kind") == "port":
            result["cpu_baseline"]["warning"] = ("oracle/_ref (the compiled reference, built by oracle/Makefile where /root/reference "
                                                 "exists) is ABSENT: this is the oracle PORT, which walks one BVH over all triangles and is "
                                                 "3-4x faster than the reference -- not comparable with a kind == 'reference' figure")
            print("bench.py: " + result["cpu_baseline"]["warning"], file=sys.stderr, flush=True)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
