#!/usr/bin/env python3
"""C5 (50 M triangles) as the bandwidth case: in-kernel record counts + the rocprofv3 passes of scripts/gpu_r3_c5.sh -> one JSON.

    python scripts/c5_summary.py gpurun_out/r3c5 > profiles/r03_c5_pmc.json

Per camera and for the traversal launch (k_trace): records fetched by the walks (counted in-kernel, 64 B each: what the lanes ask
the L1 for), bytes the L2 fetched from the memory side (FETCH_SIZE, KiB; doubled as MI355X_MICROARCH.md prescribes for gfx950 --
calibrated there for wide coalesced reads, so both figures are kept), L2 / L1 hit rates, and what that is per second of kernel time
against the 8 TB/s HBM peak.  The passes sum over the process: a small sizing frame (1/64 of the samples) rides along."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd._native import kernel_source_hash
d = sys.argv[1]
out = {"kernel_source_hash": kernel_source_hash(), "hbm_peak_GBs": 8000.0, "hbm_achievable_GBs": 6300.0,
       "source": "scripts/gpu_r3_c5.sh: rocprofv3 --pmc FETCH_SIZE | TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum | WRITE_SIZE, "
                 "one process per pass (python3 scripts/c5_frame.py directly behind --), in-kernel counters from a run without the profiler",
       "cameras": {}}
for cam in ("stress", "stress-diagonal"):
    pl = json.load(open(f"{d}/{cam}.plain.json"))
    fe = json.load(open(f"{d}/{cam}.fetch.json")).get("k_trace", {})
    ca = json.load(open(f"{d}/{cam}.cache.json")).get("k_trace", {})
    wr = json.load(open(f"{d}/{cam}.write.json")).get("k_trace", {})
    sec = pl["ms_trace"] / 1e3
    records = pl["nodes_visited"] + pl["tris_tested"] + pl["nodes_visited_any"] + pl["tris_tested_any"]
    rays = pl["rays_closest"] + pl["rays_shadow"]
    fetch_raw = fe.get("FETCH_SIZE", 0.0) * 1024.0
    hit, miss = ca.get("TCC_HIT_sum", 0.0), ca.get("TCC_MISS_sum", 0.0)
    l1_acc, l1_miss = ca.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0), ca.get("TCP_TCC_READ_REQ_sum", 0.0)
    out["cameras"][cam] = {
        "workload": f"{pl['scene']} {pl['width']}x{pl['height']} @{pl['spp']}spp depth 3", "triangles": pl["triangles"], "bvh_nodes": pl["bvh_nodes"],
        "tree_bytes": 64 * (pl["bvh_nodes"] + pl["triangles"]), "frame_ms": pl["ms_total"], "k_trace_ms": pl["ms_trace"], "k_shade_ms": pl["ms_shade"],
        "chunks": pl["chunks"], "rays": rays, "records_per_ray": records / rays,
        "node_fetch_bytes_per_lane_walks": 64.0 * records, "node_fetch_GBs_at_the_L1": 64.0 * records / sec / 1e9,
        "k_trace_fetch_bytes_raw": fetch_raw, "k_trace_hbm_read_bytes_x2": 2 * fetch_raw,
        "k_trace_write_bytes": wr.get("WRITE_SIZE", 0.0) * 1024.0,
        "hbm_read_GBs_raw": fetch_raw / sec / 1e9, "hbm_read_GBs_x2": 2 * fetch_raw / sec / 1e9,
        "hbm_frac_of_peak_raw": fetch_raw / sec / 1e9 / 8000.0, "hbm_frac_of_peak_x2": 2 * fetch_raw / sec / 1e9 / 8000.0,
        "l2_hit_rate": hit / max(hit + miss, 1.0), "l2_miss_bytes_at_128B": 128.0 * miss, "l1_hit_rate": 1.0 - l1_miss / max(l1_acc, 1.0),
        "dispatches_profiled": fe.get("dispatches"),
    }
json.dump(out, sys.stdout, indent=1)
