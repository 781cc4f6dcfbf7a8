#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; both in KiB) of one bench.py frame.

    python scripts/pmc_traffic.py FETCH_DIR WRITE_DIR "workload string" > profiles/r02_pmc_traffic.json

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte requests as 64 bytes for wide coalesced
reads, so it is doubled; WRITE_SIZE is taken as is.  Both raw values are kept in the output."""
import re, collections, csv, glob, json, sys


def collect(d, counter):
    tot, calls = collections.Counter(), collections.Counter()
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            full = r["Kernel_Name"].split("(")[0].replace("void ", "")
            k = full.split("<")[0]
            if k == "k_any" and re.match(r"k_any<\d+, 2\b", full) is not None:
                k = "k_any_split"  # follow-up rounds over the items of split walks (usually empty)
            tot[k] += float(r["Counter_Value"])
            calls[k] += 1
    return tot, calls


fetch, calls = collect(sys.argv[1], "FETCH_SIZE")
write, _ = collect(sys.argv[2], "WRITE_SIZE")
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd._native import kernel_source_hash
out = {"workload": sys.argv[3], "kernel_source_hash": kernel_source_hash(), "unit": "bytes", "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950)",
       "kernels": {}}
for k in sorted(fetch):
    if not k.startswith("k_"):
        continue
    n = max(calls[k], 1)
    f_raw, w_raw = fetch[k] * 1024.0, write.get(k, 0.0) * 1024.0
    out["kernels"][k] = {"launches": n, "fetch_bytes_raw": f_raw, "write_bytes_raw": w_raw,
                         "hbm_bytes_per_frame": 2 * f_raw + w_raw, "hbm_bytes_per_launch": (2 * f_raw + w_raw) / n}
json.dump(out, sys.stdout, indent=1)
