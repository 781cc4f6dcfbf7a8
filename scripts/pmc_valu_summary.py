#!/usr/bin/env python3
"""Summarises scripts/gpu_pmc_valu.sh: per kernel the VALU wave instructions, the clock held and the fraction of the
SIMDs' VALU issue capacity used.  Capacity: MI355X_MICROARCH.md gives 2 cycles per wave64 VALU instruction per SIMD
(4 for one wave alone) => 0.5 instructions per cycle per SIMD, 1024 SIMDs.  Writes DIR/pmc_valu.json, prints a table.
The clock comes from the counters, not from a nominal figure: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel time
of the SAME pass (the pass's own kernel trace)."""
import re, collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd._native import kernel_source_hash

d = sys.argv[1]
SIMDS = 1024


def key(name):
    full = name.split("(")[0].replace("void ", "")
    k = full.split("<")[0]
    return "k_any_split" if k == "k_any" and re.match(r"k_any<\d+, 2\b", full) is not None else k


def durations(sub):
    dur = collections.Counter()
    for f in glob.glob(f"{d}/{sub}/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[key(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    if not dur:  # a counter pass without a kernel trace: the counter rows carry the dispatch's timestamps
        seen = set()
        for f in glob.glob(f"{d}/{sub}/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "Start_Timestamp" not in r or (f, r["Dispatch_Id"]) in seen:
                    continue
                seen.add((f, r["Dispatch_Id"]))
                dur[key(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return dur


def counters(sub):
    cnt = collections.defaultdict(collections.Counter)
    for f in glob.glob(f"{d}/{sub}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[key(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return cnt


plain = durations("trace")
out = {"workload": "competition scene 3840x2160 @64spp depth 3, one frame", "kernel_source_hash": kernel_source_hash(),
       "source": "rocprofv3 --pmc, two SQ passes (scripts/gpu_pmc_valu.sh); seconds = un-profiled kernel trace", "kernels": {}}
passes = [(counters("sq"), durations("sq")), (counters("sq2"), durations("sq2"))]
for k in sorted(plain):
    if not k.startswith("k_") or plain[k] < 1e-3:
        continue
    e = {"seconds": plain[k]}
    for cnt, dur in passes:
        c = cnt.get(k, {})
        for name, v in c.items():
            if name in ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES") and name in e:
                continue
            e[name] = v
        if "GRBM_GUI_ACTIVE" in c and dur.get(k) and "sclk_ghz" not in e:
            e["seconds_profiled"] = dur[k]
            e["sclk_ghz"] = c["GRBM_GUI_ACTIVE"] / 8 / dur[k] / 1e9
    if e.get("SQ_INSTS_VALU") and e.get("sclk_ghz"):
        simd_cycles = e["seconds_profiled"] * e["sclk_ghz"] * 1e9 * SIMDS
        e["simd_cycles_per_valu_inst"] = simd_cycles / e["SQ_INSTS_VALU"]
        e["valu_issue_frac"] = 2.0 * e["SQ_INSTS_VALU"] / simd_cycles          # of the 2-cycles-per-instruction peak
        e["salu_per_valu"] = e.get("SQ_INSTS_SALU", 0) / e["SQ_INSTS_VALU"]
    if e.get("SQ_WAVE_CYCLES"):
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if n in e:
                e["frac_of_wave_cycles_" + n[3:].lower()] = e[n] / e["SQ_WAVE_CYCLES"]
    if e.get("SQ_THREAD_CYCLES_VALU") and e.get("SQ_ACTIVE_INST_VALU"):
        e["lanes_per_valu_inst"] = e["SQ_THREAD_CYCLES_VALU"] / e["SQ_ACTIVE_INST_VALU"]
    out["kernels"][k] = e
json.dump(out, open(f"{d}/pmc_valu.json", "w"), indent=1)
print("%-12s %8s %7s %10s %9s %9s %9s %8s %8s %8s" % ("kernel", "ms", "sclk", "VALU inst", "cyc/inst", "issue %", "SALU/VALU", "wait_any", "wait_inst", "lanes"))
for k, e in out["kernels"].items():
    print("%-12s %8.1f %7.2f %10.3g %9.2f %9.1f %9.2f %8.2f %8.2f %8.1f" % (
        k, e["seconds"] * 1e3, e.get("sclk_ghz", 0), e.get("SQ_INSTS_VALU", 0), e.get("simd_cycles_per_valu_inst", 0),
        100 * e.get("valu_issue_frac", 0), e.get("salu_per_valu", 0), e.get("frac_of_wave_cycles_wait_any", 0),
        e.get("frac_of_wave_cycles_wait_inst_any", 0), e.get("lanes_per_valu_inst", 0)))
