#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per kernel: python scripts/pmc_summary.py DIR [DIR...]"""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            agg[k]["_vgpr"] = float(r["VGPR_Count"]); agg[k]["_lds"] = float(r["LDS_Block_Size"])
            if (d, r["Dispatch_Id"]) not in seen:
                seen.add((d, r["Dispatch_Id"]))
for k, v in sorted(agg.items()):
    if not k.startswith("k_"): continue
    print(k, {a: (int(b) if b == int(b) else b) for a, b in sorted(v.items())})
    g = v.get
    if g("SQ_INSTS_VALU") and g("SQ_THREAD_CYCLES_VALU"): print("   lanes/VALU inst: %.1f" % (g("SQ_THREAD_CYCLES_VALU") / g("SQ_INSTS_VALU")))
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY"): print("   wait_any %.2f  wait_inst %.2f  active_valu %.2f of wave cycles" % (g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_VALU", 0) / g("SQ_WAVE_CYCLES")))
    if g("SQ_INST_LEVEL_VMEM") and g("SQ_INSTS_VMEM"): print("   vmem latency ~%.0f cycles" % (g("SQ_INST_LEVEL_VMEM") / g("SQ_INSTS_VMEM")))
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0: print("   L2 hit rate %.3f" % (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
