#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage (no GPU needed).
usage: python scripts/kernel_resources.py [extra hipcc flags...]"""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-c",
       "rt_kernels.hip", "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
out = subprocess.run(cmd, cwd=os.path.join(REPO, "raytracer_amd", "csrc"), capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +(?:Function Name|Name): (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-28s %5s %5s %7s %6s %6s %5s" % ("kernel", "VGPR", "SGPR", "scratch", "vspill", "sspill", "occ"))
for k, v in rows.items():
    print("%-28s %5d %5d %7d %6d %6d %5d" % (k[:28], v.get("VGPRs", 0), v.get("TotalSGPRs", 0), v.get("ScratchSize", 0),
                                          v.get("VGPRs Spill", 0), v.get("SGPRs Spill", 0), v.get("Occupancy", 0)))
