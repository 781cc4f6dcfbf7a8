# SQ instruction counters of one bench.py frame (competition 4K/64spp) -> gpurun_out/$1/pmc_valu.json
tag=${1:-pmc_valu}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout -s KILL 300 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/$tag/trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-roofline > gpurun_out/$tag/trace.log 2>&1 || { tail -5 gpurun_out/$tag/trace.log; exit 1; }
timeout -s KILL 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d gpurun_out/$tag/sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-roofline > gpurun_out/$tag/sq.log 2>&1 || { tail -5 gpurun_out/$tag/sq.log; exit 1; }
python3 - <<PY
import collections, csv, glob, json
def key(name):
    full = name.split("(")[0].replace("void ", "")
    k = full.split("<")[0]
    return "k_any_split" if k == "k_any" and full.rstrip().endswith(", 2>") else k
cnt = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob("gpurun_out/$tag/sq/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[key(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
dur = collections.Counter()
for f in glob.glob("gpurun_out/$tag/trace/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[key(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
out = {"workload": "competition scene 3840x2160 @64spp depth 3, one frame", "kernels": {}}
for k, c in cnt.items():
    if not k.startswith("k_"): continue
    out["kernels"][k] = dict(c, seconds=dur.get(k, 0.0))
json.dump(out, open("gpurun_out/$tag/pmc_valu.json", "w"), indent=1)
for k, v in out["kernels"].items():
    if v["seconds"] > 0.001:
        print(k, "%.1f ms" % (v["seconds"] * 1e3), "VALU wave-insts %.3g" % v.get("SQ_INSTS_VALU", 0), "SALU %.3g" % v.get("SQ_INSTS_SALU", 0), "VMEM %.3g" % v.get("SQ_INSTS_VMEM", 0), "LDS %.3g" % v.get("SQ_INSTS_LDS", 0), "waves %.3g" % v.get("SQ_WAVES", 0), "active_valu %.3g" % v.get("SQ_ACTIVE_INST_VALU", 0), "busy %.3g" % v.get("SQ_BUSY_CYCLES", 0), "wavecyc %.3g" % v.get("SQ_WAVE_CYCLES", 0))
PY
find gpurun_out/$tag -name "*counter_collection.csv" -delete; find gpurun_out/$tag -name "*_kernel_trace.csv" -delete
