# L1 (TCP) / L2 (TCC) hit rates of the kernels on one bench.py frame -> gpurun_out/$1
tag=${1:-pmc_cache}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
BENCH="python3 bench.py --sizing-frame --steps 1 --warmup 0 --no-cpu --no-roofline"
timeout -s KILL 120 rocprofv3 --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/$tag/c1 -- $BENCH > gpurun_out/$tag/c1.log 2>&1 || { tail -5 gpurun_out/$tag/c1.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json, os, sys
sys.path.insert(0, os.getcwd())
from raytracer_amd._native import kernel_source_hash
cnt = collections.defaultdict(collections.Counter)
for f in glob.glob("gpurun_out/$tag/c1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
out = {"workload": "competition scene 3840x2160 @64spp depth 3", "kernel_source_hash": kernel_source_hash(),
       "source": "rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum, one frame (+ a 1/64 sizing frame)", "kernels": {}}
for k, c in sorted(cnt.items()):
    a, m = c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0), c.get("TCP_TCC_READ_REQ_sum", 0)
    h, ms = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    if a > 1e6:
        out["kernels"][k] = {"l1_accesses": a, "l1_read_requests_to_l2": m, "l1_hit_rate": 1 - m / a, "l2_hits": h, "l2_misses": ms, "l2_hit_rate": h / max(h + ms, 1)}
        print("%-28s L1 accesses %.3g, to L2 %.3g (L1 hit %.1f %%), L2 hit %.3g miss %.3g (L2 hit %.1f %%)" % (k[:28], a, m, 100 * (1 - m / a), h, ms, 100 * h / max(h + ms, 1)))
json.dump(out, open("gpurun_out/$tag/pmc_cache.json", "w"), indent=1)
PY
find gpurun_out/$tag -name "*counter_collection.csv" -delete
