# C5 (50 M triangles) as the bandwidth case: both stress cameras, in-kernel record counts + rocprofv3 passes of ONE frame each
# (memory-side read bytes of the L2, L2 / L1 hit rates, written bytes) -> gpurun_out/$1, summarised by scripts/c5_summary.py
tag=${1:-r3c5}
W=${W:-3840}; H=${H:-2160}; SPP=${SPP:-16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for cam in stress stress-diagonal; do
  scene=$cam:733
  timeout -k 10 300 python3 scripts/c5_frame.py $scene $W $H $SPP count > gpurun_out/$tag/$cam.plain.json 2> gpurun_out/$tag/$cam.plain.err || { tail -5 gpurun_out/$tag/$cam.plain.err; exit 1; }
  cat gpurun_out/$tag/$cam.plain.json
  for pass in "fetch:FETCH_SIZE" "cache:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "write:WRITE_SIZE"; do
    pname=${pass%%:*}; ctrs=${pass#*:}
    timeout -s KILL 400 rocprofv3 --output-format csv --pmc $ctrs -d gpurun_out/$tag/$cam.$pname -- python3 scripts/c5_frame.py $scene $W $H $SPP > gpurun_out/$tag/$cam.$pname.log 2>&1 || { tail -5 gpurun_out/$tag/$cam.$pname.log; exit 1; }
    python3 - <<PY
import csv, glob, collections, json
cnt = collections.defaultdict(collections.Counter); disp = collections.defaultdict(set)
for f in glob.glob("gpurun_out/$tag/$cam.$pname/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        cnt[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
json.dump({k: dict(v, dispatches=len(disp[k])) for k, v in cnt.items() if k.startswith("k_")}, open("gpurun_out/$tag/$cam.$pname.json", "w"), indent=1)
PY
    rm -rf gpurun_out/$tag/$cam.$pname
    echo "$cam $pname done"
  done
done
