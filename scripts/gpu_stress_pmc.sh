# C5-style stress (733 replicas ~ 50 M triangles, 8K @4spp): kernel split + counters, then HBM traffic from two PMC passes
tag=${1:-stress_r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout -k 10 400 python3 scripts/gpu_stress.py 733 7680 4320 4 > gpurun_out/$tag/detail.log 2>&1 || { tail -5 gpurun_out/$tag/detail.log; exit 1; }
cat gpurun_out/$tag/detail.log
export RT_STRESS_DETAIL=0
timeout -s KILL 400 rocprofv3 --output-format csv --pmc FETCH_SIZE -d gpurun_out/$tag/fetch -- python3 scripts/gpu_stress.py 733 7680 4320 4 > gpurun_out/$tag/fetch.log 2>&1 || { tail -5 gpurun_out/$tag/fetch.log; exit 1; }
timeout -s KILL 400 rocprofv3 --output-format csv --pmc WRITE_SIZE -d gpurun_out/$tag/write -- python3 scripts/gpu_stress.py 733 7680 4320 4 > gpurun_out/$tag/write.log 2>&1 || { tail -5 gpurun_out/$tag/write.log; exit 1; }
python3 scripts/pmc_traffic.py gpurun_out/$tag/fetch gpurun_out/$tag/write "stress:733 7680x4320 @4spp depth 3 (2 frames)" > gpurun_out/$tag/pmc_traffic.json
find gpurun_out/$tag -name "*counter_collection.csv" -delete
python3 - <<PY
import json
t=json.load(open("gpurun_out/$tag/pmc_traffic.json"))
for k,v in t["kernels"].items(): print(k, v["launches"], "launches", round(v["hbm_bytes_per_frame"]/1e9,1), "GB total: fetch", round(2*v["fetch_bytes_raw"]/1e9,1), "write", round(v["write_bytes_raw"]/1e9,1))
PY
