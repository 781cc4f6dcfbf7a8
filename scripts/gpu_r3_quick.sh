# quick check of a kernel change: the parity-critical GPU tests, then the 4K frame (bench.py line with the in-kernel lane statistics)
tag=${1:-r3q}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_chunks.py tests/test_gpu_split.py tests/test_zz_no_alloc.py -x -q --timeout 200 --timeout-method thread > gpurun_out/$tag/gpu_tests.log 2>&1 || { tail -40 gpurun_out/$tag/gpu_tests.log; exit 1; }
tail -2 gpurun_out/$tag/gpu_tests.log
fi
timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-cpu $BENCH_ARGS > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { tail -20 gpurun_out/$tag/bench.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("gpurun_out/$tag/bench.json"))
print("Mrays/s", round(d["value"]), "ms/frame", round(d["ms_per_step"],1), d["config"].get("chunks_per_frame"), "chunks", {k: round(v,1) for k,v in d.get("kernel_ms",{}).items()})
r=d.get("roofline",{}); print({k:r.get(k) for k in ("kernel","achieved","frac","records_per_ray")}, r.get("lanes"))
PY
