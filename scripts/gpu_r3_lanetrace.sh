# kernel trace of the 4K frame with two lanes: which HW queues the lanes' kernels land on and whether they overlap
tag=${1:-r3lanetrace}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
RT_LANES=2 timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag/trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-roofline > gpurun_out/$tag/trace.log 2>&1 || { tail -5 gpurun_out/$tag/trace.log; exit 1; }
find gpurun_out/$tag/trace -name "*_kernel_trace.csv" -exec cp {} gpurun_out/$tag/kernel_trace.csv \;
rm -rf gpurun_out/$tag/trace
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("gpurun_out/$tag/kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_assemble' in r['Kernel_Name']]
fr=rows[idx[-2]+1:idx[-1]+1]
print("launches", len(fr), "queues", collections.Counter((r['Queue_Id'], r['Stream_Id']) for r in fr))
t0=int(fr[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in fr)
ev=[]
for r in fr: ev.append((int(r['Start_Timestamp']),1)); ev.append((int(r['End_Timestamp']),-1))
ev.sort(); cur=0; last=t0; hist=collections.Counter()
for t,d in ev:
    hist[cur]+=t-last; last=t; cur+=d
print("frame ms", (t1-t0)/1e6, "time with k kernels in flight (ms):", {k: round(v/1e6,1) for k,v in sorted(hist.items())})
PY
