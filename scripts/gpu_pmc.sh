# PMC passes (separate runs, counters only) over one 1080p/16spp frame; outputs under gpurun_out/$1
tag=${1:-pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
CMD="python3 bench.py --width 1920 --height 1080 --spp 16 --steps 1 --warmup 0 --no-cpu --no-roofline"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_READ_sum TCP_TAGRAM0_REQ_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --output-format csv --pmc $set -d gpurun_out/$tag/p$i -- $CMD > gpurun_out/$tag/p$i.log 2>&1 || { echo pass $i failed; tail -5 gpurun_out/$tag/p$i.log; exit 1; }
done
python3 scripts/pmc_summary.py gpurun_out/$tag/p1 gpurun_out/$tag/p2 gpurun_out/$tag/p3 gpurun_out/$tag/p4 gpurun_out/$tag/p5 > gpurun_out/$tag/summary.txt
cat gpurun_out/$tag/summary.txt
