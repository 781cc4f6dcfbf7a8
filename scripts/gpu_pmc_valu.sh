# SQ issue counters of one bench.py frame (competition 4K/64spp) -> gpurun_out/$1/pmc_valu.json
# Two passes (8 SQ counters + GRBM_GUI_ACTIVE each): the kernel trace gives durations, the counters give VALU instructions,
# busy cycles and waits; sclk held = GRBM_GUI_ACTIVE / 8 XCDs / kernel time (MI355X_MICROARCH.md, DVFS give-back).
tag=${1:-pmc_valu}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
BENCH="python3 bench.py --sizing-frame --steps 1 --warmup 0 --no-cpu --no-roofline"
timeout -s KILL 300 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/$tag/trace -- $BENCH > gpurun_out/$tag/trace.log 2>&1 || { tail -5 gpurun_out/$tag/trace.log; exit 1; }
timeout -s KILL 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/$tag/sq -- $BENCH > gpurun_out/$tag/sq.log 2>&1 || { tail -5 gpurun_out/$tag/sq.log; exit 1; }
timeout -s KILL 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/$tag/sq2 -- $BENCH > gpurun_out/$tag/sq2.log 2>&1 || { tail -5 gpurun_out/$tag/sq2.log; exit 1; }
python3 scripts/pmc_valu_summary.py gpurun_out/$tag > gpurun_out/$tag/pmc_valu.txt && cat gpurun_out/$tag/pmc_valu.txt
find gpurun_out/$tag -name "*counter_collection.csv" -delete; find gpurun_out/$tag -name "*_kernel_trace.csv" -delete; find gpurun_out/$tag -name "*agent_info.csv" -delete
