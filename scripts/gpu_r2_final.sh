# round-2 evidence in one call: kernel-trace stats + HBM traffic passes, SQ issue passes, the bench line, the other BASELINE
# configurations, C5 (stated size, reduced spp and full), ingest timing.  Outputs under gpurun_out/$1.
tag=${1:-r2_final}
mkdir -p gpurun_out/$tag
bash scripts/gpu_profile_bench.sh $tag/prof || exit 1
bash scripts/gpu_pmc_valu.sh $tag/valu || exit 1
timeout -k 10 400 python bench.py --steps 5 --warmup 2 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { tail -5 gpurun_out/$tag/bench.err; exit 1; }
bash scripts/gpu_configs.sh > gpurun_out/$tag/configs.txt 2>&1 || { tail -5 gpurun_out/$tag/configs.txt; exit 1; }
cp gpurun_out/configs/c2.json gpurun_out/configs/c3.json gpurun_out/$tag/ 2>/dev/null
cat gpurun_out/$tag/configs.txt
timeout -k 10 400 python3 scripts/gpu_stress.py 733 7680 4320 4 > gpurun_out/$tag/c5_4spp.log 2>&1 || { tail -5 gpurun_out/$tag/c5_4spp.log; exit 1; }
cat gpurun_out/$tag/c5_4spp.log
RT_STRESS_DETAIL=0 timeout -k 10 600 python3 scripts/gpu_stress.py 733 7680 4320 256 > gpurun_out/$tag/c5_full.log 2>&1 || { tail -5 gpurun_out/$tag/c5_full.log; exit 1; }
cat gpurun_out/$tag/c5_full.log
timeout -k 10 300 python3 scripts/gpu_stress.py 64 3840 2160 16 > gpurun_out/$tag/c5_lite.log 2>&1 || { tail -5 gpurun_out/$tag/c5_lite.log; exit 1; }
tail -4 gpurun_out/$tag/c5_lite.log
for s in competition sample; do
  timeout -k 10 120 raytracer_amd/host/raytracer_hip --scene $s --width 1024 --height 768 --assets assets gpurun_out/$tag/$s.ppm > gpurun_out/$tag/ingest_$s.txt 2>&1 || { cat gpurun_out/$tag/ingest_$s.txt; exit 1; }
  cat gpurun_out/$tag/ingest_$s.txt
  rm -f gpurun_out/$tag/$s.ppm
done
