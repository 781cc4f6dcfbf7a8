# lanes 1 vs 2 on one box: 4K frame and the N = 8 share
mkdir -p gpurun_out/r3ab
for rep in 1 2; do for lanes in 1 2; do
  RT_LANES=$lanes timeout -k 10 120 python bench.py --steps 3 --warmup 2 --no-cpu --no-roofline > gpurun_out/r3ab/l$lanes.json 2> gpurun_out/r3ab/l$lanes.err && echo lanes $lanes: $(python -c "import json;d=json.load(open('gpurun_out/r3ab/l$lanes.json'));print(round(d['ms_per_step'],1), 'ms', d['config']['chunks_per_frame'], 'chunks')") || exit 1
done; done
for lanes in 1 2; do
  RT_LANES=$lanes timeout -k 10 300 python scripts/gpu_rank_share.py > gpurun_out/r3ab/rank_share_l$lanes.txt 2>&1 || { tail -5 gpurun_out/r3ab/rank_share_l$lanes.txt; exit 1; }
  echo lanes $lanes; grep slowest gpurun_out/r3ab/rank_share_l$lanes.txt
done
