for c in 33554432 67108864; do
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu --no-roofline --chunk $c > gpurun_out/chunk_$c.json 2>gpurun_out/chunk_$c.err || { tail -3 gpurun_out/chunk_$c.err; exit 1; }
  echo chunk $c: $(python -c "import json;d=json.load(open('gpurun_out/chunk_$c.json'));print(round(d['value']), round(d['ms_per_step'],1))")
done
