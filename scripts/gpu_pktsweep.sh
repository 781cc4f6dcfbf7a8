for lv in 0,0 1,0 0,1 1,1 ; do
  RT_PACKET_LEVELS=$lv timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu --no-roofline > gpurun_out/pk_$lv.json 2>gpurun_out/pk_$lv.err || exit 1
  echo levels $lv: $(python -c "import json;d=json.load(open('gpurun_out/pk_$lv.json'));print(round(d['value']), round(d['ms_per_step'],1))")
done
