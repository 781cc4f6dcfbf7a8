set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1b
for c in 524288 1048576 4194304 8388608 16777216; do
  timeout -k 10 120 python bench.py --steps 2 --warmup 1 --no-cpu --no-roofline --chunk $c > gpurun_out/r1b/chunk_$c.json 2> gpurun_out/r1b/chunk_$c.err && echo chunk $c: $(python -c "import json;d=json.load(open('gpurun_out/r1b/chunk_$c.json'));print(d['value'], d['ms_per_step'])")
done
