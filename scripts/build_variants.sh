# Builds libraytracer_hip.so variants with overridden tuning macros into build_variants/<name>.so (parameter sweeps:
# scripts/gpu_variants.sh runs bench.py on each).  usage: bash scripts/build_variants.sh "name:-DMACRO=v -DMACRO2=w" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
build_one() {
  name=${1%%:*}; defs=${1#*:}
  d=build_variants/obj_$name; mkdir -p $d
  for f in rt_kernels.hip rt_abi.cpp rt_bvh.cpp; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize $defs -Iraytracer_amd/csrc -c raytracer_amd/csrc/$f -o $d/${f%.*}.o
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o build_variants/$name.so $d/rt_kernels.o $d/rt_abi.o $d/rt_bvh.o
  rm -rf $d
  echo built $name "($defs)"
}
for v in "$@"; do build_one "$v" & 
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
