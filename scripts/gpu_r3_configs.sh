# BASELINE.json configurations C2..C5 on the round-3 kernels -> gpurun_out/configs3 (table: profiles/r03_configs.md)
mkdir -p gpurun_out/configs3
bash scripts/gpu_configs.sh > gpurun_out/configs3/c2c3.txt 2>&1 || { tail -5 gpurun_out/configs3/c2c3.txt; exit 1; }
cat gpurun_out/configs3/c2c3.txt
timeout -k 10 300 python -u scripts/gpu_stress.py 64 3840 2160 16 > gpurun_out/configs3/c5lite.txt 2>&1 || { tail -5 gpurun_out/configs3/c5lite.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/configs3/c5lite.txt
timeout -k 10 400 python -u scripts/gpu_stress.py 733 7680 4320 4 > gpurun_out/configs3/c5_4spp.txt 2>&1 || { tail -5 gpurun_out/configs3/c5_4spp.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/configs3/c5_4spp.txt
RT_STRESS_DETAIL=0 timeout -k 10 400 python -u scripts/gpu_stress.py 733 7680 4320 256 > gpurun_out/configs3/c5_full.txt 2>&1 || { tail -5 gpurun_out/configs3/c5_full.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/configs3/c5_full.txt
