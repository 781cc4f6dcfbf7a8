# tests/test_gpu_raylog.py's packet test on the library before the interest masks (/tmp copy made by the caller) and after
cp raytracer_amd/csrc/libraytracer_hip.so /tmp/new.so
if [ -f build_variants/prev.so ]; then
  cp build_variants/prev.so raytracer_amd/csrc/libraytracer_hip.so
  timeout -k 10 300 python -u -m pytest tests/test_gpu_raylog.py -x -q -s -k packet --timeout 200 --timeout-method thread 2>&1 | grep -a "rays:\|passed\|failed"
  cp /tmp/new.so raytracer_amd/csrc/libraytracer_hip.so
fi
timeout -k 10 300 python -u -m pytest tests/test_gpu_raylog.py -x -q -s -k packet --timeout 200 --timeout-method thread 2>&1 | grep -a "rays:\|passed\|failed"
