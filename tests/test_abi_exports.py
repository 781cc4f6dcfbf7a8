"""The C-ABI library loads and exports every symbol include/rt_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

from raytracer_amd import _native as N


def declared_symbols():
    text = open(os.path.join(N.REPO_DIR, "include", "rt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(N.HIP_LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), s


def test_struct_sizes_match_header():
    assert ctypes.sizeof(N.Material) == 96 and ctypes.sizeof(N.Camera) == 96 and ctypes.sizeof(N.Hit) == 96
    assert ctypes.sizeof(N.RenderParams) == 48


def test_no_device_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        return
    lib = N.hip_lib()
    ctx = ctypes.c_void_p()
    assert lib.rt_create(0, ctypes.byref(ctx)) != 0
    assert b"no HIP device" in lib.rt_last_error() or b"hip" in lib.rt_last_error().lower()


def test_tile_counts():
    lib = N.hip_lib()
    a, b = ctypes.c_int64(), ctypes.c_int64()
    assert lib.rt_tile_counts(100, 70, 0, 1, ctypes.byref(a), ctypes.byref(b)) == 0
    assert a.value == b.value == 7 * 5
    tot = 0
    for r in range(8):
        lib.rt_tile_counts(3840, 2160, r, 8, ctypes.byref(a), ctypes.byref(b))
        tot += a.value
    assert tot == b.value == 240 * 135


def test_product_does_not_import_the_oracle():
    import subprocess
    import sys
    code = "import sys; import raytracer_amd, raytracer_amd.renderer, raytracer_amd.scene; assert not any(m.startswith('oracle') for m in sys.modules)"
    subprocess.run([sys.executable, "-c", code], cwd=N.REPO_DIR, check=True)
    for root, _, files in os.walk(os.path.join(N.REPO_DIR, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "librt_oracle" not in txt and "oracle/" not in txt.replace("oracle/ref_harness.cpp", "").replace("oracle/rt_oracle.cpp", ""), f


def test_ctypes_mirrors_have_the_header_sizes(tmp_path):
    """rt_stats & co. are filled by the library through a pointer: a ctypes mirror that is shorter than the C struct would be
    written past its end.  The sizes come from the header itself, compiled with gcc (host code, no GPU)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        return  # (no compiler: nothing to compare with)
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "rt_hip.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(rt_stats), '
                   'sizeof(rt_commit_stats), sizeof(rt_render_params), sizeof(rt_camera), sizeof(rt_material), sizeof(rt_hit)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(N.REPO_DIR, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [ctypes.sizeof(t) for t in (N.Stats, N.CommitStats, N.RenderParams, N.Camera, N.Material, N.Hit)]
    assert got == want
