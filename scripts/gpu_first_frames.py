"""chunks / retries / time of the first frames after rt_commit (adaptive chunk sizing): python scripts/gpu_first_frames.py [W H SPP]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
W, H, SPP = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (3840, 2160, 64)))
scene = sys.argv[4] if len(sys.argv) > 4 else "competition"
single = os.environ.get("SINGLE", "0") == "1"   # RT_RENDER_SINGLE_FRAME on every frame
r = Renderer(0).set_scene(FlatScene.builtin(scene))
for k in range(2 if single else 4):
    t0 = time.perf_counter()
    st = r.render(W, H, spp=SPP, want=(), single_frame=single)["stats"]
    dt = (time.perf_counter() - t0) * 1e3
    print(f"frame {k}: wall {dt:.0f} ms, GPU {st['ms_total']:.0f} ms, chunks {st['chunks']}, retries {st['chunk_retries']}", flush=True)
