"""C5 (stress:733) 7680x4320 @4 and @16 spp under workspace sizes and lane counts (one commit, many frames)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
scene = FlatScene.builtin(sys.argv[1] if len(sys.argv) > 1 else "stress:733")
r = Renderer(0).set_scene(scene)
W, H = 7680, 4320
for spp in (4, 16):
    for ws in ("30", "60", "120"):
        for lanes in ("1", "2"):
            os.environ["RT_WORKSPACE_GB"] = ws
            os.environ["RT_LANES"] = lanes
            for k in range(3):
                st = r.render(W, H, spp=spp, want=())["stats"]
            rays = st["rays_closest"] + st["rays_shadow"]
            print(f"spp {spp} workspace {ws} GB lanes {lanes}: {st['ms_total']:.1f} ms, {st['chunks']} chunks, {rays / st['ms_total'] / 1e3:.0f} Mrays/s", flush=True)
