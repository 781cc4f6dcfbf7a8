"""lanes per phase of the shadow and bounce phases of the 4K frame (counting build of the kernels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
r = Renderer(0).set_scene(FlatScene.builtin(sys.argv[1] if len(sys.argv) > 1 else "competition"))
r.render(3840, 2160, spp=16, want=())
st = r.render(3840, 2160, spp=16, collect_stats=1, want=())["stats"]
for name in ("prof_closest", "prof_any"):
    p = st[name]
    print(name, f"node phases {p[2]} at {p[3] / max(p[2], 1):.1f} lanes, triangle phases {p[5]} at {p[6] / max(p[5], 1):.1f} lanes, refills {p[8]}, rays per refill {p[9] / max(p[8], 1):.1f}")
print({k: st[k] for k in ("rays_closest", "rays_shadow", "shadow_unoccluded", "nodes_visited", "tris_tested", "nodes_visited_any", "tris_tested_any")})
