"""longest walks of the 4K frame (stats build counters)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
r = Renderer(0).set_scene(FlatScene.builtin(sys.argv[1] if len(sys.argv) > 1 else "competition"))
r.render(3840, 2160, spp=16, want=())
st = r.render(3840, 2160, spp=16, collect_stats=1, want=())["stats"]
print({k: st[k] for k in ("max_walk_closest", "max_walk_any", "rays_closest", "rays_shadow", "nodes_visited", "tris_tested", "nodes_visited_any", "tris_tested_any", "split_items", "bvh_max_stack", "bvh_depth")})
