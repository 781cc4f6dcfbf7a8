"""debug: stats-build frame of the stress scene (prints the long shadow rays the kernel reports)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
name = sys.argv[1]
W, H, SPP = (int(x) for x in sys.argv[2:5])
scene = FlatScene.builtin(name)
r = Renderer(0); r.set_scene(scene)
c = r.render(W, H, spp=SPP, want=(), collect_stats=1)["stats"]
print("longest walk closest", c["max_walk_closest"], "any", c["max_walk_any"], "ms", c["ms_total"])
