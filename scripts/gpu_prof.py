"""wave-time breakdown of the traversal kernels (stats build): python scripts/gpu_prof.py [W H SPP]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
W, H, SPP = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1920, 1080, 16)))
scene = sys.argv[4] if len(sys.argv) > 4 else "competition"
r = Renderer(0).set_scene(FlatScene.builtin(scene))
r.render(W, H, spp=SPP, want=())
st = r.render(W, H, spp=SPP, collect_stats=5, want=())["stats"]
print("max_stack", st["bvh_max_stack"], "depth", st["bvh_depth"], "nodes", st["bvh_nodes"])
for name, rays in (("prof_closest", st["rays_closest"]), ("prof_any", st["rays_shadow"])):
    p = st[name]
    life = p[7]
    print(name, "rays", rays)
    print("  refill %.1f%%  node %.1f%%  tri %.1f%%  other %.1f%% of wave lifetime" % (100 * p[0] / life, 100 * p[1] / life, 100 * p[4] / life, 100 * (life - p[0] - p[1] - p[4]) / life))
    print("  node phases %d (%.0f cycles each, %.1f lanes)   tri phases %d (%.0f cycles each, %.1f lanes)" % (p[2], p[1] / max(p[2], 1), p[3] / max(p[2], 1), p[5], p[4] / max(p[5], 1), p[6] / max(p[5], 1)))
    print("  refills %d (%.0f cycles each, %.1f rays each)" % (p[8], p[0] / max(p[8], 1), p[9] / max(p[8], 1)))
    print("  lane-steps per ray: node %.2f tri %.2f" % (p[3] / max(rays, 1), p[6] / max(rays, 1)))
p = st["prof_shade"]
tot = sum(p[:5])
print("prof_shade hits", st["hits_shaded"], "iterations", p[5], "cycles/iteration %.0f" % (tot / max(p[5], 1)))
for k, name in enumerate(("load+reconstruct", "material/cull/texture/ambient/bounce", "scan for hits", "direct light + shadow emission", "next-level emission")):
    print("  %-40s %5.1f%%  %.0f cycles/iteration" % (name, 100 * p[k] / tot, p[k] / max(p[5], 1)))
for name in ("prof_packet_closest", "prof_packet_any"):
    p = st[name]
    print(name, "packets", p[0], "node steps/packet %.1f (%.1f lanes interested)" % (p[1] / max(p[0], 1), p[2] / max(p[1], 1)),
          "tri steps/packet %.1f (%.1f live lanes)" % (p[3] / max(p[0], 1), p[4] / max(p[3], 1)),
          "over budget %d, octant groups %d, rays to lane kernel %d" % (p[5], p[6], p[7]))
