"""One frame of a built-in scene (default: the 50 M-triangle stress scene, second camera) -- the program the rocprofv3 passes of
scripts/gpu_r3_c5.sh run.  Prints one JSON line: frame time, rays, records fetched (in-kernel counters of a separate frame)."""
import json, os, sys
os.environ.setdefault("RT_LANES", "1")  # one pipeline: the kernels run one after the other, a launch's time is its own (the default for this scene is two)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
name = sys.argv[1] if len(sys.argv) > 1 else "stress-diagonal:733"
W, H, SPP = (int(x) for x in (sys.argv[2:5] if len(sys.argv) > 4 else (3840, 2160, 16)))
counted = len(sys.argv) > 5 and sys.argv[5] == "count"
r = Renderer(0).set_scene(FlatScene.builtin(name))
r.render(max(W // 8, 64), max(H // 8, 64), spp=SPP, want=())      # a SMALL frame sizes the queues from the scene's fan-out (1.6 % of the
                                                                   # measured frame's samples: the rocprofv3 passes sum over the process)
st = r.render(W, H, spp=SPP, collect_stats=2, want=())["stats"]    # the frame the counters are read on
out = {"scene": name, "width": W, "height": H, "spp": SPP, "ms_total": st["ms_total"], "ms_trace": st["ms_trace"], "ms_shade": st["ms_shade"],
       "chunks": st["chunks"], "rays_closest": st["rays_closest"], "rays_shadow": st["rays_shadow"], "triangles": st["n_triangles"],
       "bvh_nodes": st["bvh_nodes"]}
if counted:
    c = r.render(W, H, spp=SPP, collect_stats=1, want=())["stats"]
    pk = c["prof_packet_closest"]
    out.update(nodes_visited=c["nodes_visited"], tris_tested=c["tris_tested"], nodes_visited_any=c["nodes_visited_any"],
               tris_tested_any=c["tris_tested_any"], packet_node_steps=pk[1], packet_tri_steps=pk[3], packets=pk[0])
print(json.dumps(out), flush=True)
r.close()
