"""where rt_commit spends its time on the 50 M-triangle scene (rt_get_commit_stats + the builder's own phase times)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_VERBOSE"] = "1"
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
t0 = time.time()
s = FlatScene.builtin(sys.argv[1] if len(sys.argv) > 1 else "stress:733")
t1 = time.time()
r = Renderer(0)
for k in range(2):
    t2 = time.time()
    r.set_scene(s)
    cs = r.commit_stats()
    print(f"scene built in {t1 - t0:.1f} s; set_scene (rt_set_* + rt_commit) {time.time() - t2:.2f} s: BVH {cs['ms_bvh_build'] / 1e3:.2f} s, reference order {cs['ms_reference_order'] / 1e3:.2f} s, "
          f"records {cs['ms_records'] / 1e3:.2f} s, upload {cs['ms_upload'] / 1e3:.2f} s = {(cs['ms_bvh_build'] + cs['ms_reference_order'] + cs['ms_records'] + cs['ms_upload']) / 1e3:.2f} s; {cs['bvh_nodes']} nodes", flush=True)
