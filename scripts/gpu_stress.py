"""C5-style stress: competition meshes replicated on a lattice; reports build and frame times.
python scripts/gpu_stress.py REPLICAS W H SPP"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
rep, W, H, SPP = (int(x) for x in sys.argv[1:5])
t0 = time.time(); scene = FlatScene.builtin(f"stress:{rep}"); t1 = time.time()
print(f"scene: {scene.n_triangles} triangles flattened in {t1 - t0:.1f} s", flush=True)
r = Renderer(0); r.set_scene(scene); t2 = time.time()
print(f"rt_commit (host BVH build + upload): {t2 - t1:.1f} s", flush=True)
for k in range(2):
    st = r.render(W, H, spp=SPP, want=())["stats"]
rays = st["rays_closest"] + st["rays_shadow"]
print(f"{W}x{H} @{SPP}spp: {st['ms_total']:.1f} ms, {rays / st['ms_total'] / 1e3:.0f} Mrays/s, {rays / st['primary_samples']:.2f} rays/primary, "
      f"nodes {st['bvh_nodes']} depth {st['bvh_depth']} max_stack {st['bvh_max_stack']}")
if os.environ.get("RT_STRESS_DETAIL", "1") != "0":
    t = r.render(W, H, spp=SPP, want=(), collect_stats=2)["stats"]
    print("kernel ms:", {k: round(t[k], 1) for k in ("ms_gen", "ms_trace", "ms_packet", "ms_closest", "ms_shade", "ms_any", "ms_resolve", "ms_total") if t.get(k, 0) > 0.05}, flush=True)
    c = r.render(W, H, spp=SPP, want=(), collect_stats=1)["stats"]
    print(f"records per closest ray: nodes {c['nodes_visited'] / c['rays_closest']:.2f} tris {c['tris_tested'] / c['rays_closest']:.2f}; "
          f"per shadow ray: nodes {c['nodes_visited_any'] / c['rays_shadow']:.2f} tris {c['tris_tested_any'] / c['rays_shadow']:.2f}; "
          f"rays closest {c['rays_closest']} shadow {c['rays_shadow']}; longest walk closest {c['max_walk_closest']} any {c['max_walk_any']}")
