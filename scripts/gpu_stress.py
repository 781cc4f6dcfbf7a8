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
