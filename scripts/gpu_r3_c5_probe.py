"""C5 with its two cameras: records fetched per ray, frame time, node-fetch bytes per second (64 B per record)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 733
W, H, SPP = (int(x) for x in (sys.argv[2:5] if len(sys.argv) > 4 else (1920, 1080, 4)))
for name in (f"stress:{reps}", f"stress-diagonal:{reps}"):
    t0 = time.time()
    s = FlatScene.builtin(name)
    r = Renderer(0).set_scene(s)
    print(f"== {name}: {s.n_triangles} triangles, load + commit {time.time() - t0:.1f} s", flush=True)
    r.render(W, H, spp=SPP, want=())
    st = r.render(W, H, spp=SPP, collect_stats=1 | 2, want=())["stats"]
    pl = r.render(W, H, spp=SPP, collect_stats=2, want=())["stats"]
    rays = st["rays_closest"] + st["rays_shadow"]
    rec = st["nodes_visited"] + st["tris_tested"] + st["nodes_visited_any"] + st["tris_tested_any"]
    pk = st["prof_packet_closest"]
    print(f"   {W}x{H} @{SPP}: {pl['ms_total']:.1f} ms ({st['chunks']} chunks), trace {pl['ms_trace']:.1f} shade {pl['ms_shade']:.1f}; rays {rays/1e6:.1f} M "
          f"({rays / st['primary_samples']:.2f} per primary); records per ray: closest {(st['nodes_visited'] + st['tris_tested']) / max(st['rays_closest'], 1):.1f}, "
          f"shadow {(st['nodes_visited_any'] + st['tris_tested_any']) / max(st['rays_shadow'], 1):.1f}; packets: {pk[1] / max(pk[0], 1):.1f} node steps each, "
          f"{pk[7]} rays finished per lane; longest walk {st['max_walk_closest']} / {st['max_walk_any']}; "
          f"per-lane record fetches {rec * 64 / 1e9:.1f} GB -> {rec * 64 / 1e9 / (pl['ms_trace'] / 1e3):.0f} GB/s of k_trace time", flush=True)
    r.close()
