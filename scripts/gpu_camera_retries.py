"""How often a chunk sized from the fan-out seen so far overflows when the camera moves: random cameras, 4K @16 spp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
from test_gpu_cameras import camera
name = sys.argv[1] if len(sys.argv) > 1 else "competition"
scene = FlatScene.builtin(name)
r = Renderer(0).set_scene(scene)
rng = np.random.default_rng(7)
for k in range(12):
    cam = camera(scene.camera, rng, 1.5 if name == "competition" else 4.0)
    st = r.render(3840, 2160, spp=16, camera=cam, want=())["stats"]
    rays = st["rays_closest"] + st["rays_shadow"]
    print(f"camera {k}: {st['ms_total']:.0f} ms, chunks {st['chunks']}, retries {st['chunk_retries']}, rays/primary {rays / st['primary_samples']:.2f}", flush=True)
