"""raytracer_amd -- MI355X-native trace loop for cosmcif/raytracer scenes (HIP wavefront pipeline behind a C ABI)."""
from ._native import Camera, Hit, Material, RenderParams, Stats, TEX, PRIM_MISS, PRIM_ANALYTIC  # noqa: F401
from .scene import FlatScene  # noqa: F401


def __getattr__(name):
    if name in ("Renderer",):
        from .renderer import Renderer
        return Renderer
    raise AttributeError(name)
