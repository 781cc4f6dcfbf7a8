"""ctypes view of the two product libraries.

* ``libraytracer_hip.so``  -- the C ABI of include/rt_hip.h (HIP kernels + BVH builder + render loop)
* ``libraytracer_host.so`` -- the C++ scene-definition surface (host/) with C entry points (host/host_capi.cpp)

There is no Python or CPU fallback: if the HIP library is missing or cannot be loaded, importing the
renderer raises.  Structures mirror include/rt_hip.h field for field.
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
HIP_LIB_PATH = os.path.join(PKG_DIR, "csrc", "libraytracer_hip.so")
HOST_LIB_PATH = os.path.join(PKG_DIR, "host", "libraytracer_host.so")
ASSET_DIR = os.path.join(REPO_DIR, "assets")

TEX = {
    "none": 0, "perlinNormal": 1, "perlinWater": 2, "opal": 3, "perlinTerrain": 4, "perlinIceTerrain": 5,
    "snowTerrain": 6, "qwilfishTexture": 7, "colorAt": 8, "normalAt": 9, "roughnessAt": 10,
    "ambientOcclusionAt": 11,
}
PRIM_MISS = 0xFFFFFFFF
PRIM_ANALYTIC = 0x80000000
JITTER_AUTO, JITTER_TABLE, JITTER_HASH = 0, 1, 2


class Material(C.Structure):
    _fields_ = [
        ("ambient", C.c_float * 3), ("diffuse", C.c_float * 3), ("specular", C.c_float * 3),
        ("reflection", C.c_float), ("refraction", C.c_float), ("sigma", C.c_float), ("shininess", C.c_float),
        ("texture", C.c_int32), ("has_normal_map", C.c_int32), ("normal_map", C.c_int32),
        ("is_anisotropic", C.c_int32), ("alpha_x", C.c_float), ("alpha_y", C.c_float),
        ("has_img_texture", C.c_int32), ("roughness", C.c_int32), ("occlusion", C.c_int32),
        ("_pad", C.c_int32 * 2),
    ]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("dz", C.c_float), ("rot", C.c_float * 16), ("fov_deg", C.c_float),
                ("_pad", C.c_float * 3)]


class RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
        ("seed", C.c_uint32), ("jitter_mode", C.c_int32), ("tile_rank", C.c_int32), ("tile_world", C.c_int32),
        ("chunk_samples", C.c_int64), ("collect_stats", C.c_int32), ("flags", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("primary_samples", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
        ("shadow_culled", C.c_uint64), ("hits_shaded", C.c_uint64), ("nodes_visited", C.c_uint64),
        ("tris_tested", C.c_uint64), ("queue_overflow", C.c_uint64), ("ms_total", C.c_double),
        ("ms_gen", C.c_double), ("ms_closest", C.c_double), ("ms_shade", C.c_double), ("ms_any", C.c_double),
        ("ms_resolve", C.c_double), ("launches_closest", C.c_uint64), ("launches_any", C.c_uint64),
        ("bvh_nodes", C.c_uint64), ("bvh_depth", C.c_uint64), ("n_triangles", C.c_uint64),
        ("nodes_visited_any", C.c_uint64), ("tris_tested_any", C.c_uint64), ("bvh_max_stack", C.c_uint64), ("prof_closest", C.c_uint64 * 10), ("prof_any", C.c_uint64 * 10), ("prof_shade", C.c_uint64 * 6), ("prof_packet_closest", C.c_uint64 * 8), ("prof_packet_any", C.c_uint64 * 8),
        ("max_walk_closest", C.c_uint64), ("max_walk_any", C.c_uint64),
        ("ms_packet", C.c_double), ("shadow_unoccluded", C.c_uint64), ("split_items", C.c_uint64), ("shadow_zero", C.c_uint64),
        ("chunks", C.c_uint64), ("chunk_retries", C.c_uint64), ("ms_trace", C.c_double),
    ]

    def as_dict(self):
        return {name: (list(getattr(self, name)) if name.startswith("prof_") else getattr(self, name)) for name, _ in self._fields_}


class MultiTimings(C.Structure):
    _fields_ = [("ms_render_wall", C.c_double), ("ms_share_max", C.c_double), ("ms_share_min", C.c_double),
                ("ms_exchange", C.c_double), ("ms_assemble", C.c_double), ("ms_readback", C.c_double)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class CommitStats(C.Structure):
    _fields_ = [("n_triangles", C.c_uint64), ("bvh_nodes", C.c_uint64), ("ms_bvh_build", C.c_double),
                ("ms_reference_order", C.c_double), ("ms_records", C.c_double), ("ms_upload", C.c_double)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class Hit(C.Structure):
    _fields_ = [
        ("hit", C.c_int32), ("object", C.c_int32), ("triangle", C.c_int32), ("prim", C.c_uint32),
        ("distance", C.c_float), ("intersection", C.c_float * 3), ("normal", C.c_float * 3),
        ("normal_shading", C.c_float * 3), ("uv", C.c_float * 2), ("tangent", C.c_float * 3),
        ("bitangent", C.c_float * 3), ("_pad", C.c_float * 2),
    ]


def kernel_source_hash():
    """sha256 over the device-side sources and the BVH builder (the tree shapes every counter): profiles/*.json carry it so that
    counter values are only ever attached to measurements of the kernels and the tree they were collected on (bench.py)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("rt_kernels.hip", "rt_device.h", "rt_types.h", "rt_bvh.cpp"):
        h.update(open(os.path.join(PKG_DIR, "csrc", f), "rb").read())
    return h.hexdigest()[:16]


assert C.sizeof(Material) == 96 and C.sizeof(Hit) == 96 and C.sizeof(Camera) == 96

_hip = None
_host = None


def hip_lib():
    """Loads libraytracer_hip.so and declares the C ABI.  Raises if it is absent: no fallback."""
    global _hip
    if _hip is not None:
        return _hip
    if not os.path.exists(HIP_LIB_PATH):
        raise RuntimeError(
            f"{HIP_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
            "raytracer_amd has no CPU fallback")
    lib = C.CDLL(HIP_LIB_PATH, mode=C.RTLD_GLOBAL)
    P, I32, I64, VP = C.POINTER, C.c_int32, C.c_int64, C.c_void_p
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_create.argtypes = [C.c_int, P(VP)]
    lib.rt_destroy.argtypes = [VP]
    lib.rt_set_materials.argtypes = [VP, I32, VP]
    lib.rt_set_object_materials.argtypes = [VP, I32, VP]
    lib.rt_set_triangles.argtypes = [VP, I64] + [VP] * 11
    lib.rt_set_spheres.argtypes = [VP, I32, VP, VP, VP, VP]
    lib.rt_set_planes.argtypes = [VP, I32, VP, VP, VP]
    lib.rt_set_cones.argtypes = [VP, I32, VP, VP, VP, VP]
    lib.rt_set_lights.argtypes = [VP, I32, VP, VP, VP]
    lib.rt_set_bmp.argtypes = [VP, I32, I32, I32, I32, VP]
    lib.rt_commit.argtypes = [VP]
    lib.rt_render.argtypes = [VP, P(Camera), P(RenderParams), VP, VP, VP, P(Stats)]
    lib.rt_tile_counts.argtypes = [I32, I32, I32, I32, P(I64), P(I64)]
    lib.rt_render_tiles_device.argtypes = [VP, P(Camera), P(RenderParams), VP, VP, VP, VP, P(Stats)]
    lib.rt_render_tiles_static.argtypes = [VP, P(Camera), P(RenderParams), VP, VP, VP, VP]
    lib.rt_assemble_tiles_device.argtypes = [VP, I32, I32, I32, I64, VP, VP, VP, VP, VP, VP, VP]
    lib.rt_multi_create.argtypes = [P(I32), I32, C.c_uint32, P(VP)]
    lib.rt_multi_destroy.argtypes = [VP]
    lib.rt_multi_size.argtypes = [VP]
    lib.rt_multi_transport.argtypes = [VP]
    lib.rt_multi_transport.restype = C.c_char_p
    lib.rt_multi_context.argtypes = [VP, I32]
    lib.rt_multi_context.restype = VP
    lib.rt_multi_commit.argtypes = [VP]
    lib.rt_multi_render.argtypes = [VP, P(Camera), P(RenderParams), VP, VP, VP, VP]
    lib.rt_multi_last_timings.argtypes = [VP, P(MultiTimings)]
    lib.rt_get_commit_stats.argtypes = [VP, P(CommitStats)]
    lib.rt_trace_batch.argtypes = [VP, I64, VP, I32, VP, VP, VP, P(Stats)]
    lib.rt_eval_texture.argtypes = [VP, I32, I64, VP, VP]
    lib.rt_eval_perlin.argtypes = [VP, I64, VP, VP]
    lib.rt_get_bvh.argtypes = [VP, P(I64), VP, P(I64), VP]
    lib.rt_get_bvh_info.argtypes = [VP, P(C.c_uint32), P(C.c_float), P(I32), P(I32)]
    lib.rt_bvh_check.argtypes = [I64, VP, VP, VP, I32, P(I64), P(C.c_uint64), P(I64)]
    _hip = lib
    return lib


def host_lib(lazy=False):
    """Loads libraytracer_host.so (scene-definition surface).  It links against the HIP library; ``lazy``
    lets CPU-only tooling (fixture generation, oracle pinning) use the flattening entry points without a GPU."""
    global _host
    if _host is not None:
        return _host
    if not os.path.exists(HOST_LIB_PATH):
        raise RuntimeError(f"{HOST_LIB_PATH} is missing: run __graft_entry__.build()")
    if not lazy:
        hip_lib()
    lib = C.CDLL(HOST_LIB_PATH, mode=(os.RTLD_LAZY if lazy else os.RTLD_NOW) | os.RTLD_GLOBAL)
    VP = C.c_void_p
    lib.rth_last_error.restype = C.c_char_p
    lib.rth_scene_create.restype = VP
    lib.rth_scene_create.argtypes = [C.c_char_p, C.c_char_p]
    lib.rth_scene_destroy.argtypes = [VP]
    lib.rth_scene_counts.argtypes = [VP, VP]
    lib.rth_scene_camera.argtypes = [VP, C.POINTER(Camera)]
    lib.rth_scene_get_triangles.argtypes = [VP] * 12
    lib.rth_scene_get_analytic.argtypes = [VP] * 8
    lib.rth_scene_get_shading.argtypes = [VP] * 6
    lib.rth_scene_get_cones.argtypes = [VP] * 5
    lib.rth_read_bmp.argtypes = [C.c_char_p, VP, VP, VP, VP]
    lib.rth_mesh_create.restype = VP
    lib.rth_mesh_create.argtypes = [C.c_char_p, C.c_float, C.c_float, C.c_float, VP]
    lib.rth_scene_upload.argtypes = [VP, VP]
    lib.rth_scan_float.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    _host = lib
    return lib
