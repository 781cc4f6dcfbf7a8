"""ctypes binding of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never
by the raytracer_amd package.  Takes the same FlatScene arrays the HIP renderer takes.
"""
import ctypes as C
import os

import numpy as np

from raytracer_amd import _native as N

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librt_oracle.so")
REF_O2 = os.path.join(HERE, "_ref", "ref_harness_O2")
REF_OFAST = os.path.join(HERE, "_ref", "ref_harness_Ofast")


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_materials", C.c_int32), ("materials", C.c_void_p),
        ("n_objects", C.c_int32), ("object_material", C.c_void_p),
        ("n_triangles", C.c_int64),
        ("v0", C.c_void_p), ("v1", C.c_void_p), ("v2", C.c_void_p), ("nA", C.c_void_p), ("nB", C.c_void_p),
        ("nC", C.c_void_p), ("tA", C.c_void_p), ("tB", C.c_void_p), ("tC", C.c_void_p),
        ("tri_flags", C.c_void_p), ("tri_object", C.c_void_p),
        ("n_spheres", C.c_int32), ("sph_M", C.c_void_p), ("sph_Minv", C.c_void_p), ("sph_N", C.c_void_p),
        ("sph_object", C.c_void_p),
        ("n_planes", C.c_int32), ("pl_point", C.c_void_p), ("pl_normal", C.c_void_p), ("pl_object", C.c_void_p),
        ("n_lights", C.c_int32), ("light_pos", C.c_void_p), ("light_color", C.c_void_p),
        ("ambient", C.c_float * 3),
        ("bmp_w", C.c_int32 * 4), ("bmp_h", C.c_int32 * 4), ("bmp_c", C.c_int32 * 4), ("bmp", C.c_void_p * 4),
        ("n_cones", C.c_int32), ("cone_M", C.c_void_p), ("cone_Minv", C.c_void_p), ("cone_N", C.c_void_p),
        ("cone_object", C.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle librt_oracle.so`")
        L = C.CDLL(LIB_PATH)
        VP, P = C.c_void_p, C.POINTER
        L.ro_create.restype = VP
        L.ro_create.argtypes = [P(SceneDesc)]
        L.ro_destroy.argtypes = [VP]
        L.ro_render.argtypes = [VP, P(N.Camera), P(N.RenderParams), VP, VP, VP, VP, P(N.Stats), C.c_int]
        L.ro_trace_batch.argtypes = [VP, C.c_int64, VP, C.c_int32, VP, VP, VP, P(N.Stats), C.c_int]
        L.ro_eval_texture.argtypes = [VP, C.c_int32, C.c_int64, VP, VP]
        L.ro_eval_perlin.argtypes = [C.c_int64, VP, VP]
        L.ro_reference_order.argtypes = [VP, VP]
        L.ro_set_ray_log.argtypes = [C.c_char_p]
        L.ro_walk_counts.argtypes = [VP, C.c_uint32, C.c_float, C.c_int64, VP, VP, VP, VP, C.c_int64, VP, VP, VP, VP, VP]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def set_ray_log(path):
    """every closest-hit ray the oracle traces from now on goes to `path` (None stops); see read_ray_log"""
    if lib().ro_set_ray_log(path.encode() if path else None):
        raise RuntimeError(f"cannot write {path}")


def read_ray_log(path):
    """-> rays (n, 6) float32, hit (n,), object (n,), triangle (n,; scene numbering), distance (n,) float32 -- exact bits"""
    raw = np.loadtxt(path, dtype=str, ndmin=2)
    if raw.size == 0:
        z = np.zeros(0, np.int64)
        return np.zeros((0, 6), np.float32), z, z, z, np.zeros(0, np.float32)
    rays = np.array([[int(x, 16) for x in r[:6]] for r in raw], np.uint32).view(np.float32)
    dist = np.array([int(r[9], 16) for r in raw], np.uint32).view(np.float32)
    return rays, raw[:, 6].astype(np.int64), raw[:, 7].astype(np.int64), raw[:, 8].astype(np.int64), dist


def eval_perlin(points):
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    out = np.zeros(len(pts), np.float32)
    lib().ro_eval_perlin(len(pts), _p(pts), _p(out))
    return out


class Oracle:
    """CPU restatement of the reference trace loop over a FlatScene."""

    def __init__(self, scene):
        self.scene = scene
        s = scene
        d = SceneDesc()
        self._keep = [np.ascontiguousarray(a) for a in (
            s.materials, s.object_material, s.v0, s.v1, s.v2, s.nA, s.nB, s.nC, s.tA, s.tB, s.tC, s.tri_flags,
            s.tri_object, s.sph_M, s.sph_Minv, s.sph_N, s.sph_object, s.pl_point, s.pl_normal, s.pl_object,
            s.light_pos, s.light_color, s.cone_M, s.cone_Minv, s.cone_N, s.cone_object)]
        k = self._keep
        d.n_materials, d.materials = len(s.materials), _p(k[0])
        d.n_objects, d.object_material = s.n_objects, _p(k[1])
        d.n_triangles = s.n_triangles
        (d.v0, d.v1, d.v2, d.nA, d.nB, d.nC, d.tA, d.tB, d.tC, d.tri_flags, d.tri_object) = [_p(a) for a in k[2:13]]
        d.n_spheres = len(s.sph_object)
        d.sph_M, d.sph_Minv, d.sph_N, d.sph_object = [_p(a) for a in k[13:17]]
        d.n_planes = len(s.pl_object)
        d.pl_point, d.pl_normal, d.pl_object = [_p(a) for a in k[17:20]]
        d.n_lights = len(s.light_pos)
        d.light_pos, d.light_color = _p(k[20]), _p(k[21])
        d.n_cones = len(s.cone_object)
        d.cone_M, d.cone_Minv, d.cone_N, d.cone_object = [_p(a) for a in k[22:26]]
        for i in range(3):
            d.ambient[i] = float(s.ambient[i])
        self._bmps = []
        for i, b in enumerate(s.bmp):
            if b is not None:
                b = np.ascontiguousarray(b)
                self._bmps.append(b)
                d.bmp_h[i], d.bmp_w[i], d.bmp_c[i] = b.shape
                d.bmp[i] = b.ctypes.data
        self._h = lib().ro_create(C.byref(d))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ro_destroy(self._h)
            self._h = None

    def render(self, width, height, spp=4, max_depth=3, seed=0, jitter_mode=0, camera=None, threads=0,
               tile_rank=0, tile_world=1):
        p = N.RenderParams(width=width, height=height, spp=spp, max_depth=max_depth, seed=seed,
                           jitter_mode=jitter_mode, tile_rank=tile_rank, tile_world=tile_world)
        cam = camera if camera is not None else self.scene.camera
        rgb8 = np.zeros((height, width, 3), np.uint8)
        raw = np.zeros((height, width, 3), np.int32)
        rgbf = np.zeros((height, width, 3), np.float32)
        prim = np.full((height, width), N.PRIM_MISS, np.uint32)
        st = N.Stats()
        rc = lib().ro_render(self._h, C.byref(cam), C.byref(p), _p(rgb8), _p(raw), _p(rgbf), _p(prim), C.byref(st), threads)
        if rc:
            raise RuntimeError("ro_render failed")
        return dict(rgb8=rgb8, rgb8_i32=raw, rgbf=rgbf, primid=prim, stats=st.as_dict())

    def trace_batch(self, rays, depth=3, threads=0):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = len(rays)
        col = np.zeros((n, 3), np.float32)
        hits = (N.Hit * n)()
        sh = np.zeros((n, 4), np.uint8)
        st = N.Stats()
        lib().ro_trace_batch(self._h, n, _p(rays), depth, _p(col), C.cast(hits, C.c_void_p), _p(sh), C.byref(st), threads)
        return col, hits_to_arrays(hits), sh, st.as_dict()

    def eval_texture(self, tex_id, uv):
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        out = np.zeros((len(uv), 3), np.float32)
        lib().ro_eval_texture(self._h, tex_id, len(uv), _p(uv), _p(out))
        return out

    def walk_counts(self, nodes, tri_order, info, rays, slot_tests=None):
        """node steps / triangle tests / closest BVH slot per ray of the KERNEL's closest-hit walk over the committed tree
        (nodes, tri_order: Renderer.get_bvh(); info: Renderer.get_bvh_info()), restated on the CPU (ro_walk_counts)"""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        nodes = np.ascontiguousarray(nodes, np.uint32)
        tri_order = np.ascontiguousarray(tri_order, np.uint32)
        tri_object = np.ascontiguousarray(self.scene.tri_object, np.int32)
        rank = self.reference_order()
        n = len(rays)
        on, ot, sl = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.int64)
        lib().ro_walk_counts(self._h, info["root_ref"], info["bound"], len(nodes), _p(nodes), _p(tri_order), _p(tri_object), _p(rank),
                             n, _p(rays), _p(on), _p(ot), _p(sl), _p(slot_tests))
        return on, ot, sl

    def reference_order(self):
        rank = np.zeros(self.scene.n_triangles, np.uint32)
        lib().ro_reference_order(self._h, _p(rank))
        return rank


def hits_to_arrays(hits):
    """(N.Hit * n) -> dict of numpy arrays"""
    n = len(hits)
    raw = np.frombuffer(hits, dtype=np.uint8).reshape(n, 96)
    i32 = raw[:, :12].copy().view(np.int32).reshape(n, 3)
    f = raw[:, 16:88].copy().view(np.float32).reshape(n, 18)
    return dict(hit=i32[:, 0], object=i32[:, 1], triangle=i32[:, 2], prim=raw[:, 12:16].copy().view(np.uint32).reshape(n),
                distance=f[:, 0], intersection=f[:, 1:4], normal=f[:, 4:7], normal_shading=f[:, 7:10], uv=f[:, 10:12],
                tangent=f[:, 12:15], bitangent=f[:, 15:18])
