"""Flattened scenes as numpy arrays -- the argument lists of rt_set_* (include/rt_hip.h).

``FlatScene.builtin(name)`` builds one of the shipped scenes through the C++ scene-definition surface
(raytracer_amd/host, mirroring the reference's Objects.h / Material.h / MeshLoader.h / scene functions) and
copies the SoA buffers out.  The same arrays feed the HIP renderer and, in tests, the CPU oracle.
"""
import ctypes as C
import os

import numpy as np

from . import _native as N


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class FlatScene:
    def __init__(self):
        self.materials = np.zeros(0, dtype=np.dtype((np.void, 96)))
        self.object_material = np.zeros(0, np.int32)
        z3, z2 = np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32)
        self.v0 = self.v1 = self.v2 = self.nA = self.nB = self.nC = z3
        self.tA = self.tB = self.tC = z2
        self.tri_flags = np.zeros(0, np.uint32)
        self.tri_object = np.zeros(0, np.int32)
        self.sph_M = self.sph_Minv = self.sph_N = np.zeros((0, 16), np.float32)
        self.sph_object = np.zeros(0, np.int32)
        self.cone_M = self.cone_Minv = self.cone_N = np.zeros((0, 16), np.float32)
        self.cone_object = np.zeros(0, np.int32)
        self.pl_point = self.pl_normal = z3
        self.pl_object = np.zeros(0, np.int32)
        self.light_pos = self.light_color = z3
        self.ambient = np.zeros(3, np.float32)
        self.bmp = [None, None, None, None]  # (h, w, c) uint8, top-down BGR
        self.camera = N.Camera()
        self.name = "custom"

    # ------------------------------------------------------------------------------------------------
    @property
    def n_triangles(self):
        return int(self.tri_flags.shape[0])

    @property
    def n_objects(self):
        return int(self.object_material.shape[0])

    def material_structs(self):
        return (N.Material * len(self.materials)).from_buffer_copy(self.materials.tobytes())

    def mesh_first_triangle(self):
        """first global triangle index of every mesh object: {object_id: first}"""
        out = {}
        for k, o in enumerate(self.tri_object.tolist()):
            out.setdefault(o, k)
        return out

    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def read_bmp(path, lazy=True):
        lib = N.host_lib(lazy=lazy)
        w, h, c = C.c_int32(), C.c_int32(), C.c_int32()
        if lib.rth_read_bmp(path.encode(), C.byref(w), C.byref(h), C.byref(c), None):
            raise RuntimeError(lib.rth_last_error().decode())
        buf = np.zeros((h.value, w.value, c.value), np.uint8)
        lib.rth_read_bmp(path.encode(), C.byref(w), C.byref(h), C.byref(c), _ptr(buf))
        return buf

    @classmethod
    def _from_handle(cls, lib, h, counts):
        s = cls()
        nt, ns, npl, nm, no, nl = (int(x) for x in counts[:6])
        f3 = lambda n: np.zeros((n, 3), np.float32)
        f2 = lambda n: np.zeros((n, 2), np.float32)
        s.v0, s.v1, s.v2, s.nA, s.nB, s.nC = (f3(nt) for _ in range(6))
        s.tA, s.tB, s.tC = (f2(nt) for _ in range(3))
        s.tri_flags = np.zeros(nt, np.uint32)
        s.tri_object = np.zeros(nt, np.int32)
        lib.rth_scene_get_triangles(h, *[_ptr(a) for a in (s.v0, s.v1, s.v2, s.nA, s.nB, s.nC, s.tA, s.tB, s.tC,
                                                           s.tri_flags, s.tri_object)])
        s.sph_M, s.sph_Minv, s.sph_N = (np.zeros((ns, 16), np.float32) for _ in range(3))
        s.sph_object = np.zeros(ns, np.int32)
        s.pl_point, s.pl_normal = f3(npl), f3(npl)
        s.pl_object = np.zeros(npl, np.int32)
        lib.rth_scene_get_analytic(h, *[_ptr(a) for a in (s.sph_M, s.sph_Minv, s.sph_N, s.sph_object, s.pl_point,
                                                          s.pl_normal, s.pl_object)])
        nc = int(counts[7])
        s.cone_M, s.cone_Minv, s.cone_N = (np.zeros((nc, 16), np.float32) for _ in range(3))
        s.cone_object = np.zeros(nc, np.int32)
        lib.rth_scene_get_cones(h, *[_ptr(a) for a in (s.cone_M, s.cone_Minv, s.cone_N, s.cone_object)])
        s.materials = np.zeros(nm, dtype=np.dtype((np.void, 96)))
        s.object_material = np.zeros(no, np.int32)
        s.light_pos, s.light_color = f3(nl), f3(nl)
        s.ambient = np.zeros(3, np.float32)
        lib.rth_scene_get_shading(h, *[_ptr(a) for a in (s.materials, s.object_material, s.light_pos, s.light_color,
                                                         s.ambient)])
        return s

    @classmethod
    def builtin(cls, name, asset_dir=None, lazy=False):
        """name: 'sample' | 'bunny' | 'competition' | 'cones' | 'stress:<replicas>' | 'stress-diagonal:<replicas>'"""
        asset_dir = asset_dir or N.ASSET_DIR
        lib = N.host_lib(lazy=lazy)
        h = lib.rth_scene_create(name.encode(), asset_dir.encode())
        if not h:
            raise RuntimeError(lib.rth_last_error().decode())
        try:
            counts = np.zeros(8, np.int64)
            lib.rth_scene_counts(h, _ptr(counts))
            s = cls._from_handle(lib, h, counts)
            lib.rth_scene_camera(h, C.byref(s.camera))
            if counts[6]:
                names = ["basecolor.bmp", "normal.bmp", "roughness.bmp", "ambientOcclusion.bmp"]
                s.bmp = [cls.read_bmp(os.path.join(asset_dir, "textures", n), lazy=lazy) for n in names]
        finally:
            lib.rth_scene_destroy(h)
        s.name = name
        return s

    @classmethod
    def mesh(cls, path, translation=(0.0, 0.0, 0.0), lazy=False):
        """one .obj parsed with MeshLoader semantics; returns (FlatScene with triangles only, counts dict)"""
        lib = N.host_lib(lazy=lazy)
        counts = np.zeros(4, np.int64)
        h = lib.rth_mesh_create(path.encode(), *[float(t) for t in translation], _ptr(counts))
        if not h:
            raise RuntimeError(lib.rth_last_error().decode())
        try:
            c8 = np.zeros(8, np.int64)
            c8[0] = counts[0]
            s = cls._from_handle(lib, h, c8)
        finally:
            lib.rth_scene_destroy(h)
        return s, dict(triangles=int(counts[0]), vertices=int(counts[1]), normals=int(counts[2]), uvs=int(counts[3]))
