"""Python host of the HIP trace loop: a thin object over the C ABI (include/rt_hip.h).

``Renderer`` owns one rt_context (one GPU).  Scenes go in as FlatScene arrays (raytracer_amd.scene); images,
hit records and texture values come back as numpy arrays.  The device-side entry points that multi-GPU rendering
needs (tile-major device buffers, assembly) take raw device pointers so torch tensors can be used for the RCCL
gather (raytracer_amd.dist) -- torch is plumbing here, never a compute path.
"""
import ctypes as C

import numpy as np

from . import _native as N
from .scene import FlatScene


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class RtError(RuntimeError):
    pass


class Renderer:
    def __init__(self, device=0, _borrowed_ctx=None):
        self.lib = N.hip_lib()
        self._ctx = C.c_void_p()
        self._owns = _borrowed_ctx is None
        if _borrowed_ctx is None:
            self._check(self.lib.rt_create(int(device), C.byref(self._ctx)))
        else:
            self._ctx = C.c_void_p(_borrowed_ctx)  # a rank of an rt_multi: destroyed with it
        self.device = int(device)
        self.scene = None

    def _check(self, rc):
        if rc != 0:
            raise RtError(self.lib.rt_last_error().decode())

    def close(self):
        if getattr(self, "_ctx", None):
            if self._owns:
                self.lib.rt_destroy(self._ctx)
            self._ctx = None

    def commit_stats(self):
        """where rt_commit spent its time: dict(n_triangles, bvh_nodes, ms_bvh_build, ms_reference_order, ms_records, ms_upload)"""
        st = N.CommitStats()
        self._check(self.lib.rt_get_commit_stats(self._ctx, C.byref(st)))
        return st.as_dict()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene ---------------------------------------------------------------------------------------
    def set_scene(self, s: FlatScene, commit=True):
        L, ctx = self.lib, self._ctx
        c = np.ascontiguousarray
        mats = c(s.materials)
        self._check(L.rt_set_materials(ctx, len(mats), _p(mats)))
        self._check(L.rt_set_object_materials(ctx, s.n_objects, _p(c(s.object_material))))
        arrs = [c(a, np.float32) for a in (s.v0, s.v1, s.v2, s.nA, s.nB, s.nC, s.tA, s.tB, s.tC)]
        self._check(L.rt_set_triangles(ctx, s.n_triangles, *[_p(a) for a in arrs], _p(c(s.tri_flags)), _p(c(s.tri_object))))
        self._check(L.rt_set_spheres(ctx, len(s.sph_object), _p(c(s.sph_M)), _p(c(s.sph_Minv)), _p(c(s.sph_N)), _p(c(s.sph_object))))
        self._check(L.rt_set_cones(ctx, len(s.cone_object), _p(c(s.cone_M)), _p(c(s.cone_Minv)), _p(c(s.cone_N)), _p(c(s.cone_object))))
        self._check(L.rt_set_planes(ctx, len(s.pl_object), _p(c(s.pl_point)), _p(c(s.pl_normal)), _p(c(s.pl_object))))
        self._check(L.rt_set_lights(ctx, len(s.light_pos), _p(c(s.light_pos)), _p(c(s.light_color)), _p(c(s.ambient, np.float32))))
        for slot, b in enumerate(s.bmp):
            if b is not None:
                b = c(b)
                self._check(L.rt_set_bmp(ctx, slot, b.shape[1], b.shape[0], b.shape[2], _p(b)))
        if commit:
            self._check(L.rt_commit(ctx))
        self.scene = s
        return self

    # ---- rendering -----------------------------------------------------------------------------------
    def _params(self, width, height, spp, max_depth, seed, jitter_mode, rank, world, chunk_samples, collect_stats):
        return N.RenderParams(width=width, height=height, spp=spp, max_depth=max_depth, seed=seed,
                              jitter_mode=jitter_mode, tile_rank=rank, tile_world=world,
                              chunk_samples=chunk_samples, collect_stats=collect_stats)

    def render(self, width, height, spp=4, max_depth=3, seed=0, jitter_mode=N.JITTER_AUTO, camera=None,
               chunk_samples=0, collect_stats=0, want=("rgb8", "rgbf", "primid"), single_frame=False):
        """Whole image on this GPU.  Returns dict(rgb8 u8[h,w,3], rgbf f32[h,w,3], primid u32[h,w], stats).
        single_frame: RT_RENDER_SINGLE_FRAME (no queue growth inside the frame or for a repeat of it)."""
        p = self._params(width, height, spp, max_depth, seed, jitter_mode, 0, 1, chunk_samples, collect_stats)
        p.flags = 1 if single_frame else 0
        cam = camera if camera is not None else self.scene.camera
        out = {}
        rgb8 = np.zeros((height, width, 3), np.uint8) if "rgb8" in want else None
        rgbf = np.zeros((height, width, 3), np.float32) if "rgbf" in want else None
        prim = np.zeros((height, width), np.uint32) if "primid" in want else None
        st = N.Stats()
        self._check(self.lib.rt_render(self._ctx, C.byref(cam), C.byref(p), _p(rgb8), _p(rgbf), _p(prim), C.byref(st)))
        out.update(rgb8=rgb8, rgbf=rgbf, primid=prim, stats=st.as_dict())
        return out

    def tile_counts(self, width, height, rank, world):
        a, b = C.c_int64(), C.c_int64()
        self._check(self.lib.rt_tile_counts(width, height, rank, world, C.byref(a), C.byref(b)))
        return a.value, b.value

    def render_tiles_device(self, width, height, rank, world, d_rgb8=0, d_rgbf=0, d_primid=0, stream=0, spp=4,
                            max_depth=3, seed=0, jitter_mode=N.JITTER_AUTO, camera=None, chunk_samples=0,
                            collect_stats=0):
        """This rank's interleaved tiles into device buffers (raw pointers); returns the stats dict."""
        p = self._params(width, height, spp, max_depth, seed, jitter_mode, rank, world, chunk_samples, collect_stats)
        cam = camera if camera is not None else self.scene.camera
        st = N.Stats()
        self._check(self.lib.rt_render_tiles_device(self._ctx, C.byref(cam), C.byref(p), C.c_void_p(d_rgb8 or None),
                                                    C.c_void_p(d_rgbf or None), C.c_void_p(d_primid or None),
                                                    C.c_void_p(stream or None), C.byref(st)))
        return st.as_dict()

    def assemble_tiles_device(self, width, height, world, tiles_per_rank, t8=0, tf=0, tp=0, o8=0, of=0, op=0, stream=0):
        v = lambda x: C.c_void_p(x or None)
        self._check(self.lib.rt_assemble_tiles_device(self._ctx, width, height, world, tiles_per_rank, v(t8), v(tf), v(tp),
                                                      v(o8), v(of), v(op), v(stream)))

    # ---- trace_ray-style API and known-answer hooks -----------------------------------------------------
    def eval_texture(self, tex_id, uv):
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        out = np.zeros((len(uv), 3), np.float32)
        self._check(self.lib.rt_eval_texture(self._ctx, int(tex_id), len(uv), _p(uv), _p(out)))
        return out

    def eval_perlin(self, pts):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
        out = np.zeros(len(pts), np.float32)
        self._check(self.lib.rt_eval_perlin(self._ctx, len(pts), _p(pts), _p(out)))
        return out

    def get_bvh(self):
        nn, nt = C.c_int64(), C.c_int64()
        self._check(self.lib.rt_get_bvh(self._ctx, C.byref(nn), None, C.byref(nt), None))
        nodes = np.zeros((nn.value, 16), np.uint32)
        order = np.zeros(nt.value, np.uint32)
        self._check(self.lib.rt_get_bvh(self._ctx, C.byref(nn), _p(nodes), C.byref(nt), _p(order)))
        return nodes, order

    def get_bvh_info(self):
        """root reference, error-bound scale, deepest stack and LDS stack entries of the committed tree (rt_get_bvh_info)"""
        root, bound, ms, le = C.c_uint32(), C.c_float(), C.c_int32(), C.c_int32()
        self._check(self.lib.rt_get_bvh_info(self._ctx, C.byref(root), C.byref(bound), C.byref(ms), C.byref(le)))
        return dict(root_ref=root.value, bound=bound.value, max_stack=ms.value, lds_stack_entries=le.value)


def hits_to_arrays(hits):
    """(N.Hit * n) -> dict of numpy arrays (field names of include/rt_hip.h: rt_hit)"""
    n = len(hits)
    raw = np.frombuffer(hits, dtype=np.uint8).reshape(n, 96)
    i32 = raw[:, :12].copy().view(np.int32).reshape(n, 3)
    f = raw[:, 16:88].copy().view(np.float32).reshape(n, 18)
    return dict(hit=i32[:, 0], object=i32[:, 1], triangle=i32[:, 2],
                prim=raw[:, 12:16].copy().view(np.uint32).reshape(n), distance=f[:, 0], intersection=f[:, 1:4],
                normal=f[:, 4:7], normal_shading=f[:, 7:10], uv=f[:, 10:12], tangent=f[:, 12:15],
                bitangent=f[:, 15:18])


def _trace_batch(self, rays, depth=3, want_hits=True, want_shadow=True):
    """trace_ray(ray, depth) for a batch: (colors f32[n,3], hits dict or None, shadow u8[n,4] or None, stats)"""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    n = len(rays)
    col = np.zeros((n, 3), np.float32)
    hits = (N.Hit * max(n, 1))() if want_hits else None
    sh = np.zeros((n, 4), np.uint8) if want_shadow else None
    st = N.Stats()
    self._check(self.lib.rt_trace_batch(self._ctx, n, _p(rays), int(depth), _p(col),
                                        C.cast(hits, C.c_void_p) if want_hits else None, _p(sh), C.byref(st)))
    h = hits_to_arrays(hits) if want_hits else None
    if h is not None and n == 0:
        h = {k: v[:0] for k, v in h.items()}
    return col, h, sh, st.as_dict()


Renderer.trace_batch = _trace_batch


class MultiRenderer:
    """One frame over N GPUs of this process through the C ABI's rt_multi (include/rt_hip.h): interleaved tiles, one host
    thread per GPU inside the library, RCCL send/recv gather at frame end.  `devices` may repeat a device (a rehearsal of
    N ranks on one card: those exchange by device copies); rehearse_rccl pushes a single rank's tiles through RCCL."""

    def __init__(self, devices, rehearse_rccl=False):
        self.lib = N.hip_lib()
        self._m = C.c_void_p()
        ids = (C.c_int32 * len(devices))(*devices)
        if self.lib.rt_multi_create(ids, len(devices), 1 if rehearse_rccl else 0, C.byref(self._m)):
            raise RtError(self.lib.rt_last_error().decode())
        self.n = self.lib.rt_multi_size(self._m)
        self.transport = self.lib.rt_multi_transport(self._m).decode()
        self.rank0 = Renderer(devices[0], _borrowed_ctx=self.lib.rt_multi_context(self._m, 0))
        self.scene = None

    def _check(self, rc):
        if rc != 0:
            raise RtError(self.lib.rt_last_error().decode())

    def set_scene(self, s: FlatScene):
        self.rank0.set_scene(s, commit=False)       # the scene goes to rank 0 only ...
        self._check(self.lib.rt_multi_commit(self._m))  # ... one BVH build, uploaded to every GPU
        self.scene = s
        return self

    def render(self, width, height, spp=4, max_depth=3, seed=0, jitter_mode=N.JITTER_AUTO, camera=None, chunk_samples=0,
               collect_stats=0, want=("rgb8", "rgbf", "primid")):
        p = N.RenderParams(width=width, height=height, spp=spp, max_depth=max_depth, seed=seed, jitter_mode=jitter_mode,
                           tile_rank=0, tile_world=1, chunk_samples=chunk_samples, collect_stats=collect_stats)
        cam = camera if camera is not None else self.scene.camera
        rgb8 = np.zeros((height, width, 3), np.uint8) if "rgb8" in want else None
        rgbf = np.zeros((height, width, 3), np.float32) if "rgbf" in want else None
        prim = np.zeros((height, width), np.uint32) if "primid" in want else None
        st = (N.Stats * self.n)()
        self._check(self.lib.rt_multi_render(self._m, C.byref(cam), C.byref(p), _p(rgb8), _p(rgbf), _p(prim), C.cast(st, C.c_void_p)))
        t = N.MultiTimings()
        self._check(self.lib.rt_multi_last_timings(self._m, C.byref(t)))
        return dict(rgb8=rgb8, rgbf=rgbf, primid=prim, stats=[x.as_dict() for x in st], timings=t.as_dict())

    def close(self):
        if getattr(self, "_m", None):
            self.rank0.close()
            self.lib.rt_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
