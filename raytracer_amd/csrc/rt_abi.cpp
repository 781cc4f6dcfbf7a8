// rt_abi.cpp -- implementation of the C ABI declared in include/rt_hip.h: context, scene upload, BVH commit, the
// chunked wavefront render loop and the test hooks.  Host code only; kernels live in rt_kernels.hip.
// There is no CPU rendering path in this library: every entry point that produces pixels, hits or texture
// values launches HIP kernels and fails if no device is usable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include "rt_bvh.h"
#include "rt_kernels.h"

#define RT_SCRATCH_MAX_DEVICES 64

namespace {

thread_local std::string g_err;
int fail(const std::string &msg) {
    g_err = msg;
    return 1;
}
#define HIPOK(expr)                                                                                   \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int alloc(size_t count) {
        if (count <= n && p) return 0;
        release();
        if (count == 0) return 0;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            return fail(std::string("hipMalloc of ") + std::to_string(count * sizeof(T)) + " bytes: " + hipGetErrorString(e));
        }
        n = count;
        return 0;
    }
    template <class A>
    int upload(const std::vector<T, A> &h) {
        if (alloc(std::max<size_t>(h.size(), 1))) return 1;
        if (!h.empty()) HIPOK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

// (vectors whose resize() leaves the new elements uninitialised: 8 GB of records for 50 M triangles are written once, by
// the threads that assemble them, instead of being zeroed by one thread first)
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
}  // namespace

struct rt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    // a frame's second pipeline runs on a stream of its own, forked from and joined to the caller's stream by these events
    // (created at rt_create: the render entry points create no stream or event)
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    RtLaunchCfg cfg;
    // host-side scene as handed over through rt_set_*
    std::vector<rt_material> materials;
    std::vector<int32_t> object_material;
    std::vector<float> v0, v1, v2, nA, nB, nC, tA, tB, tC;
    std::vector<uint32_t> tri_flags;
    std::vector<int32_t> tri_object;
    std::vector<RtSphere> spheres, cones;  // uploaded as one array (RtSphere::kind)
    std::vector<RtPlane> planes;
    std::vector<RtLight> lights;
    float ambient[3] = {0, 0, 0};
    std::vector<uint8_t> bmp[4];
    int bmp_w[4] = {0, 0, 0, 0}, bmp_h[4] = {0, 0, 0, 0}, bmp_c[4] = {0, 0, 0, 0};
    // committed scene
    bool committed = false;
    rt_commit_stats commit_stats{};
    uint64_t plane_general = 0;  // planes / scene with hits for the general k_shade (commitHost)
    bool has_general = false;
    struct { uint64_t n_nodes = 0, n_triangles = 0; } info;  // (a replica of rt_multi keeps no triangle arrays / tree of its own)
    RtBvh bvh;
    std::vector<int64_t> mesh_first;  // per object: first global triangle index (or -1)
    DevBuf<RtNode4> d_records;  // BVH nodes, then the triangle records (rt_types.h)
    DevBuf<uint32_t> d_spill;   // traversal stack entries beyond the LDS part: one area of spill_words per lane of a frame (rt_frame.inl)
    size_t spill_words = 0;
    DevBuf<RtNodeP> d_nodes_p;  // float-box form of the nodes (packet kernels)
    DevBuf<RtTriShade> d_shade;
    DevBuf<RtSphere> d_spheres;
    DevBuf<RtPlane> d_planes;
    DevBuf<rt_material> d_materials;
    DevBuf<int32_t> d_objmat;
    DevBuf<RtLight> d_lights;
    DevBuf<uint8_t> d_bmp[4];
    RtSceneDev dev{};
    // Wavefront workspace: ONE arena, carved anew for every frame (carveQueues).  It is allocated once -- the whole budget
    // (RT_WORKSPACE_GB, 30 GiB by default) for a frame that needs more than one chunk, what the frame needs otherwise -- and
    // grows only when a later call cannot fit a single tile into it or asks for a chunk size of its own.
    DevBuf<unsigned char> d_arena;
    // Fan-out observed on this scene since rt_commit: queue entries per primary sample of a chunk that one iteration of the
    // pipeline holds in steady state (all levels together; shadow rays over all lights), without the slab slack, and the
    // deepest bounce budget it was seen at.  Frames are planned from it once it exists (planFrame); reset by rt_commit.
    bool obs_valid = false;
    double obs_ray = 0.0, obs_sh = 0.0;
    double obs_l0_share = 0.0;  // share of the last frame's shadow rays that were level-0 rays
    int obs_depth = -1;
    // the library's static scratch on this context's device (rt_render_tiles_static); resolved at rt_create
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // tile-major outputs of rt_render
    DevBuf<uint8_t> d_t8, d_o8;
    DevBuf<float> d_tf, d_of;
    DevBuf<uint32_t> d_tp, d_op;
};

namespace {

int bind(rt_context *ctx) {
    HIPOK(hipSetDevice(ctx->device));
    return 0;
}

double nowMs();

// the frame pipeline: workspace carving, planning, the iteration loop (renderTiles, renderTilesStatic, traceBatch)
#include "rt_frame.inl"

}  // namespace

extern "C" {

const char *rt_last_error(void) { return g_err.c_str(); }

int rt_create(int device_id, rt_context **out) {
    if (!out) return fail("rt_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail("rt_create: no HIP device available (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail("rt_create: device_id out of range");
    HIPOK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPOK(hipGetDeviceProperties(&prop, device_id));
    rt_context *ctx = new rt_context();
    ctx->device = device_id;
    ctx->cfg.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
        if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
        if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
        delete ctx;
        return fail("rt_create: hipStreamCreate / hipEventCreate failed");
    }
    {   // address of the static scratch on THIS device (looked up once per device, with that device current)
        static std::mutex mu;
        static void *table[RT_SCRATCH_MAX_DEVICES];
        static size_t bytes = 0;
        std::lock_guard<std::mutex> g(mu);
        if (device_id >= RT_SCRATCH_MAX_DEVICES) {
            (void)hipStreamDestroy(ctx->stream);
            delete ctx;
            return fail("rt_create: device_id beyond the static-scratch table");
        }
        if (!table[device_id]) {
            const char *why = "";
            if (rtScratchResolve(&table[device_id], &bytes, &why)) {
                table[device_id] = nullptr;
                (void)hipStreamDestroy(ctx->stream);
                delete ctx;
                return fail(std::string("rt_create: hipGetSymbolAddress(rt_static_scratch): ") + why);
            }
        }
        ctx->scratch = table[device_id];
        ctx->scratch_bytes = bytes;
    }
    *out = ctx;
    return 0;
}

int rt_destroy(rt_context *ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    ctx->d_records.release(); ctx->d_spill.release(); ctx->d_nodes_p.release(); ctx->d_arena.release(); ctx->d_shade.release(); ctx->d_spheres.release(); ctx->d_planes.release();
    ctx->d_materials.release(); ctx->d_objmat.release(); ctx->d_lights.release();
    for (int k = 0; k < 4; k++) ctx->d_bmp[k].release();
    ctx->d_t8.release(); ctx->d_o8.release(); ctx->d_tf.release(); ctx->d_of.release(); ctx->d_tp.release(); ctx->d_op.release();
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    delete ctx;
    return 0;
}

int rt_set_materials(rt_context *ctx, int32_t n, const rt_material *m) {
    if (!ctx || n < 0 || (n && !m)) return fail("rt_set_materials: bad arguments");
    for (int k = 0; k < n; k++) {
        const int ids[4] = {m[k].texture, m[k].normal_map, m[k].roughness, m[k].occlusion};
        for (int id : ids)
            if (id < 0 || id >= RT_TEX_COUNT) return fail("rt_set_materials: unknown texture id");
        if (m[k].has_normal_map && m[k].normal_map == RT_TEX_NONE) return fail("rt_set_materials: has_normal_map without normal_map");
        if (m[k].has_img_texture && (m[k].roughness == RT_TEX_NONE || m[k].occlusion == RT_TEX_NONE))
            return fail("rt_set_materials: has_img_texture without roughness/occlusion");
    }
    ctx->materials.assign(m, m + n);
    ctx->committed = false;
    return 0;
}

int rt_set_object_materials(rt_context *ctx, int32_t n, const int32_t *om) {
    if (!ctx || n < 0 || (n && !om)) return fail("rt_set_object_materials: bad arguments");
    ctx->object_material.assign(om, om + n);
    ctx->committed = false;
    return 0;
}

int rt_set_triangles(rt_context *ctx, int64_t n, const float *v0, const float *v1, const float *v2, const float *nA,
                     const float *nB, const float *nC, const float *tA, const float *tB, const float *tC,
                     const uint32_t *flags, const int32_t *object_id) {
    if (!ctx || n < 0) return fail("rt_set_triangles: bad arguments");
    if (n >= (1ll << 27)) return fail("rt_set_triangles: at most 2^27-1 triangles (28-bit record index: nodes + triangles)");
    if (n && (!v0 || !v1 || !v2 || !flags || !object_id)) return fail("rt_set_triangles: NULL vertex/flags/object arrays");
    for (int64_t k = 1; k < n; k++)  // (validated before anything of the context changes)
        if (object_id[k] < object_id[k - 1]) return fail("rt_set_triangles: object_id must be non-decreasing");
    auto put = [&](std::vector<float> &dst, const float *src, int w) {
        dst.assign((size_t)n * w, 0.0f);
        if (src && n) std::memcpy(dst.data(), src, (size_t)n * w * sizeof(float));
    };
    put(ctx->v0, v0, 3); put(ctx->v1, v1, 3); put(ctx->v2, v2, 3);
    put(ctx->nA, nA, 3); put(ctx->nB, nB, 3); put(ctx->nC, nC, 3);
    put(ctx->tA, tA, 2); put(ctx->tB, tB, 2); put(ctx->tC, tC, 2);
    ctx->tri_flags.assign(flags, flags + n);
    ctx->tri_object.assign(object_id, object_id + n);
    ctx->committed = false;
    return 0;
}

int rt_set_spheres(rt_context *ctx, int32_t n, const float *M, const float *Minv, const float *Nmat, const int32_t *object_id) {
    if (!ctx || n < 0 || n > RT_MAX_SPHERES || (n && (!M || !Minv || !Nmat || !object_id))) return fail("rt_set_spheres: bad arguments (at most 64 spheres)");
    ctx->spheres.resize(n);
    for (int k = 0; k < n; k++) {
        RtSphere &s = ctx->spheres[k];
        std::memcpy(s.M, M + 16 * k, 64);
        std::memcpy(s.Minv, Minv + 16 * k, 64);
        std::memcpy(s.N, Nmat + 16 * k, 64);
        s.object = object_id[k];
        s.material = 0;
        for (int a = 0; a < 3; a++) {  // extent of M * unit ball along axis a = |row a of the linear part|
            const double r = std::sqrt((double)s.M[a] * s.M[a] + (double)s.M[4 + a] * s.M[4 + a] + (double)s.M[8 + a] * s.M[8 + a]);
            const double c = s.M[12 + a], pad = 1e-3 * r + 1e-5 * std::fabs(c) + 1e-30;
            s.box_mn[a] = (float)(c - r - pad);
            s.box_mx[a] = (float)(c + r + pad);
        }
        s.kind = RT_KIND_SPHERE;
        std::memset(s._pad, 0, sizeof s._pad);
    }
    ctx->committed = false;
    return 0;
}

int rt_set_cones(rt_context *ctx, int32_t n, const float *M, const float *Minv, const float *Nmat, const int32_t *object_id) {
    if (!ctx || n < 0 || n > RT_MAX_SPHERES || (n && (!M || !Minv || !Nmat || !object_id))) return fail("rt_set_cones: bad arguments (at most 64 spheres + cones)");
    ctx->cones.resize(n);
    for (int k = 0; k < n; k++) {
        RtSphere &s = ctx->cones[k];
        std::memset(&s, 0, sizeof s);
        std::memcpy(s.M, M + 16 * k, 64);
        std::memcpy(s.Minv, Minv + 16 * k, 64);
        std::memcpy(s.N, Nmat + 16 * k, 64);
        s.object = object_id[k];
        s.kind = RT_KIND_CONE;
        // world bounds of the object-space box [-1,1] x [0,1] x [-1,1] that holds the unit cone and its cap
        double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int c = 0; c < 8; c++) {
            const double x = (c & 1) ? 1.0 : -1.0, y = (c & 2) ? 1.0 : 0.0, z = (c & 4) ? 1.0 : -1.0;
            for (int ax = 0; ax < 3; ax++) {
                const double v = s.M[ax] * x + s.M[4 + ax] * y + s.M[8 + ax] * z + s.M[12 + ax];
                mn[ax] = std::fmin(mn[ax], v);
                mx[ax] = std::fmax(mx[ax], v);
            }
        }
        for (int ax = 0; ax < 3; ax++) {
            const double pad = 1e-3 * (mx[ax] - mn[ax]) + 1e-5 * std::fmax(std::fabs(mn[ax]), std::fabs(mx[ax])) + 1e-30;
            s.box_mn[ax] = (float)(mn[ax] - pad);
            s.box_mx[ax] = (float)(mx[ax] + pad);
        }
    }
    ctx->committed = false;
    return 0;
}

int rt_set_planes(rt_context *ctx, int32_t n, const float *point, const float *normal, const int32_t *object_id) {
    if (!ctx || n < 0 || n > RT_MAX_PLANES || (n && (!point || !normal || !object_id))) return fail("rt_set_planes: bad arguments (at most 64 planes)");
    ctx->planes.resize(n);
    for (int k = 0; k < n; k++) {
        RtPlane &p = ctx->planes[k];
        std::memcpy(p.point, point + 3 * k, 12);
        std::memcpy(p.normal, normal + 3 * k, 12);
        p.object = object_id[k];
        p.material = 0;
    }
    ctx->committed = false;
    return 0;
}

int rt_set_lights(rt_context *ctx, int32_t n, const float *position, const float *color, const float *ambient) {
    if (!ctx || n < 0 || n > RT_MAX_LIGHTS || (n && (!position || !color))) return fail("rt_set_lights: bad arguments (at most 16 lights)");
    ctx->lights.resize(n);
    for (int k = 0; k < n; k++) {
        RtLight &l = ctx->lights[k];
        std::memcpy(l.pos, position + 3 * k, 12);
        std::memcpy(l.color, color + 3 * k, 12);
        l._p0 = l._p1 = 0;
    }
    if (ambient) std::memcpy(ctx->ambient, ambient, 12);
    ctx->committed = false;
    return 0;
}

int rt_set_bmp(rt_context *ctx, int32_t slot, int32_t w, int32_t h, int32_t c, const uint8_t *bytes) {
    if (!ctx || slot < 0 || slot > 3 || w <= 0 || h <= 0 || c < 3 || c > 4 || !bytes) return fail("rt_set_bmp: bad arguments (24/32-bit images, slot 0..3)");
    if (w > 65535 || h > 65535) return fail("rt_set_bmp: image larger than the reference's uint16 texel index");
    ctx->bmp[slot].assign(bytes, bytes + (size_t)w * h * c);
    ctx->bmp_w[slot] = w; ctx->bmp_h[slot] = h; ctx->bmp_c[slot] = c;
    ctx->committed = false;
    return 0;
}

namespace {
double nowMs() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Host half of rt_commit: checks, BVH build, the reference's candidate order, the 64-byte record array.  The result
// is device independent, so a multi-GPU set-up (rt_multi_commit) builds it once and uploads it to every GPU.
struct CommitImage {
    std::vector<RtNode4, NoInitAlloc<RtNode4>> records;   // BVH nodes, then one record per triangle in leaf (slot) order
    std::vector<RtTriShade, NoInitAlloc<RtTriShade>> shade;
    std::vector<RtSphere> units;    // spheres, then cones: one device array
};

int commitHost(rt_context *ctx, CommitImage &img) {
    const double t0 = nowMs();
    const int n_obj = (int)ctx->object_material.size(), n_mat = (int)ctx->materials.size();
    const int64_t nt = (int64_t)ctx->tri_flags.size();
    for (int m : ctx->object_material)
        if (m < 0 || m >= n_mat) return fail("rt_commit: object refers to a material that was not set");
    auto objOk = [&](int o) { return o >= 0 && o < n_obj; };
    for (int64_t k = 0; k < nt; k++)
        if (!objOk(ctx->tri_object[k])) return fail("rt_commit: triangle object_id out of range");
    if (ctx->spheres.size() + ctx->cones.size() > RT_MAX_SPHERES) return fail("rt_commit: more than 64 spheres + cones");
    for (std::vector<RtSphere> *v : {&ctx->spheres, &ctx->cones})
        for (RtSphere &s : *v) {
            if (!objOk(s.object)) return fail("rt_commit: sphere / cone object_id out of range");
            s.material = ctx->object_material[s.object];
        }
    img.units = ctx->spheres;
    img.units.insert(img.units.end(), ctx->cones.begin(), ctx->cones.end());
    for (RtPlane &p : ctx->planes) {
        if (!objOk(p.object)) return fail("rt_commit: plane object_id out of range");
        p.material = ctx->object_material[p.object];
    }
    bool needs_bmp[4] = {false, false, false, false};
    for (const rt_material &m : ctx->materials) {
        const int ids[4] = {m.texture, m.has_normal_map ? m.normal_map : 0, m.has_img_texture ? m.roughness : 0, m.has_img_texture ? m.occlusion : 0};
        for (int id : ids)
            if (id >= RT_TEX_BMP_COLOR) needs_bmp[id - RT_TEX_BMP_COLOR] = true;
    }
    for (int k = 0; k < 4; k++)
        if (needs_bmp[k] && ctx->bmp_w[k] == 0) return fail("rt_commit: a material samples a BMP map that was not set (rt_set_bmp)");

    // materials whose hits the fast k_shade leaves to the general instance (rt_kernels.hip): BMP maps, anisotropic branch
    std::vector<uint8_t> general_mat((size_t)n_mat, 0);
    for (int k = 0; k < n_mat; k++) {
        const rt_material &m = ctx->materials[k];
        general_mat[k] = m.is_anisotropic || m.has_img_texture || m.texture >= RT_TEX_BMP_COLOR ||
                         (m.has_normal_map && m.normal_map >= RT_TEX_BMP_COLOR);
    }
    ctx->plane_general = 0;
    for (size_t k = 0; k < ctx->planes.size(); k++)
        if (general_mat[(size_t)ctx->planes[k].material]) ctx->plane_general |= 1ull << k;
    ctx->has_general = !img.units.empty() || ctx->plane_general != 0;
    for (int64_t k = 0; k < nt && !ctx->has_general; k++)
        if (general_mat[(size_t)ctx->object_material[(size_t)ctx->tri_object[k]]]) ctx->has_general = true;

    // acceleration structure + reordered triangle records
    rtBuildBvh(nt, ctx->v0.data(), ctx->v1.data(), ctx->v2.data(), ctx->bvh);
    const double t1 = nowMs();
    std::vector<uint32_t> rank((size_t)nt);
    rtReferenceOrder(nt, ctx->v0.data(), ctx->v1.data(), ctx->v2.data(), ctx->tri_object.data(), rank.data());
    const double t2 = nowMs();
    ctx->mesh_first.assign(n_obj, -1);
    for (int64_t k = nt - 1; k >= 0; k--) ctx->mesh_first[ctx->tri_object[k]] = k;
    // record array: nodes first, then one 64-byte record per triangle in leaf (slot) order
    const size_t nn = ctx->bvh.nodes.size();
    static_assert(sizeof(RtNode4) == 64 && sizeof(RtTriGeom) == 64, "records are 64 bytes");
    // (one record per SLOT: a large triangle that was cut into pieces before the build owns one slot per piece, rt_bvh.cpp)
    const size_t ns = ctx->bvh.order.size();
    if (nn + ns >= (1ull << 28) || ns >= (1ull << 27)) return fail("rt_commit: more than 2^28 BVH records");
    img.records.resize(nn + ns);
    if (nn) std::memcpy(img.records.data(), ctx->bvh.nodes.data(), nn * sizeof(RtNode4));
    RtTriGeom *geom = reinterpret_cast<RtTriGeom *>(img.records.data() + nn);
    img.shade.resize(ns);
    // (one contiguous piece of the slots per host thread: every record is a gather from the caller's arrays, 2.6 s on one
    // thread for 50 M triangles)
    auto assemble = [&](size_t slot0, size_t slot1) {
    for (size_t slot = slot0; slot < slot1; slot++) {
        const size_t t = ctx->bvh.order[slot];
        const float *A = &ctx->v0[3 * t], *B = &ctx->v1[3 * t], *C = &ctx->v2[3 * t];
        // Triangle ctor, MeshLoader.h:58: normal = normalize(cross(B - A, C - A)), glm operand order
        const float e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
        const float cx = e1[1] * e2[2] - e2[1] * e1[2], cy = e1[2] * e2[0] - e2[2] * e1[0], cz = e1[0] * e2[1] - e2[0] * e1[1];
        const float xx = cx * cx, yy = cy * cy, zz = cz * cz;
        const float inv = 1.0f / std::sqrt(xx + yy + zz);
        RtTriGeom &g = geom[slot];
        g.q0[0] = A[0]; g.q0[1] = A[1]; g.q0[2] = A[2]; g.q0[3] = cx * inv;
        g.q1[0] = B[0]; g.q1[1] = B[1]; g.q1[2] = B[2]; g.q1[3] = cy * inv;
        g.q2[0] = C[0]; g.q2[1] = C[1]; g.q2[2] = C[2]; g.q2[3] = cz * inv;
        g.object = (uint32_t)ctx->tri_object[t];
        g.rank = rank[t];
        g.prim = (uint32_t)t;
        g.general = general_mat[(size_t)ctx->object_material[(size_t)ctx->tri_object[t]]] ? RT_ID_GENERAL : 0u;
        RtTriShade &s = img.shade[slot];
        std::memcpy(s.nA, &ctx->nA[3 * t], 12); std::memcpy(s.nB, &ctx->nB[3 * t], 12); std::memcpy(s.nC, &ctx->nC[3 * t], 12);
        std::memcpy(s.tA, &ctx->tA[2 * t], 8); std::memcpy(s.tB, &ctx->tB[2 * t], 8); std::memcpy(s.tC, &ctx->tC[2 * t], 8);
        s.flags = ctx->tri_flags[t];
        s.object = ctx->tri_object[t];
        s.prim = (uint32_t)t;
        s.rank = rank[t];
        s.material = ctx->object_material[(size_t)ctx->tri_object[t]];
    }
    };
    {
        auto *fnp = &assemble;
        rtParallelFor(ns, [](size_t a, size_t e, void *arg) { (*static_cast<decltype(fnp)>(arg))(a, e); }, fnp);
    }
    const double t3 = nowMs();
    ctx->commit_stats = rt_commit_stats{};
    ctx->commit_stats.n_triangles = (uint64_t)nt;
    ctx->commit_stats.bvh_nodes = nn;
    ctx->commit_stats.ms_bvh_build = t1 - t0;
    ctx->commit_stats.ms_reference_order = t2 - t1;
    ctx->commit_stats.ms_records = t3 - t2;
    return 0;
}

// Device half of rt_commit: uploads `img` (built from `src`'s host-side scene; src == ctx for a single GPU) to ctx's GPU
int commitDevice(rt_context *ctx, const rt_context *src, const CommitImage &img) {
    const double t0 = nowMs();
    if (bind(ctx)) return 1;
    if (ctx != src) {  // replica: the small host-side tables the render paths read
        ctx->materials = src->materials;
        ctx->object_material = src->object_material;
        ctx->planes = src->planes;
        ctx->lights = src->lights;
        std::memcpy(ctx->ambient, src->ambient, sizeof ctx->ambient);
        ctx->mesh_first = src->mesh_first;
        ctx->bvh = RtBvh{};
        ctx->bvh.root_ref = src->bvh.root_ref; ctx->bvh.depth = src->bvh.depth; ctx->bvh.max_stack = src->bvh.max_stack;
        ctx->bvh.bound = src->bvh.bound;
        // (commit_stats: copied by rt_multi_commit once every upload has finished -- rank 0's thread writes src's meanwhile)
    }
    const size_t nn = src->bvh.nodes.size();
    const int64_t nt = (int64_t)src->tri_flags.size();
    ctx->info.n_nodes = nn;
    ctx->info.n_triangles = (uint64_t)nt;
    if (ctx->d_records.upload(img.records) || ctx->d_shade.upload(img.shade) || ctx->d_nodes_p.upload(src->bvh.nodes_p)) return 1;
    if (ctx->d_spheres.upload(img.units) || ctx->d_planes.upload(src->planes)) return 1;
    if (ctx->d_materials.upload(src->materials) || ctx->d_objmat.upload(src->object_material) || ctx->d_lights.upload(src->lights)) return 1;
    RtSceneDev &d = ctx->dev;
    std::memset(&d, 0, sizeof d);
    for (int k = 0; k < 4; k++) {
        if (src->bmp_w[k]) {
            if (ctx->d_bmp[k].upload(src->bmp[k])) return 1;
            d.bmp[k] = ctx->d_bmp[k].p;
        }
        d.bmp_w[k] = src->bmp_w[k]; d.bmp_h[k] = src->bmp_h[k]; d.bmp_c[k] = src->bmp_c[k];
    }
    d.nodes = ctx->d_records.p;
    d.tri_geom = reinterpret_cast<const RtTriGeom *>(ctx->d_records.p + nn);
    d.bound = src->bvh.bound;
    d.reach = src->bvh.bound;  // (+ the unit primitives' boxes: how far the finite geometry reaches, rtUnitCull)
    for (const RtSphere &u : img.units)
        for (int a = 0; a < 3; a++) d.reach = std::fmax(d.reach, std::fmax(std::fabs(u.box_mn[a]), std::fabs(u.box_mx[a])));
    d.nodes_p = ctx->d_nodes_p.p;
    d.tri_shade = ctx->d_shade.p;
    d.spheres = ctx->d_spheres.p;
    d.planes = ctx->d_planes.p;
    d.materials = ctx->d_materials.p;
    d.object_material = ctx->d_objmat.p;
    d.lights = ctx->d_lights.p;
    std::memcpy(d.ambient, src->ambient, 12);
    d.n_nodes = (int)nn;
    d.n_tris = (int)img.shade.size();  // triangle SLOTS (>= triangles: pre-split pieces)
    d.n_spheres = (int)img.units.size();
    d.n_planes = (int)src->planes.size();
    d.n_lights = (int)src->lights.size();
    d.n_objects = (int)src->object_material.size();
    d.n_materials = (int)src->materials.size();
    d.plane_general = src->plane_general;
    d.has_general = src->has_general ? 1 : 0;
    // traversal stack: up to RT_LDS_STACK rows per lane in LDS (16 KB per 256-thread block), the rest of the worst case
    // (sum of children - 1 along the deepest path; rarely reached) in a global spill area per grid thread
    const int entries = std::max(1, std::min(src->bvh.max_stack, RT_LDS_STACK));
    ctx->cfg.stack_entries = entries;
    ctx->cfg.packets = nn > 0 && src->bvh.max_stack <= 64;  // the packet walk keeps its stack in one VGPR (64 lanes)
    if (const char *e = getenv("RT_PACKET_LEVELS")) {  // tuning knob: 0 = primary rays walk per lane like every other ray
        if (atoi(e) <= 0) ctx->cfg.packets = false;
    }
    ctx->dev.any_ordered = 0;  // tuning knob: shadow rays descend near-first (costs ~1 % on the competition scene)
    if (const char *e = getenv("RT_ANY_ORDERED")) ctx->dev.any_ordered = atoi(e) != 0;
    ctx->dev.occluder_cache = 1;  // RT_OCCLUDER_CACHE=0: shadow queries independent of scheduling down to the last float bit
    if (const char *e = getenv("RT_OCCLUDER_CACHE")) ctx->dev.occluder_cache = atoi(e) != 0;
    ctx->cfg.analytic_bytes = (int)(img.units.size() * sizeof(RtSphere) + src->planes.size() * sizeof(RtPlane));
    const int lds_per_block = (entries + RT_TRACE_LDS_ROWS) * 256 * 4 + ctx->cfg.analytic_bytes;  // stack rows + the age row + the aux row + analytic primitives
    if (const char *e = getenv("RT_SPLIT_BUDGET")) {  // 0 = never split a shadow walk; rounded down to a power of two
        int b = std::max(0, atoi(e)), p2 = 0;
        while (b > 1) { b >>= 1; p2++; }
        ctx->cfg.split_budget = atoi(e) > 0 ? 1 << p2 : 0;
    }
    ctx->cfg.trace_blocks_per_cu = std::max(1, std::min(RT_TRACE_MIN_BLOCKS, (160 * 1024) / lds_per_block));
    if (const char *e = getenv("RT_TRACE_BLOCKS")) ctx->cfg.trace_blocks_per_cu = std::max(1, std::min(ctx->cfg.trace_blocks_per_cu, atoi(e)));  // (tuning knob)
    const size_t spill_entries = (size_t)std::max(0, src->bvh.max_stack - entries);
    ctx->spill_words = std::max<size_t>(1, spill_entries * (size_t)ctx->cfg.num_cus * 8 * 256);
    if (ctx->d_spill.alloc(2 * ctx->spill_words)) return 1;
    HIPOK(hipDeviceSynchronize());
    ctx->commit_stats.ms_upload = nowMs() - t0;
    ctx->committed = true;
    ctx->obs_valid = false;  // a new scene: its fan-out is unknown
    return 0;
}
}  // namespace

int rt_commit(rt_context *ctx) {
    if (!ctx) return fail("rt_commit: NULL context");
    ctx->committed = false;
    CommitImage img;
    if (commitHost(ctx, img)) return 1;
    return commitDevice(ctx, ctx, img);
}

int rt_get_commit_stats(rt_context *ctx, rt_commit_stats *out) {
    if (!ctx || !out) return fail("rt_get_commit_stats: NULL argument");
    if (!ctx->committed) return fail("rt_get_commit_stats: no committed scene");
    *out = ctx->commit_stats;
    return 0;
}

int rt_tile_counts(int32_t width, int32_t height, int32_t rank, int32_t world, int64_t *n_local, int64_t *n_total) {
    if (width <= 0 || height <= 0 || world <= 0 || rank < 0 || rank >= world) return fail("rt_tile_counts: bad arguments");
    const int64_t tiles = (int64_t)((width + RT_TILE - 1) / RT_TILE) * ((height + RT_TILE - 1) / RT_TILE);
    if (n_total) *n_total = tiles;
    if (n_local) *n_local = tiles > rank ? (tiles - rank + world - 1) / world : 0;
    return 0;
}

int rt_render_tiles_device(rt_context *ctx, const rt_camera *cam, const rt_render_params *p, void *d_rgb8, void *d_rgbf,
                           void *d_primid, void *stream, rt_stats *stats) {
    if (!ctx || !cam || !p) return fail("rt_render_tiles_device: NULL argument");
    return renderTiles(ctx, cam, p, (uint8_t *)d_rgb8, (float *)d_rgbf, (uint32_t *)d_primid,
                       stream ? (hipStream_t)stream : ctx->stream, stats);
}

int rt_assemble_tiles_device(rt_context *ctx, int32_t width, int32_t height, int32_t world, int64_t tiles_per_rank,
                             const void *t8, const void *tf, const void *tp, void *o8, void *of, void *op, void *stream) {
    if (!ctx || width <= 0 || height <= 0 || world <= 0) return fail("rt_assemble_tiles_device: bad arguments");
    int64_t need = 0;
    rt_tile_counts(width, height, 0, world, &need, nullptr);
    if (tiles_per_rank < need) return fail("rt_assemble_tiles_device: tiles_per_rank smaller than rank 0's tile count");
    if (bind(ctx)) return 1;
    RtLaunchCfg cfg = ctx->cfg;
    cfg.stream = stream ? (hipStream_t)stream : ctx->stream;
    rtLaunchAssemble(cfg, width, height, world, tiles_per_rank, t8, tf, tp, o8, of, op);
    HIPOK(hipStreamSynchronize(cfg.stream));
    HIPOK(hipGetLastError());
    return 0;
}

int rt_render_tiles_static(rt_context *ctx, const rt_camera *cam, const rt_render_params *p, void *d_rgb8, void *d_rgbf,
                           void *d_primid, void *stream) {
    if (!ctx || !cam || !p) return fail("rt_render_tiles_static: NULL argument");
    return renderTilesStatic(ctx, cam, p, (uint8_t *)d_rgb8, (float *)d_rgbf, (uint32_t *)d_primid, stream ? (hipStream_t)stream : ctx->stream);
}

int rt_render(rt_context *ctx, const rt_camera *cam, const rt_render_params *p, uint8_t *out_rgb8, float *out_rgbf,
              uint32_t *out_primid, rt_stats *stats) {
    if (!ctx || !cam || !p) return fail("rt_render: NULL argument");
    if (p->tile_world > 1) return fail("rt_render renders whole images; use rt_render_tiles_device for tile_world > 1");
    if (bind(ctx)) return 1;
    int64_t n_local = 0;
    if (rt_tile_counts(p->width, p->height, 0, 1, &n_local, nullptr)) return 1;
    const size_t tpix = (size_t)n_local * RT_TILE_PIXELS, npix = (size_t)p->width * p->height;
    if (ctx->d_t8.alloc(tpix * 3) || ctx->d_tf.alloc(tpix * 3) || ctx->d_tp.alloc(tpix)) return 1;
    if (ctx->d_o8.alloc(npix * 3) || ctx->d_of.alloc(npix * 3) || ctx->d_op.alloc(npix)) return 1;
    rt_render_params q = *p;
    q.tile_rank = 0;
    q.tile_world = 1;
    if (renderTiles(ctx, cam, &q, ctx->d_t8.p, ctx->d_tf.p, ctx->d_tp.p, ctx->stream, stats)) return 1;
    if (rt_assemble_tiles_device(ctx, p->width, p->height, 1, n_local, ctx->d_t8.p, ctx->d_tf.p, ctx->d_tp.p, ctx->d_o8.p,
                                 ctx->d_of.p, ctx->d_op.p, ctx->stream))
        return 1;
    if (out_rgb8) HIPOK(hipMemcpy(out_rgb8, ctx->d_o8.p, npix * 3, hipMemcpyDeviceToHost));
    if (out_rgbf) HIPOK(hipMemcpy(out_rgbf, ctx->d_of.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (out_primid) HIPOK(hipMemcpy(out_primid, ctx->d_op.p, npix * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}

int rt_trace_batch(rt_context *ctx, int64_t n, const float *rays, int32_t depth, float *out_color, rt_hit *out_hits,
                   uint8_t *out_shadow, rt_stats *stats) {
    if (!ctx || n < 0 || (n && !rays) || depth < 0 || depth > 15) return fail("rt_trace_batch: bad arguments");
    if (!ctx->committed) return fail("rt_commit has not been called");
    if (bind(ctx)) return 1;
    rt_stats local{};
    rt_stats *st = stats ? stats : &local;
    std::memset(st, 0, sizeof *st);
    int rc = traceBatch(ctx, n, rays, depth, out_color, out_hits, out_shadow, st, stats != nullptr);
    st->primary_samples = (uint64_t)n;
    if (!rc && st->queue_overflow) rc = fail("wavefront queue overflow");
    return rc;
}

int rt_eval_texture(rt_context *ctx, int32_t id, int64_t n, const float *uv, float *out) {
    if (!ctx || n < 0 || id < 0 || id >= RT_TEX_COUNT || (n && (!uv || !out))) return fail("rt_eval_texture: bad arguments");
    if (!ctx->committed && id >= RT_TEX_BMP_COLOR) return fail("rt_eval_texture: BMP maps need a committed scene");
    if (bind(ctx)) return 1;
    if (n == 0) return 0;
    DevBuf<float> d_uv, d_out;
    if (d_uv.alloc((size_t)n * 2) || d_out.alloc((size_t)n * 3)) return 1;
    int rc = 0;
    hipError_t e = hipMemcpy(d_uv.p, uv, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice);
    RtLaunchCfg cfg = ctx->cfg;
    cfg.stream = ctx->stream;
    if (e == hipSuccess) {
        rtLaunchEvalTexture(cfg, ctx->dev, id, n, d_uv.p, d_out.p);
        e = hipStreamSynchronize(cfg.stream);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, d_out.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(hipGetErrorString(e));
    d_uv.release();
    d_out.release();
    return rc;
}

int rt_eval_perlin(rt_context *ctx, int64_t n, const float *p, float *out) {
    if (!ctx || n < 0 || (n && (!p || !out))) return fail("rt_eval_perlin: bad arguments");
    if (bind(ctx)) return 1;
    if (n == 0) return 0;
    DevBuf<float> d_p, d_out;
    if (d_p.alloc((size_t)n * 3) || d_out.alloc((size_t)n)) return 1;
    int rc = 0;
    hipError_t e = hipMemcpy(d_p.p, p, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice);
    RtLaunchCfg cfg = ctx->cfg;
    cfg.stream = ctx->stream;
    if (e == hipSuccess) {
        rtLaunchEvalPerlin(cfg, n, d_p.p, d_out.p);
        e = hipStreamSynchronize(cfg.stream);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, d_out.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(hipGetErrorString(e));
    d_p.release();
    d_out.release();
    return rc;
}

int rt_bvh_check(int64_t n, const float *v0, const float *v1, const float *v2, int32_t threads, int64_t *n_nodes,
                 uint64_t *hash, int64_t *violations) {
    if (n < 0 || (n && (!v0 || !v1 || !v2))) return fail("rt_bvh_check: bad arguments");
    if (threads > 0) setenv("RT_BUILD_THREADS", std::to_string(threads).c_str(), 1);
    RtBvh bvh;
    rtBuildBvh(n, v0, v1, v2, bvh);
    if (threads > 0) unsetenv("RT_BUILD_THREADS");
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t bytes) {
        const uint8_t *b = (const uint8_t *)p;
        for (size_t k = 0; k < bytes; k++) h = (h ^ b[k]) * 1099511628211ull;
    };
    mix(bvh.nodes.data(), bvh.nodes.size() * sizeof(RtNode4));
    mix(bvh.nodes_p.data(), bvh.nodes_p.size() * sizeof(RtNodeP));
    mix(bvh.order.data(), bvh.order.size() * sizeof(uint32_t));
    // containment: walking down from the root, every slot's piece of its triangle must lie inside the decoded (quantised)
    // box and the float box of every child slot above it.  A triangle may own several slots (large triangles are cut into
    // pieces before the build, rt_bvh.cpp): what must hold is that the pieces' boxes together cover it -- checked on 16
    // points of every triangle (corners, edge midpoints, centroid, interior points): each must lie inside the box chain of
    // at least one of the triangle's slots -- and that every slot is reached exactly once.
    int64_t bad = 0;
    static const double kBary[16][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {.5, .5, 0}, {0, .5, .5}, {.5, 0, .5}, {1 / 3., 1 / 3., 1 / 3.},
                                         {.6, .2, .2}, {.2, .6, .2}, {.2, .2, .6}, {.8, .1, .1}, {.1, .8, .1}, {.1, .1, .8},
                                         {.45, .45, .1}, {.1, .45, .45}, {.45, .1, .45}};
    std::vector<uint16_t> covered((size_t)n, 0);
    std::vector<uint8_t> seen(bvh.order.size(), 0);
    if ((int64_t)bvh.order.size() < n || bvh.references != (int64_t)bvh.order.size()) bad++;
    const uint32_t nn = (uint32_t)bvh.nodes.size();
    struct Item { uint32_t ref; float lo[3], hi[3]; };
    std::vector<Item> stack;
    Item root{bvh.root_ref, {-INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY}};
    if (n) stack.push_back(root);
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        if (it.ref & 0x80000000u) {
            const uint32_t first = (it.ref & 0x0FFFFFFFu) - nn, cnt = ((it.ref >> 28) & 7u) + 1u;
            for (uint32_t s = first; s < first + cnt; s++) {
                if (s >= (uint32_t)bvh.order.size()) { bad++; continue; }
                const uint32_t t = bvh.order[s];
                if (t >= (uint32_t)n) { bad++; continue; }
                seen[s]++;
                const float *vs[3] = {v0 + 3 * (size_t)t, v1 + 3 * (size_t)t, v2 + 3 * (size_t)t};
                for (int k = 0; k < 16; k++) {
                    bool in = true;
                    for (int ax = 0; ax < 3; ax++) {
                        const double x = kBary[k][0] * vs[0][ax] + kBary[k][1] * vs[1][ax] + kBary[k][2] * vs[2][ax];
                        // (the point is a rounded combination of the corners: a box face it sits on may miss it by an ulp)
                        const double tol = 4e-7 * std::fmax(1e-3, std::fmax(std::fabs(vs[0][ax]), std::fmax(std::fabs(vs[1][ax]), std::fabs(vs[2][ax]))));
                        if (x < it.lo[ax] - tol || x > it.hi[ax] + tol) in = false;
                    }
                    if (in) covered[t] |= (uint16_t)(1u << k);
                }
            }
            continue;
        }
        const RtNode4 &nd = bvh.nodes[it.ref];
        const RtNodeP &np = bvh.nodes_p[it.ref];
        const double sc[3] = {nd.scale_x, nd.scale_y, nd.scale_z};
        const uint32_t qhi[3] = {nd.qhi_x, nd.qhi_y, nd.qhi_z};
        for (int k = 0; k < 4; k++) {
            if (nd.child[k] == 0xFFFFFFFFu) continue;
            if (np.child[k] != nd.child[k]) bad++;
            Item c;
            c.ref = nd.child[k];
            for (int ax = 0; ax < 3; ax++) {  // intersection of the ancestors' boxes with this slot's two boxes
                const double qlo = nd.origin[ax] + ((nd.qlo[ax] >> (8 * k)) & 255u) * sc[ax];
                const double qh = nd.origin[ax] + ((qhi[ax] >> (8 * k)) & 255u) * sc[ax];
                if (qlo > np.lo[ax][k] || qh < np.hi[ax][k]) bad++;  // the quantised box must contain the float box
                c.lo[ax] = std::fmax(it.lo[ax], np.lo[ax][k]);
                c.hi[ax] = std::fmin(it.hi[ax], np.hi[ax][k]);
            }
            stack.push_back(c);
        }
    }
    for (int64_t t = 0; t < n; t++)
        if (covered[t] != 0xFFFFu) bad++;
    for (uint8_t c : seen)
        if (c != 1) bad++;
    if (n_nodes) *n_nodes = (int64_t)bvh.nodes.size();
    if (hash) *hash = h;
    if (violations) *violations = bad;
    return 0;
}

int rt_get_bvh(rt_context *ctx, int64_t *n_nodes, float *nodes, int64_t *n_triangles, uint32_t *tri_order) {
    if (!ctx || !ctx->committed) return fail("rt_get_bvh: no committed scene");
    if (n_nodes) *n_nodes = (int64_t)ctx->bvh.nodes.size();
    if (n_triangles) *n_triangles = (int64_t)ctx->bvh.order.size();
    if (nodes && !ctx->bvh.nodes.empty()) std::memcpy(nodes, ctx->bvh.nodes.data(), ctx->bvh.nodes.size() * sizeof(RtNode4));
    if (tri_order && !ctx->bvh.order.empty()) std::memcpy(tri_order, ctx->bvh.order.data(), ctx->bvh.order.size() * sizeof(uint32_t));
    return 0;
}

int rt_get_bvh_info(rt_context *ctx, uint32_t *root_ref, float *bound, int32_t *max_stack, int32_t *lds_stack_entries) {
    if (!ctx || !ctx->committed) return fail("rt_get_bvh_info: no committed scene");
    if (root_ref) *root_ref = ctx->bvh.root_ref;
    if (bound) *bound = ctx->dev.bound;
    if (max_stack) *max_stack = ctx->bvh.max_stack;
    if (lds_stack_entries) *lds_stack_entries = ctx->cfg.stack_entries;
    return 0;
}

// ---- rt_multi: one frame over N GPUs of this process ------------------------------------------------------------------
// Stands where `#pragma omp parallel for schedule(dynamic, 1)` over the 16x16 tiles stood (main.cpp:607): the tiles are
// interleaved over the GPUs (tile t -> rank t % N), scene + BVH are replicated (built once on the host, uploaded N times),
// every GPU renders its tiles from a host thread of its own, and ONE exchange step assembles the frame on rank 0:
// RCCL point-to-point over xGMI -- ncclGroupStart; ncclRecv x (N-1) on rank 0 / ncclSend on the peers; ncclGroupEnd --
// straight into the slots of a preallocated [rank][tile][256] buffer (rank 0 renders into its own slot), then
// k_assemble de-interleaves.  No collective during rendering.  RCCL (573 MB of code objects) is loaded with dlopen when
// the first rt_multi is created, so single-GPU users of the library never map it.
namespace {
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) commInitAll = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclGroupStart) groupStart = nullptr;
    decltype(&ncclGroupEnd) groupEnd = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGetErrorString) errorString = nullptr;
};
RcclApi g_rccl;
int loadRccl() {
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    if (g_rccl.lib) return 0;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(std::string("rt_multi: cannot load RCCL: ") + dlerror());
    RcclApi a;
    a.lib = lib;
#define RT_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(lib, name)); if (!a.field) return fail(std::string("rt_multi: RCCL lacks ") + name)
    RT_SYM(commInitAll, "ncclCommInitAll");
    RT_SYM(commDestroy, "ncclCommDestroy");
    RT_SYM(groupStart, "ncclGroupStart");
    RT_SYM(groupEnd, "ncclGroupEnd");
    RT_SYM(send, "ncclSend");
    RT_SYM(recv, "ncclRecv");
    RT_SYM(errorString, "ncclGetErrorString");
#undef RT_SYM
    g_rccl = a;
    return 0;
}
#define NCCLOK(expr)                                                                                   \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) return fail(std::string(#expr) + ": " + g_rccl.errorString(r_));       \
    } while (0)
}  // namespace

// One host thread per GPU, kept for the life of the rt_multi: a frame hands every worker its rank's share and waits.
struct RtWorkers {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable wake, done;
    std::function<void(int)> job;
    uint64_t generation = 0;
    int pending = 0;
    bool quit = false;
    void start(int n) {
        for (int k = 0; k < n; k++)
            threads.emplace_back([this, k] {
                uint64_t seen = 0;
                for (;;) {
                    std::function<void(int)> f;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        wake.wait(lk, [&] { return quit || generation != seen; });
                        if (quit) return;
                        seen = generation;
                        f = job;
                    }
                    f(k);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) done.notify_all();
                    }
                }
            });
    }
    void run(const std::function<void(int)> &f) {  // f(rank) on every worker; returns when all have finished
        std::unique_lock<std::mutex> lk(mu);
        job = f;
        pending = (int)threads.size();
        generation++;
        wake.notify_all();
        done.wait(lk, [&] { return pending == 0; });
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        wake.notify_all();
        for (std::thread &t : threads) t.join();
        threads.clear();
    }
};

struct rt_multi {
    int n = 0;
    RtWorkers workers;
    rt_multi_timings last{};  // of the last rt_multi_render
    std::vector<int> dev;
    std::vector<rt_context *> ctx;
    bool rccl = false;       // transport of the frame-end exchange: RCCL send/recv, else device-to-device copies
    bool self_rccl = false;  // n == 1: push rank 0's own tiles through ncclSend/ncclRecv to itself (rehearsal on a 1-GPU box)
    std::vector<ncclComm_t> comms;
    int64_t per = 0;         // tiles per rank the buffers below are sized for
    // rank-local tile-major buffers of ranks >= 1 (and of rank 0 under self_rccl), each on its rank's GPU
    std::vector<DevBuf<uint8_t>> t8;
    std::vector<DevBuf<float>> tf;
    std::vector<DevBuf<uint32_t>> tp;
    // on rank 0's GPU: the gathered buffers [rank][per][256][..] and the assembled row-major images
    DevBuf<uint8_t> g8, o8;
    DevBuf<float> gf, of;
    DevBuf<uint32_t> gp, op;
};

int rt_multi_create(const int32_t *device_ids, int32_t n, uint32_t flags, rt_multi **out) {
    if (!out) return fail("rt_multi_create: out is NULL");
    *out = nullptr;
    if (!device_ids || n <= 0 || n > 64) return fail("rt_multi_create: bad arguments (1..64 devices)");
    rt_multi *m = new rt_multi();
    m->n = n;
    m->dev.assign(device_ids, device_ids + n);
    m->t8.resize(n); m->tf.resize(n); m->tp.resize(n);
    bool distinct = true;
    for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++)
            if (device_ids[a] == device_ids[b]) distinct = false;
    for (int k = 0; k < n; k++) {
        rt_context *c = nullptr;
        if (rt_create(device_ids[k], &c)) {
            for (rt_context *x : m->ctx) rt_destroy(x);
            delete m;
            return 1;
        }
        m->ctx.push_back(c);
    }
    const char *tr = getenv("RT_MULTI_TRANSPORT");  // "copy": device-to-device copies instead of RCCL
    const bool want_copy = tr && std::string(tr) == "copy";
    m->self_rccl = n == 1 && (flags & RT_MULTI_REHEARSE_RCCL) != 0;
    // RCCL needs one GPU per rank: ranks that share a GPU (a rehearsal of N ranks on one card) exchange by copies
    m->rccl = !want_copy && distinct && (n > 1 || m->self_rccl);
    if (m->rccl) {
        if (loadRccl()) { rt_multi_destroy(m); return 1; }
        m->comms.assign(n, nullptr);
        ncclResult_t r = g_rccl.commInitAll(m->comms.data(), n, m->dev.data());
        if (r != ncclSuccess) {
            m->comms.clear();
            rt_multi_destroy(m);
            return fail(std::string("rt_multi_create: ncclCommInitAll: ") + g_rccl.errorString(r));
        }
    }
    m->workers.start(n);
    *out = m;
    return 0;
}

int rt_multi_destroy(rt_multi *m) {
    if (!m) return 0;
    m->workers.stop();
    for (int k = 0; k < m->n && k < (int)m->ctx.size(); k++) {
        (void)hipSetDevice(m->dev[k]);
        (void)hipDeviceSynchronize();
        m->t8[k].release(); m->tf[k].release(); m->tp[k].release();
    }
    for (ncclComm_t c : m->comms)
        if (c) (void)g_rccl.commDestroy(c);
    if (!m->ctx.empty()) {
        (void)hipSetDevice(m->dev[0]);
        m->g8.release(); m->gf.release(); m->gp.release(); m->o8.release(); m->of.release(); m->op.release();
    }
    for (rt_context *c : m->ctx) rt_destroy(c);
    delete m;
    return 0;
}

int32_t rt_multi_size(const rt_multi *m) { return m ? m->n : 0; }
const char *rt_multi_transport(const rt_multi *m) { return !m ? "" : m->rccl ? "rccl" : (m->n > 1 ? "copy" : "none"); }

rt_context *rt_multi_context(rt_multi *m, int32_t rank) {
    if (!m || rank < 0 || rank >= m->n) return nullptr;
    return m->ctx[rank];
}

int rt_multi_commit(rt_multi *m) {
    if (!m) return fail("rt_multi_commit: NULL argument");
    rt_context *src = m->ctx[0];
    for (rt_context *c : m->ctx) c->committed = false;
    CommitImage img;
    if (commitHost(src, img)) return 1;
    // uploads run concurrently, one host thread per GPU
    std::vector<int> rc(m->n, 0);
    std::vector<std::string> err(m->n);
    m->workers.run([&](int k) {
        rc[k] = commitDevice(m->ctx[k], src, img);
        if (rc[k]) err[k] = g_err;
    });
    for (int k = 0; k < m->n; k++)
        if (rc[k]) return fail("rt_multi_commit: rank " + std::to_string(k) + ": " + err[k]);
    for (int k = 1; k < m->n; k++) {  // the replicas report the build's split with their own upload time
        const double up = m->ctx[k]->commit_stats.ms_upload;
        m->ctx[k]->commit_stats = src->commit_stats;
        m->ctx[k]->commit_stats.ms_upload = up;
    }
    return 0;
}

int rt_multi_render(rt_multi *m, const rt_camera *cam, const rt_render_params *p, uint8_t *out_rgb8, float *out_rgbf,
                    uint32_t *out_primid, rt_stats *stats) {
    if (!m || !cam || !p) return fail("rt_multi_render: NULL argument");
    if (p->width <= 0 || p->height <= 0) return fail("rt_multi_render: bad image size");
    const int n = m->n;
    int64_t per = 0;
    rt_tile_counts(p->width, p->height, 0, n, &per, nullptr);  // rank 0 owns the most tiles: every slot has that size
    const size_t slot = (size_t)per * RT_TILE_PIXELS, npix = (size_t)p->width * p->height;
    const bool want_f = out_rgbf != nullptr;
    // buffers (kept across frames): gathered + assembled on rank 0's GPU, tile buffers on the peers' GPUs
    HIPOK(hipSetDevice(m->dev[0]));
    if (m->g8.alloc(slot * 3 * n) || m->gp.alloc(slot * n) || m->o8.alloc(npix * 3) || m->op.alloc(npix)) return 1;
    if (want_f && (m->gf.alloc(slot * 3 * n) || m->of.alloc(npix * 3))) return 1;
    for (int k = (m->self_rccl ? 0 : 1); k < n; k++) {
        HIPOK(hipSetDevice(m->dev[k]));
        if (m->t8[k].alloc(slot * 3) || m->tp[k].alloc(slot) || (want_f && m->tf[k].alloc(slot * 3))) return 1;
    }
    m->per = per;
    // ---- render: one host thread per GPU, each rank its interleaved tiles ----
    std::vector<int> rc(n, 0);
    std::vector<std::string> err(n);
    std::vector<rt_stats> st(n);
    auto local8 = [&](int k) { return (k == 0 && !m->self_rccl) ? m->g8.p : m->t8[k].p; };  // rank 0 renders into its slot
    auto localf = [&](int k) { return !want_f ? nullptr : (k == 0 && !m->self_rccl) ? m->gf.p : m->tf[k].p; };
    auto localp = [&](int k) { return (k == 0 && !m->self_rccl) ? m->gp.p : m->tp[k].p; };
    const double t_render0 = nowMs();
    m->workers.run([&](int k) {
        rt_render_params q = *p;
        q.tile_rank = k;
        q.tile_world = n;
        rc[k] = renderTiles(m->ctx[k], cam, &q, local8(k), localf(k), localp(k), m->ctx[k]->stream, &st[k]);
        if (rc[k]) err[k] = g_err;
    });
    const double t_render1 = nowMs();
    for (int k = 0; k < n; k++)
        if (rc[k]) return fail("rt_multi_render: rank " + std::to_string(k) + ": " + err[k]);
    // ---- the one exchange step ----
    hipStream_t s0 = m->ctx[0]->stream;
    const int first = m->self_rccl ? 0 : 1;
    if (m->rccl) {
        // Every call between ncclGroupStart and ncclGroupEnd is only queued; an early return inside the group would leave
        // this thread's group open, and every later RCCL call of the thread (torch's included) would queue into it and
        // never launch.  So: remember the first error, always close the group, then report.
        ncclResult_t bad = ncclSuccess;
        const char *where = "";
        auto note = [&](ncclResult_t r, const char *what) {
            if (r != ncclSuccess && bad == ncclSuccess) {
                bad = r;
                where = what;
            }
        };
        note(g_rccl.groupStart(), "ncclGroupStart");
        if (bad == ncclSuccess) {
            for (int k = first; k < n && bad == ncclSuccess; k++) {
                note(g_rccl.recv(m->g8.p + (size_t)k * slot * 3, slot * 3, ncclUint8, k, m->comms[0], s0), "ncclRecv(rgb8)");
                note(g_rccl.recv(m->gp.p + (size_t)k * slot, slot, ncclUint32, k, m->comms[0], s0), "ncclRecv(primid)");
                if (want_f) note(g_rccl.recv(m->gf.p + (size_t)k * slot * 3, slot * 3, ncclFloat32, k, m->comms[0], s0), "ncclRecv(rgbf)");
                hipStream_t sk = m->ctx[k]->stream;
                note(g_rccl.send(m->t8[k].p, slot * 3, ncclUint8, 0, m->comms[k], sk), "ncclSend(rgb8)");
                note(g_rccl.send(m->tp[k].p, slot, ncclUint32, 0, m->comms[k], sk), "ncclSend(primid)");
                if (want_f) note(g_rccl.send(m->tf[k].p, slot * 3, ncclFloat32, 0, m->comms[k], sk), "ncclSend(rgbf)");
            }
            note(g_rccl.groupEnd(), "ncclGroupEnd");
        }
        if (bad != ncclSuccess) return fail(std::string("rt_multi_render: ") + where + ": " + g_rccl.errorString(bad));
        for (int k = first; k < n; k++) {  // the peers' sends have left their buffers (next frame may overwrite them)
            if (k == 0) continue;
            HIPOK(hipSetDevice(m->dev[k]));
            HIPOK(hipStreamSynchronize(m->ctx[k]->stream));
        }
    } else {
        for (int k = first; k < n; k++) {  // ranks that share a GPU (rehearsal) or RT_MULTI_TRANSPORT=copy
            HIPOK(hipSetDevice(m->dev[0]));
            HIPOK(hipMemcpyPeerAsync(m->g8.p + (size_t)k * slot * 3, m->dev[0], m->t8[k].p, m->dev[k], slot * 3, s0));
            HIPOK(hipMemcpyPeerAsync(m->gp.p + (size_t)k * slot, m->dev[0], m->tp[k].p, m->dev[k], slot * sizeof(uint32_t), s0));
            if (want_f) HIPOK(hipMemcpyPeerAsync(m->gf.p + (size_t)k * slot * 3, m->dev[0], m->tf[k].p, m->dev[k], slot * 3 * sizeof(float), s0));
        }
    }
    // ---- rank 0: de-interleave (stream-ordered behind the receives), hand the image to the host ----
    HIPOK(hipSetDevice(m->dev[0]));
    HIPOK(hipStreamSynchronize(s0));  // (the exchange alone, for rt_multi_timings; the assembly synchronises anyway)
    const double t_exchange1 = nowMs();
    if (rt_assemble_tiles_device(m->ctx[0], p->width, p->height, n, per, m->g8.p, want_f ? m->gf.p : nullptr, m->gp.p, m->o8.p,
                                 want_f ? m->of.p : nullptr, m->op.p, s0))
        return 1;
    const double t_assemble1 = nowMs();
    if (out_rgb8) HIPOK(hipMemcpy(out_rgb8, m->o8.p, npix * 3, hipMemcpyDeviceToHost));
    if (out_rgbf) HIPOK(hipMemcpy(out_rgbf, m->of.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (out_primid) HIPOK(hipMemcpy(out_primid, m->op.p, npix * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (stats)
        for (int k = 0; k < n; k++) stats[k] = st[k];
    m->last.ms_render_wall = t_render1 - t_render0;
    m->last.ms_exchange = t_exchange1 - t_render1;
    m->last.ms_assemble = t_assemble1 - t_exchange1;
    m->last.ms_readback = nowMs() - t_assemble1;
    m->last.ms_share_max = m->last.ms_share_min = st[0].ms_total;
    for (int k = 1; k < n; k++) {
        m->last.ms_share_max = std::max(m->last.ms_share_max, st[k].ms_total);
        m->last.ms_share_min = std::min(m->last.ms_share_min, st[k].ms_total);
    }
    return 0;
}

int rt_multi_last_timings(const rt_multi *m, rt_multi_timings *out) {
    if (!m || !out) return fail("rt_multi_last_timings: NULL argument");
    *out = m->last;
    return 0;
}

}  // extern "C"
