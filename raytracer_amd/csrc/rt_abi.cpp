// rt_abi.cpp -- implementation of the C ABI declared in include/rt_hip.h: context, scene upload, BVH commit, the
// chunked wavefront render loop and the test hooks.  Host code only; kernels live in rt_kernels.hip.
// There is no CPU rendering path in this library: every entry point that produces pixels, hits or texture
// values launches HIP kernels and fails if no device is usable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
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
    int upload(const std::vector<T> &h) {
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

}  // namespace

struct rt_context {
    int device = 0;
    hipStream_t stream = nullptr;
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
    DevBuf<uint32_t> d_spill;   // traversal stack entries beyond the LDS part
    DevBuf<RtNodeP> d_nodes_p;  // float-box form of the nodes (packet kernels)
    DevBuf<uint32_t> d_resid;   // queue indices of the rays the packet kernels hand to the per-lane kernels
    DevBuf<uint32_t> d_general; // queue indices of the hits the fast k_shade leaves to the general one
    DevBuf<uint2> d_split[2];   // work items of split any-hit walks (rt_kernels.hip), ping-pong by round
    DevBuf<RtTriShade> d_shade;
    DevBuf<RtSphere> d_spheres;
    DevBuf<RtPlane> d_planes;
    DevBuf<rt_material> d_materials;
    DevBuf<int32_t> d_objmat;
    DevBuf<RtLight> d_lights;
    DevBuf<uint8_t> d_bmp[4];
    RtSceneDev dev{};
    // wavefront workspace (sized for one chunk, reused across calls)
    size_t ws_samples = 0, ws_bytes = 0;
    // Fan-out observed on this scene: queue entries per primary sample at the fullest level, maximum over the chunks
    // rendered since rt_commit (rays; shadow rays over all lights), and the deepest bounce budget it was seen at.
    // Chunks are sized from it once it exists (renderTiles); reset by rt_commit.
    bool obs_valid = false;
    double obs_ray = 0.0, obs_sh = 0.0;
    double obs_l0_share = 0.0;  // share of the last chunk's shadow rays that were level-0 rays
    int obs_depth = -1;
    DevBuf<float4> d_rayA[2], d_rayB[2], d_shA, d_shC;
    DevBuf<uint8_t> d_shL, d_occ0;  // d_occ0 / d_col0: level-0 shadow results stored per [light][sample] (RtQueues::occ0)
    DevBuf<float4> d_col0;
    DevBuf<float> d_rayR[2];
    DevBuf<float2> d_hits;
    DevBuf<long long> d_accum;
    DevBuf<uint32_t> d_counters;
    RtQueues q{};
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

// queue entries one chunk of `samples` primary samples can need: worst case of the deepest level, plus what k_shade's
// wave-private slabs may leave unused (one partly filled slab per wave and queue, one slot per slab where a
// reflection/refraction pair met its end)
struct QueueSizes {
    size_t ray_cap, sh_cap, bytes;
    bool fits32;
};
// ray_fan / sh_fan: queue entries per primary sample at the fullest level (rays; shadow rays over all lights)
QueueSizes queueSizes(const rt_context *ctx, size_t samples, double ray_fan, double sh_fan, int nl) {
    QueueSizes z;
    const size_t waves = (size_t)rtShadeWaves(ctx->cfg);
    const size_t rays = (size_t)std::ceil((double)samples * ray_fan), shadows = (size_t)std::ceil((double)samples * sh_fan);
    // (x 2: the fast and the general instance of k_shade both reserve slabs)
    z.ray_cap = rays + 2 * waves * RT_SLAB_MAX + rays / 64 + 128;
    z.sh_cap = shadows + 2 * waves * nl * RT_SLAB_MAX;
    z.fits32 = z.ray_cap < (1ull << 32) && z.sh_cap < (1ull << 32);
    z.bytes = z.ray_cap * (2 * (16 + 16 + 4) + 8 + 4) + z.sh_cap * 33 + samples * 24 + std::max(z.sh_cap, z.ray_cap) * 4;
    return z;
}
QueueSizes queueSizes(const rt_context *ctx, size_t samples, int fanout, int nl) {  // worst case
    return queueSizes(ctx, samples, (double)fanout, (double)fanout * nl, nl);
}

// rays of the deepest level per primary sample, worst case: every hit spawns a reflection/refraction pair (main.cpp:148-205)
int worstFanout(int max_depth) { return 1 << std::min(std::max(max_depth, 0), 15); }

double nowMs();
int ensureWorkspace(rt_context *ctx, size_t samples, const QueueSizes &z);
int ensureWorkspace(rt_context *ctx, size_t samples, int max_depth, int n_lights) {
    return ensureWorkspace(ctx, samples, queueSizes(ctx, samples, worstFanout(max_depth), std::max(n_lights, 1)));
}
int ensureWorkspace(rt_context *ctx, size_t samples, const QueueSizes &z) {
    if (!z.fits32) return fail("chunk too large: queue capacity exceeds 2^32 entries");
    if (samples <= ctx->ws_samples && z.ray_cap <= ctx->q.ray_cap && z.sh_cap <= ctx->q.shadow_cap) return 0;
    // grow: capacities only ever go up (a later call with fewer samples but more bounces keeps the larger of each)
    samples = std::max(samples, ctx->ws_samples);
    // a queue that must grow grows by an eighth more than asked (capped below 2^32): the fan-out a scene shows moves by a
    // few per cent from frame to frame, and every growth is a hipFree + hipMalloc of tens of GB
    auto grown = [](size_t want, size_t have) -> size_t {
        if (want <= have) return have;
        return have ? std::min<size_t>(want + want / 8, 0xFFFFFFFEull) : want;
    };
    const size_t ray_cap = grown(z.ray_cap, ctx->q.ray_cap), sh_cap = grown(z.sh_cap, ctx->q.shadow_cap);
    // DevBuf::alloc frees before it allocates: after a failed grow some buffers are new, some are gone.  The
    // workspace is then declared empty (sizes 0, queue pointers null), so that no later call takes the early return
    // above and launches kernels on freed memory; the next call allocates from scratch.
    auto grow = [&]() -> int {
        for (int k = 0; k < 2; k++) {
            if (ctx->d_rayA[k].alloc(ray_cap) || ctx->d_rayB[k].alloc(ray_cap) || ctx->d_rayR[k].alloc(ray_cap)) return 1;
        }
        if (ctx->d_hits.alloc(ray_cap + 1)) return 1;
        if (ctx->d_shA.alloc(sh_cap) || ctx->d_shC.alloc(sh_cap) || ctx->d_shL.alloc(sh_cap)) return 1;
        if (ctx->d_accum.alloc(samples * 3)) return 1;
        if (ctx->d_counters.alloc(RT_CNT_WORDS)) return 1;
        if (ctx->d_resid.alloc(std::max(sh_cap, ray_cap))) return 1;
        if (ctx->d_general.alloc(ray_cap)) return 1;
        return 0;
    };
    const double t_grow0 = nowMs();
    size_t split_cap = (size_t)1 << 22;  // items per round; a wave that finds no room keeps walking
    if (const char *e = getenv("RT_SPLIT_CAP")) split_cap = (size_t)std::max(64, atoi(e));  // (test knob: forces the no-room path)
    if (grow() || ctx->d_split[0].alloc(split_cap) || ctx->d_split[1].alloc(split_cap)) {
        const std::string why = g_err;
        for (int k = 0; k < 2; k++) { ctx->d_rayA[k].release(); ctx->d_rayB[k].release(); ctx->d_rayR[k].release(); ctx->d_split[k].release(); }
        ctx->d_hits.release(); ctx->d_shA.release(); ctx->d_shC.release(); ctx->d_shL.release(); ctx->d_occ0.release(); ctx->d_col0.release(); ctx->d_accum.release(); ctx->d_resid.release(); ctx->d_general.release();
        ctx->ws_samples = 0;
        ctx->ws_bytes = 0;
        ctx->q = RtQueues{};
        ctx->q.spill = ctx->d_spill.p;
        return fail("workspace allocation failed (workspace released): " + why);
    }
    ctx->ws_bytes = ray_cap * (2 * (16 + 16 + 4) + 8 + 4) + sh_cap * 33 + samples * 24 + std::max(sh_cap, ray_cap) * 4 + 2 * split_cap * sizeof(uint2);
    ctx->ws_samples = samples;
    if (getenv("RT_VERBOSE"))
        std::fprintf(stderr, "[rt] workspace grown to %zu samples, %zu ray slots, %zu shadow slots (%.1f GB) in %.0f ms\n", samples, ray_cap,
                     sh_cap, (double)ctx->ws_bytes / 1e9, nowMs() - t_grow0);
    RtQueues &q = ctx->q;
    for (int k = 0; k < 2; k++) {
        q.rayA[k] = ctx->d_rayA[k].p;
        q.rayB[k] = ctx->d_rayB[k].p;
        q.rayR[k] = ctx->d_rayR[k].p;
    }
    q.hits = ctx->d_hits.p;
    q.shA = ctx->d_shA.p;
    q.shC = ctx->d_shC.p;
    q.shL = ctx->d_shL.p;
    q.occ0 = nullptr;  // (set per chunk by renderTiles)
    q.col0 = nullptr;
    q.l0_stride = 0;
    q.accum = ctx->d_accum.p;
    q.counters = ctx->d_counters.p;
    q.spill = ctx->d_spill.p;
    q.resid = ctx->d_resid.p;
    q.general = ctx->d_general.p;
    q.split[0] = ctx->d_split[0].p;
    q.split[1] = ctx->d_split[1].p;
    q.split_cap = (uint32_t)split_cap;
    q.ray_cap = (uint32_t)ray_cap;
    q.shadow_cap = (uint32_t)sh_cap;
    return 0;
}

struct KernelTimer {  // optional per-kernel-class timing with HIP events on the launch stream
    bool on = false;
    hipStream_t stream = nullptr;
    struct Span { hipEvent_t a, b; int cls; };
    std::vector<Span> spans;
    void begin(int cls) {
        if (!on) return;
        Span s;
        (void)hipEventCreate(&s.a);
        (void)hipEventCreate(&s.b);
        s.cls = cls;
        (void)hipEventRecord(s.a, stream);
        spans.push_back(s);
    }
    void end() {
        if (!on) return;
        (void)hipEventRecord(spans.back().b, stream);
    }
    void collect(double *ms /*[6]*/) {
        for (Span &s : spans) {
            float t = 0;
            (void)hipEventElapsedTime(&t, s.a, s.b);
            ms[s.cls] += t;
            (void)hipEventDestroy(s.a);
            (void)hipEventDestroy(s.b);
        }
        spans.clear();
    }
};

void addCounters(rt_stats *st, const uint32_t *c, int max_depth) {
    for (int d = 0; d <= max_depth && d < 16; d++) {
        st->rays_closest += c[RT_CNT_RAYS(d)];
        st->rays_shadow += c[RT_CNT_SHADOW(d)];
    }
    auto u64at = [&](int w) { return ((uint64_t)c[w + 1] << 32) | c[w]; };
    st->rays_closest -= c[RT_CNT_INVALID];  // slots of pixels outside the image and unused slab slots are not rays
    st->rays_shadow -= c[RT_CNT_SHADOW_HOLES];
    st->shadow_culled += c[RT_CNT_CULLED];
    st->shadow_zero += c[RT_CNT_ZERO];
    st->shadow_unoccluded += u64at(RT_CNT_UNOCCLUDED);
    for (int d = 0; d <= max_depth && d < 16; d++)
        for (int r = 0; r <= RT_SPLIT_ROUNDS; r++) st->split_items += c[RT_CNT_SPLIT(d, r)];
    st->hits_shaded += c[RT_CNT_SHADED];
    st->queue_overflow += c[RT_CNT_OVERFLOW];
    st->nodes_visited += u64at(RT_CNT_NODES);
    st->tris_tested += u64at(RT_CNT_NODES + 2);
    st->nodes_visited_any += u64at(RT_CNT_ANY_NODES);
    st->max_walk_closest = std::max<uint64_t>(st->max_walk_closest, c[RT_CNT_NODES + 4]);
    st->max_walk_any = std::max<uint64_t>(st->max_walk_any, c[RT_CNT_ANY_NODES + 4]);
    st->tris_tested_any += u64at(RT_CNT_ANY_NODES + 2);
    for (int k = 0; k < 10; k++) {
        st->prof_closest[k] += u64at(RT_CNT_PROF_CLOSEST + 2 * k);
        st->prof_any[k] += u64at(RT_CNT_PROF_ANY + 2 * k);
    }
    for (int k = 0; k < 6; k++) st->prof_shade[k] += u64at(RT_CNT_PROF_SHADE + 2 * k);
    for (int k = 0; k < 8; k++) {
        st->prof_packet_closest[k] += u64at(RT_CNT_PROF_PACKET_CLOSEST + 2 * k);
        st->prof_packet_any[k] += u64at(RT_CNT_PROF_PACKET_ANY + 2 * k);
    }
}

// levels 0..max_depth of one chunk whose level-0 queue has already been filled
void runLevels(rt_context *ctx, const RtQueues &q, const RtLaunchCfg &cfg, int max_depth, int spp, uint32_t tile0, uint32_t *primid_tiles,
               int stats, KernelTimer &kt, rt_stats *st, uint64_t level0_rays, bool level0_traced = false) {
    for (int level = 0; level <= max_depth; level++) {
        if (!(level == 0 && level0_traced)) {  // each level's queue is claimed once (dynamic fetch counters)
            if (rtLevelUsesPackets(cfg, level, false)) {
                kt.begin(5);
                rtLaunchClosest(cfg, ctx->dev, ctx->bvh.root_ref, q, level, stats, 1);
                kt.end();
            }
            kt.begin(1);
            rtLaunchClosest(cfg, ctx->dev, ctx->bvh.root_ref, q, level, stats, 2);
            kt.end();
        }
        kt.begin(2);
        rtLaunchShade(cfg, ctx->dev, q, level, max_depth, spp, tile0, primid_tiles, stats >= 2, level0_rays << level);
        kt.end();
        kt.begin(3);
        rtLaunchAny(cfg, ctx->dev, ctx->bvh.root_ref, q, level, stats);
        if (level == 0 && q.occ0) rtLaunchFold0(cfg, q, (uint32_t)level0_rays, (int)ctx->lights.size());
        kt.end();
        if (st) {  // launches that walk a level's queue (what a kernel trace lists as k_closest<.., 0> / k_any<.., 0>): the
                   // hand-over launch behind a packet level and the split rounds are not counted
            if (!(level == 0 && level0_traced) && !rtLevelUsesPackets(cfg, level, false)) st->launches_closest++;
            if (!rtLevelUsesPackets(cfg, level, true)) st->launches_any++;
        }
    }
}

// parameter checks + camera / image constants of the tile loop (main.cpp:561-563), shared by the render entry points
int makeFrame(rt_context *ctx, const rt_camera *cam, const rt_render_params *p, RtFrameDev &f, int &world) {
    if (!ctx->committed) return fail("rt_commit has not been called");
    if (p->width <= 0 || p->height <= 0 || p->spp <= 0 || p->max_depth < 0 || p->max_depth > 15)
        return fail("bad render parameters (width/height/spp > 0, 0 <= max_depth <= 15)");
    world = p->tile_world > 0 ? p->tile_world : 1;
    if (p->tile_rank < 0 || p->tile_rank >= world) return fail("tile_rank out of range");
    const bool table = p->jitter_mode == RT_JITTER_TABLE || (p->jitter_mode == RT_JITTER_AUTO && p->spp == 4);
    if (table && p->spp != 4) return fail("RT_JITTER_TABLE requires spp == 4 (reference jitterMatrix, main.cpp:601-606)");
    f = RtFrameDev{};
    std::memcpy(f.origin, cam->origin, sizeof f.origin);
    std::memcpy(f.rot, cam->rot, sizeof f.rot);
    f.dz = cam->dz;
    // main.cpp:561-563: computed in double, stored as float
    const float fov = cam->fov_deg;
    const float s = 2 * tan(0.5 * fov / 180 * M_PI) / p->width;
    f.s = s;
    f.X = -s * p->width / 2;
    f.Y = s * p->height / 2;
    f.width = p->width;
    f.height = p->height;
    f.spp = p->spp;
    f.tiles_x = (p->width + RT_TILE - 1) / RT_TILE;
    f.tile_rank = p->tile_rank;
    f.tile_world = world;
    f.use_table = table ? 1 : 0;
    f.seed = p->seed;
    return 0;
}

int renderTiles(rt_context *ctx, const rt_camera *cam, const rt_render_params *p, uint8_t *d_rgb8, float *d_rgbf,
                uint32_t *d_primid, hipStream_t stream, rt_stats *stats) {
    RtFrameDev f{};
    int world = 1;
    if (makeFrame(ctx, cam, p, f, world)) return 1;
    if (bind(ctx)) return 1;

    int64_t n_local = 0, n_total = 0;
    rt_tile_counts(p->width, p->height, p->tile_rank, world, &n_local, &n_total);
    const size_t per_tile = (size_t)RT_TILE_PIXELS * p->spp;
    if (per_tile > RT_SAMPLE_MASK) return fail("spp too large for one tile per chunk");
    const int fo = worstFanout(p->max_depth), nl = std::max<int>(1, (int)ctx->lights.size());

    RtLaunchCfg cfg = ctx->cfg;
    cfg.stream = stream;
    const int stats_on = (p->collect_stats & 4) ? 2 : (p->collect_stats & 1) ? 1 : 0;  // counters / + wave-time profile
    KernelTimer kt;
    kt.on = (p->collect_stats & 2) != 0;
    kt.stream = stream;
    rt_stats local{};
    rt_stats *st = stats ? stats : &local;
    std::memset(st, 0, sizeof *st);

    hipEvent_t ev0, ev1;
    HIPOK(hipEventCreate(&ev0));
    HIPOK(hipEventCreate(&ev1));
    bool started = false;  // ev0 is recorded once the first chunk's workspace exists: ms_total is GPU time, not hipMalloc time
    // Chunks of whole tiles.  Large chunks are faster (deep levels hold few rays, and every launch ends with its longest
    // ray: 8 chunks of 64 Mi samples cost 7 % more than 2 of 265 M on the 4K / 64 spp frame), but a chunk's queues must
    // hold its fullest level, and the WORST case (every hit spawns a pair, every light faces every hit: 2^depth rays and
    // 2^depth x lights shadow rays per sample) is 5-20 x what scenes produce.  So: the first chunk after rt_commit is
    // sized for the worst case (64 Mi samples, halved until it fits 32-bit indices and the free memory); its counters
    // give the fan-out this scene really has, and later chunks -- of this frame and of the following ones -- are sized
    // for 1.5 x the largest fan-out seen so far.  A chunk that overflows its queues after all (the kernels drop and
    // count what does not fit; nothing is written out of bounds) is rendered again at worst-case size: outputs are only
    // written by a chunk's last kernels and the accumulators are initialised by its first, so a repeat is clean.
    // A caller-given chunk size is taken as is, with worst-case queues.
    int l0_mode = -1;  // level-0 shadow results stored instead of added: -1 = where it pays (below), 0 = never, 1 = always
    if (const char *e = getenv("RT_L0_STORE")) l0_mode = atoi(e) != 0 ? 1 : 0;
    double margin = 1.5, ws_frac = 0.5;
    if (const char *e = getenv("RT_CHUNK_MARGIN")) margin = atof(e);  // (test knob: < 1 forces the overflow-and-repeat path)
    if (const char *e = getenv("RT_WORKSPACE_FRACTION")) ws_frac = std::min(0.875, std::max(0.05, atof(e)));
    std::vector<uint32_t> hc(RT_CNT_WORDS, 0);
    size_t done = 0;            // tiles of this rank rendered so far
    bool force_worst = false;   // the chunk being repeated after an overflow
    const size_t nlights = ctx->lights.size();
    // level-0 shadow results stored instead of added (RtQueues::occ0, decided per chunk below): its buffers count
    auto wantL0 = [&]() { return nlights > 0 && (l0_mode == 1 || (l0_mode < 0 && ctx->obs_valid && ctx->obs_l0_share >= 0.6)); };
    auto l0Bytes = [&](size_t samples) { return wantL0() ? samples * nlights * (sizeof(float4) + 1) : (size_t)0; };
    struct Plan { size_t tiles, samples; QueueSizes z; bool optimistic; };
    auto planChunk = [&](size_t remaining, bool worst, Plan &pl) -> int {
        const bool single = (p->flags & RT_RENDER_SINGLE_FRAME) != 0;
        pl.optimistic = p->chunk_samples <= 0 && !worst && ctx->obs_valid && p->max_depth <= ctx->obs_depth;
        const double ray_fan = pl.optimistic ? std::min((double)fo, std::max(1.0, margin * ctx->obs_ray)) : (double)fo;
        const double sh_fan = pl.optimistic ? std::min((double)fo * nl, std::max(1.0, margin * ctx->obs_sh)) : (double)fo * nl;
        // (a single frame keeps its worst-case first chunk small: that chunk's queues are all it will ever allocate)
        size_t chunk_samples = p->chunk_samples > 0 ? (size_t)p->chunk_samples
                               : pl.optimistic ? (size_t)RT_SAMPLE_MASK : (size_t)1 << (single ? 24 : 26);
        chunk_samples = std::min<size_t>(std::min<size_t>(chunk_samples, RT_SAMPLE_MASK), remaining * per_tile);
        if (p->chunk_samples <= 0) {
            // what fits 32-bit indices and 7/8 of the memory that is FREE on this device (torch, RCCL or other contexts
            // may hold part of the HBM) plus what this context's workspace already holds
            size_t free_b = 0, total_b = 0;
            HIPOK(hipMemGetInfo(&free_b, &total_b));
            // Worst-case queues may take 7/8 of it (deep bounce budgets need the room to fit at all); queues sized from
            // the observed fan-out stop at `ws_frac` (half by default, RT_WORKSPACE_FRACTION): larger chunks than that buy
            // ~1 % (profiles/r02_chunks.md) and the memory has other users (scene, outputs, torch, RCCL).
            const size_t avail = free_b + ctx->ws_bytes + ctx->d_occ0.n + ctx->d_col0.n * sizeof(float4);
            const size_t budget = pl.optimistic ? (size_t)((double)avail * ws_frac) : avail / 8 * 7;
            while (chunk_samples > per_tile) {
                const QueueSizes z = queueSizes(ctx, chunk_samples, ray_fan, sh_fan, nl);
                if (z.fits32 && z.bytes + z.bytes / 8 + l0Bytes(chunk_samples) <= budget) break;  // (+ 1/8: the headroom a growing queue gets)
                chunk_samples = pl.optimistic ? chunk_samples / 8 * 7 : chunk_samples / 2;
            }
            if (single && pl.optimistic && ctx->q.ray_cap) {  // no growth inside a single frame: what the queues hold as they are
                while (chunk_samples > per_tile) {
                    const QueueSizes z = queueSizes(ctx, chunk_samples, ray_fan, sh_fan, nl);
                    if (z.ray_cap <= ctx->q.ray_cap && z.sh_cap <= ctx->q.shadow_cap) break;
                    chunk_samples = chunk_samples / 8 * 7;
                }
            }
        }
        pl.tiles = std::min(std::max<size_t>(1, chunk_samples / per_tile), remaining);
        {  // equal chunks over what is left: no small straggler at the end of the frame
            const size_t nc = (remaining + pl.tiles - 1) / pl.tiles;
            pl.tiles = (remaining + nc - 1) / nc;
        }
        pl.samples = pl.tiles * per_tile;
        pl.z = queueSizes(ctx, pl.samples, ray_fan, sh_fan, nl);
        return 0;
    };
    while (done < (size_t)n_local) {
        Plan pl;
        if (planChunk((size_t)n_local - done, force_worst, pl)) return 1;
        const bool optimistic = pl.optimistic;
        const size_t chunk_tiles = pl.tiles, samples = pl.samples;
        QueueSizes need = pl.z;
        size_t need_samples = samples;
        if (optimistic && !(p->flags & RT_RENDER_SINGLE_FRAME) && (size_t)n_local > chunk_tiles) {
            // queues that must grow now grow to what a repeat of the WHOLE frame will want: one reallocation, not two
            Plan whole;
            if (planChunk((size_t)n_local, false, whole)) return 1;
            need.ray_cap = std::max(need.ray_cap, whole.z.ray_cap);
            need.sh_cap = std::max(need.sh_cap, whole.z.sh_cap);
            need.fits32 = need.fits32 && whole.z.fits32;
            need_samples = std::max(need_samples, whole.samples);
        }
        if (ensureWorkspace(ctx, need_samples, need)) return 1;
        if (!started) {
            HIPOK(hipEventRecord(ev0, stream));
            if (d_primid) HIPOK(hipMemsetAsync(d_primid, 0xFF, (size_t)n_local * RT_TILE_PIXELS * sizeof(uint32_t), stream));
            started = true;
        }
        const uint32_t tile0 = (uint32_t)done, ntiles = (uint32_t)chunk_tiles;
        const rt_stats before = *st;
        // Level-0 shadow results stored instead of added (RtQueues::occ0) where most of the scene's shadow rays are
        // level-0 rays: their three 64-bit atomics each are then what bounds k_any (bunny scene: 88 -> 27 ms), while a
        // frame whose shadow rays mostly come from deeper levels hides its atomics behind the walks and would only pay for
        // the extra pass.  Either way the sums are the same integers.  RT_L0_STORE=0 / 1 forces the choice.
        RtQueues cq = ctx->q;
        const bool l0 = wantL0();
        if (l0) {
            if (ctx->d_occ0.alloc(samples * nlights) || ctx->d_col0.alloc(samples * nlights)) return 1;
            cq.occ0 = ctx->d_occ0.p;
            cq.col0 = ctx->d_col0.p;
            cq.l0_stride = (uint32_t)samples;
            HIPOK(hipMemsetAsync(cq.occ0, 0, samples * nlights, stream));
        }
        HIPOK(hipMemsetAsync(ctx->q.counters, 0, RT_CNT_WORDS * sizeof(uint32_t), stream));
        // (the sample accumulators need no clearing: level-0 k_shade stores every slot of the chunk)
        kt.begin(0);
        rtLaunchGenPrimary(cfg, f, cq, tile0, ntiles);
        kt.end();
        runLevels(ctx, cq, cfg, p->max_depth, p->spp, tile0, d_primid, stats_on, kt, st, (uint64_t)samples);
        kt.begin(4);
        rtLaunchResolve(cfg, f, cq, tile0, ntiles, d_rgb8, d_rgbf);
        kt.end();
        HIPOK(hipMemcpyAsync(hc.data(), ctx->q.counters, RT_CNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIPOK(hipStreamSynchronize(stream));
        HIPOK(hipGetLastError());
        if (hc[RT_CNT_OVERFLOW] && optimistic) {  // the scene out-grew what was seen so far: again, sized for the worst case
            *st = before;  // (launch counts of the discarded attempt)
            st->chunks++;
            st->chunk_retries++;
            ctx->obs_valid = false;
            force_worst = true;
            continue;
        }
        st->chunks++;
        addCounters(st, hc.data(), p->max_depth);
        if (!hc[RT_CNT_OVERFLOW]) {
            double r = 1.0, sh = 0.0;
            for (int d = 0; d <= p->max_depth && d < 16; d++) {
                r = std::max(r, (double)hc[RT_CNT_RAYS(d)] / (double)samples);
                sh = std::max(sh, (double)hc[RT_CNT_SHADOW(d)] / (double)samples);
            }
            {
                double s0 = (double)hc[RT_CNT_SHADOW(0)] - 0.0, sall = 0.0;
                for (int d = 0; d <= p->max_depth && d < 16; d++) sall += (double)hc[RT_CNT_SHADOW(d)];
                ctx->obs_l0_share = sall > 0 ? s0 / sall : 0.0;
                if (getenv("RT_VERBOSE")) std::fprintf(stderr, "[rt] chunk of %zu samples: level-0 share of the shadow rays %.2f\n", samples, ctx->obs_l0_share);
            }
            if (!ctx->obs_valid) ctx->obs_ray = ctx->obs_sh = 0.0;
            ctx->obs_ray = std::max(ctx->obs_ray, r);
            ctx->obs_sh = std::max(ctx->obs_sh, sh);
            ctx->obs_depth = ctx->obs_valid ? std::max(ctx->obs_depth, p->max_depth) : p->max_depth;
            ctx->obs_valid = true;
        }
        force_worst = false;
        done += chunk_tiles;
    }
    if (!started) HIPOK(hipEventRecord(ev0, stream));
    HIPOK(hipEventRecord(ev1, stream));
    HIPOK(hipStreamSynchronize(stream));
    HIPOK(hipGetLastError());
    float ms = 0;
    HIPOK(hipEventElapsedTime(&ms, ev0, ev1));
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    st->ms_total = ms;
    if (n_local > 0 && p->chunk_samples <= 0 && !(p->flags & RT_RENDER_SINGLE_FRAME)) {
        // What a repeat of this frame will want, now that its fan-out is known: growing tens of GB of queues takes
        // seconds (hipFree + hipMalloc), so it is done here, at the end of the frame that found out, not inside the next
        // one.  A no-op in steady state; a failure is not this frame's (the next call sizes its workspace anew).
        Plan pl;
        if (!planChunk((size_t)n_local, false, pl)) {
            (void)ensureWorkspace(ctx, pl.samples, pl.z);
            if (wantL0()) {
                (void)ctx->d_occ0.alloc(pl.samples * nlights);
                (void)ctx->d_col0.alloc(pl.samples * nlights);
            }
        }
    }
    double cls[6] = {0, 0, 0, 0, 0, 0};
    kt.collect(cls);
    st->ms_gen = cls[0]; st->ms_closest = cls[1]; st->ms_shade = cls[2]; st->ms_any = cls[3]; st->ms_resolve = cls[4];
    st->ms_packet = cls[5];
    // primary samples = valid pixels of this rank's tiles x spp
    {
        uint64_t pix = 0;
        const int tiles_x = f.tiles_x, tiles_y = (p->height + RT_TILE - 1) / RT_TILE;
        for (int t = p->tile_rank; t < tiles_x * tiles_y; t += world) {
            const int tx = t % tiles_x, ty = t / tiles_x;
            pix += (uint64_t)std::min(RT_TILE, p->width - tx * RT_TILE) * std::min(RT_TILE, p->height - ty * RT_TILE);
        }
        st->primary_samples = pix * p->spp;
    }
    st->bvh_nodes = ctx->info.n_nodes;
    st->bvh_depth = ctx->bvh.depth;
    st->bvh_max_stack = ctx->bvh.max_stack;
    st->n_triangles = ctx->info.n_triangles;
    if (st->queue_overflow) return fail("wavefront queue overflow (" + std::to_string(st->queue_overflow) + " rays dropped)");
    return 0;
}

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
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return fail("rt_create: hipStreamCreate failed");
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
    ctx->d_records.release(); ctx->d_spill.release(); ctx->d_nodes_p.release(); ctx->d_resid.release(); ctx->d_split[0].release(); ctx->d_split[1].release(); ctx->d_shade.release(); ctx->d_spheres.release(); ctx->d_planes.release();
    ctx->d_materials.release(); ctx->d_objmat.release(); ctx->d_lights.release();
    for (int k = 0; k < 4; k++) ctx->d_bmp[k].release();
    for (int k = 0; k < 2; k++) { ctx->d_rayA[k].release(); ctx->d_rayB[k].release(); ctx->d_rayR[k].release(); }
    ctx->d_hits.release(); ctx->d_shA.release(); ctx->d_shC.release(); ctx->d_shL.release(); ctx->d_occ0.release(); ctx->d_col0.release(); ctx->d_accum.release(); ctx->d_general.release();
    ctx->d_counters.release();
    ctx->d_t8.release(); ctx->d_o8.release(); ctx->d_tf.release(); ctx->d_of.release(); ctx->d_tp.release(); ctx->d_op.release();
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
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
    std::vector<RtNode4> records;   // BVH nodes, then one record per triangle in leaf (slot) order
    std::vector<RtTriShade> shade;
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
    if (nn + (size_t)nt >= (1ull << 28)) return fail("rt_commit: more than 2^28 BVH records");
    img.records.resize(nn + (size_t)nt);
    if (nn) std::memcpy(img.records.data(), ctx->bvh.nodes.data(), nn * sizeof(RtNode4));
    RtTriGeom *geom = reinterpret_cast<RtTriGeom *>(img.records.data() + nn);
    img.shade.resize((size_t)nt);
    for (int64_t slot = 0; slot < nt; slot++) {
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
        ctx->commit_stats = src->commit_stats;
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
    d.nodes_p = ctx->d_nodes_p.p;
    d.tri_shade = ctx->d_shade.p;
    d.spheres = ctx->d_spheres.p;
    d.planes = ctx->d_planes.p;
    d.materials = ctx->d_materials.p;
    d.object_material = ctx->d_objmat.p;
    d.lights = ctx->d_lights.p;
    std::memcpy(d.ambient, src->ambient, 12);
    d.n_nodes = (int)nn;
    d.n_tris = (int)nt;
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
    if (const char *e = getenv("RT_PACKET_LEVELS")) {  // tuning knob: "<closest levels>,<shadow levels>"
        int a = 1, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2) { ctx->cfg.packet_levels_closest = a; ctx->cfg.packet_levels_any = b; }
    }
    ctx->dev.any_ordered = 0;  // tuning knob: shadow rays descend near-first (costs ~1 % on the competition scene)
    if (const char *e = getenv("RT_ANY_ORDERED")) ctx->dev.any_ordered = atoi(e) != 0;
    ctx->dev.occluder_cache = 1;  // RT_OCCLUDER_CACHE=0: shadow queries independent of scheduling down to the last float bit
    if (const char *e = getenv("RT_OCCLUDER_CACHE")) ctx->dev.occluder_cache = atoi(e) != 0;
    ctx->cfg.analytic_bytes = (int)(img.units.size() * sizeof(RtSphere) + src->planes.size() * sizeof(RtPlane));
    const int lds_per_block = (entries + 1) * 256 * 4 + ctx->cfg.analytic_bytes;  // stack rows + the age row + analytic primitives
    if (const char *e = getenv("RT_SPLIT_BUDGET")) {  // 0 = never split a shadow walk; rounded down to a power of two
        int b = std::max(0, atoi(e)), p2 = 0;
        while (b > 1) { b >>= 1; p2++; }
        ctx->cfg.split_budget = atoi(e) > 0 ? 1 << p2 : 0;
    }
    ctx->cfg.trace_blocks_per_cu = std::max(1, std::min(8, (160 * 1024) / lds_per_block));
    if (const char *e = getenv("RT_TRACE_BLOCKS")) ctx->cfg.trace_blocks_per_cu = std::max(1, std::min(ctx->cfg.trace_blocks_per_cu, atoi(e)));  // (tuning knob)
    const size_t spill_entries = (size_t)std::max(0, src->bvh.max_stack - entries);
    if (ctx->d_spill.alloc(std::max<size_t>(1, spill_entries * (size_t)ctx->cfg.num_cus * 8 * 256))) return 1;
    ctx->q.spill = ctx->d_spill.p;
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

// ---- rt_render_tiles_static: the render loop on the library's static scratch, allocating nothing ----------------
namespace {
#define RT_STATIC_SHADE_BLOCKS 64   // k_shade grid of the static path: 256 waves, each owning (lights + 1) output slabs
#define RT_STATIC_SLAB 64u          // slots per shadow slab (ray slabs: 128)
#define RT_STATIC_SPLIT_CAP 65536u  // work items of split any-hit walks per round
// carves the queues of `samples` primary samples out of [base, base + bytes); returns false if they do not fit
bool carveStatic(unsigned char *base, size_t bytes, size_t samples, int fanout, int nl, uint32_t *spill, RtQueues &q) {
    const size_t waves = (size_t)RT_STATIC_SHADE_BLOCKS * (RT_BLOCK / 64);
    const size_t total = samples * (size_t)fanout;
    const size_t ray_cap = total + 2 * waves * 128 + total / 64 + 128;        // as queueSizes(), with this path's slab sizes
    const size_t sh_cap = total * nl + 2 * waves * nl * RT_STATIC_SLAB;
    if (ray_cap >= (1ull << 31) || sh_cap >= (1ull << 31)) return false;
    size_t off = 0;
    auto take = [&](size_t n) -> void * {
        void *p = base + off;
        off += (n + 255) / 256 * 256;
        return p;
    };
    q = RtQueues{};
    q.counters = (uint32_t *)take(RT_CNT_WORDS * sizeof(uint32_t));
    for (int k = 0; k < 2; k++) {
        q.rayA[k] = (float4 *)take(ray_cap * sizeof(float4));
        q.rayB[k] = (float4 *)take(ray_cap * sizeof(float4));
        q.rayR[k] = (float *)take(ray_cap * sizeof(float));
        q.split[k] = (uint2 *)take(RT_STATIC_SPLIT_CAP * sizeof(uint2));
    }
    q.hits = (float2 *)take((ray_cap + 1) * sizeof(float2));
    q.shA = (float4 *)take(sh_cap * sizeof(float4));
    q.shC = (float4 *)take(sh_cap * sizeof(float4));
    q.shL = (uint8_t *)take(sh_cap);
    q.occ0 = nullptr;
    q.col0 = nullptr;
    q.l0_stride = 0;
    q.accum = (long long *)take(samples * 3 * sizeof(long long));
    q.resid = (uint32_t *)take(std::max(sh_cap, ray_cap) * sizeof(uint32_t));
    q.general = (uint32_t *)take(ray_cap * sizeof(uint32_t));
    q.spill = spill;
    q.ray_cap = (uint32_t)ray_cap;
    q.shadow_cap = (uint32_t)sh_cap;
    q.split_cap = RT_STATIC_SPLIT_CAP;
    return off <= bytes;
}
}  // namespace

int rt_render_tiles_static(rt_context *ctx, const rt_camera *cam, const rt_render_params *p, void *d_rgb8, void *d_rgbf,
                           void *d_primid, void *stream_) {
    if (!ctx || !cam || !p) return fail("rt_render_tiles_static: NULL argument");
    RtFrameDev f{};
    int world = 1;
    if (makeFrame(ctx, cam, p, f, world)) return 1;
    if (bind(ctx)) return 1;
    if (!ctx->scratch) return fail("rt_render_tiles_static: the static scratch was not resolved on this device");
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : ctx->stream;
    int64_t n_local = 0;
    rt_tile_counts(p->width, p->height, p->tile_rank, world, &n_local, nullptr);
    const size_t per_tile = (size_t)RT_TILE_PIXELS * p->spp;
    const int fo = worstFanout(p->max_depth), nl = std::max<int>(1, (int)ctx->lights.size());
    // whole tiles per chunk: as many as the scratch holds (binary search on the carve)
    RtQueues q{};
    size_t lo = 0, hi = std::max<int64_t>(n_local, 1);
    while (lo < hi) {
        const size_t mid = (lo + hi + 1) / 2;
        RtQueues t;
        if (mid * per_tile <= RT_SAMPLE_MASK && carveStatic((unsigned char *)ctx->scratch, ctx->scratch_bytes, mid * per_tile, fo, nl, ctx->d_spill.p, t)) lo = mid;
        else hi = mid - 1;
    }
    if (lo == 0)
        return fail("rt_render_tiles_static: one 16x16 tile at " + std::to_string(p->spp) + " spp, depth " + std::to_string(p->max_depth) +
                    " with " + std::to_string(nl) + " lights does not fit the " + std::to_string(ctx->scratch_bytes >> 20) +
                    " MiB static scratch (lower spp or max_depth, or use rt_render_tiles_device)");
    const size_t chunk_tiles = lo;
    carveStatic((unsigned char *)ctx->scratch, ctx->scratch_bytes, chunk_tiles * per_tile, fo, nl, ctx->d_spill.p, q);
    RtLaunchCfg cfg = ctx->cfg;
    cfg.stream = stream;
    cfg.shade_blocks = RT_STATIC_SHADE_BLOCKS;
    cfg.shade_slab = RT_STATIC_SLAB;
    KernelTimer kt;  // (off)
    if (d_primid) HIPOK(hipMemsetAsync(d_primid, 0xFF, (size_t)n_local * RT_TILE_PIXELS * sizeof(uint32_t), stream));
    for (size_t t0 = 0; t0 < (size_t)n_local; t0 += chunk_tiles) {
        const uint32_t ntiles = (uint32_t)std::min<size_t>(chunk_tiles, (size_t)n_local - t0);
        // the scratch keeps the last call's contents: the counters must start at zero (queue lengths, work cursors);
        // queues and accumulators are written before they are read (level-0 k_shade stores every sample slot)
        HIPOK(hipMemsetAsync(q.counters, 0, RT_CNT_WORDS * sizeof(uint32_t), stream));
        rtLaunchGenPrimary(cfg, f, q, (uint32_t)t0, ntiles);
        runLevels(ctx, q, cfg, p->max_depth, p->spp, (uint32_t)t0, (uint32_t *)d_primid, 0, kt, nullptr, (uint64_t)ntiles * per_tile);
        rtLaunchResolve(cfg, f, q, (uint32_t)t0, ntiles, (uint8_t *)d_rgb8, (float *)d_rgbf);
    }
    HIPOK(hipGetLastError());
    return 0;
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
    // rays per pass: 1 Mi, halved until the worst case of the deepest level (2^depth rays per input ray) fits 32-bit
    // queue indices and the free memory
    size_t chunk = std::min<size_t>(std::max<int64_t>(n, 1), (size_t)1 << 20);
    {
        size_t free_b = 0, total_b = 0;
        HIPOK(hipMemGetInfo(&free_b, &total_b));
        const size_t budget = (free_b + ctx->ws_bytes) / 8 * 7;
        const int fo = worstFanout(depth), nl = std::max<int>(1, (int)ctx->lights.size());
        while (chunk > 1) {
            const QueueSizes z = queueSizes(ctx, chunk, fo, nl);
            if (z.fits32 && z.bytes <= budget) break;
            chunk /= 2;
        }
    }
    if (ensureWorkspace(ctx, chunk, depth, (int)ctx->lights.size())) return 1;
    RtLaunchCfg cfg = ctx->cfg;
    cfg.stream = ctx->stream;
    DevBuf<float> d_rays, d_col;
    DevBuf<rt_hit> d_hits;
    DevBuf<uint8_t> d_sh;
    KernelTimer kt;
    int rc = 0;
    for (int64_t off = 0; off < n && !rc; off += (int64_t)chunk) {
        const uint32_t m = (uint32_t)std::min<int64_t>(chunk, n - off);
        rc = d_rays.alloc((size_t)m * 6) || d_col.alloc((size_t)m * 3) || d_hits.alloc(m) || d_sh.alloc((size_t)m * 4);
        if (rc) break;
        hipError_t e = hipMemcpyAsync(d_rays.p, rays + 6 * off, (size_t)m * 6 * sizeof(float), hipMemcpyHostToDevice, cfg.stream);
        if (e == hipSuccess) e = hipMemsetAsync(ctx->q.counters, 0, RT_CNT_WORDS * sizeof(uint32_t), cfg.stream);
        if (e != hipSuccess) { rc = fail(hipGetErrorString(e)); break; }
        rtLaunchRaysFromHost(cfg, d_rays.p, m, ctx->q);
        // level 0 closest hit first: the hit records of the batch are reported from it
        rtLaunchClosest(cfg, ctx->dev, ctx->bvh.root_ref, ctx->q, 0, 0);
        if (out_hits || out_shadow) rtLaunchHitInfo(cfg, ctx->dev, ctx->bvh.root_ref, ctx->q, m, out_hits ? d_hits.p : nullptr, out_shadow ? d_sh.p : nullptr);
        runLevels(ctx, ctx->q, cfg, depth, 1, 0, nullptr, 0, kt, st, m, true);
        rtLaunchBatchColors(cfg, ctx->q, m, d_col.p);
        std::vector<uint32_t> hc(RT_CNT_WORDS);
        e = hipStreamSynchronize(cfg.stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(hc.data(), ctx->q.counters, RT_CNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_color) e = hipMemcpy(out_color + 3 * off, d_col.p, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_hits) e = hipMemcpy(out_hits + off, d_hits.p, (size_t)m * sizeof(rt_hit), hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_shadow) e = hipMemcpy(out_shadow + 4 * off, d_sh.p, (size_t)m * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = fail(hipGetErrorString(e)); break; }
        addCounters(st, hc.data(), depth);
        if (out_hits) {  // slot -> triangle index inside its mesh
            for (uint32_t k = 0; k < m; k++) {
                rt_hit &h = out_hits[off + k];
                if (h.hit && h.triangle >= 0) h.triangle = (int32_t)((int64_t)h.prim - ctx->mesh_first[h.object]);
            }
        }
    }
    d_rays.release(); d_col.release(); d_hits.release(); d_sh.release();
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
    // containment: walking down from the root, every triangle must lie inside the decoded (quantised) box and the
    // float box of every child slot above it, and every triangle must be reached exactly once
    int64_t bad = 0;
    std::vector<uint8_t> seen((size_t)n, 0);
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
                if (s >= (uint32_t)n) { bad++; continue; }
                const uint32_t t = bvh.order[s];
                seen[t]++;
                const float *vs[3] = {v0 + 3 * (size_t)t, v1 + 3 * (size_t)t, v2 + 3 * (size_t)t};
                for (int v = 0; v < 3; v++)
                    for (int ax = 0; ax < 3; ax++)
                        if (vs[v][ax] < it.lo[ax] || vs[v][ax] > it.hi[ax]) bad++;
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
        if (seen[t] != 1) bad++;
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

struct rt_multi {
    int n = 0;
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
    *out = m;
    return 0;
}

int rt_multi_destroy(rt_multi *m) {
    if (!m) return 0;
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
    std::vector<std::thread> th;
    for (int k = 0; k < m->n; k++)
        th.emplace_back([&, k] {
            rc[k] = commitDevice(m->ctx[k], src, img);
            if (rc[k]) err[k] = g_err;
        });
    for (std::thread &t : th) t.join();
    for (int k = 0; k < m->n; k++)
        if (rc[k]) return fail("rt_multi_commit: rank " + std::to_string(k) + ": " + err[k]);
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
    {
        std::vector<std::thread> th;
        for (int k = 0; k < n; k++)
            th.emplace_back([&, k] {
                rt_render_params q = *p;
                q.tile_rank = k;
                q.tile_world = n;
                rc[k] = renderTiles(m->ctx[k], cam, &q, local8(k), localf(k), localp(k), m->ctx[k]->stream, &st[k]);
                if (rc[k]) err[k] = g_err;
            });
        for (std::thread &t : th) t.join();
    }
    for (int k = 0; k < n; k++)
        if (rc[k]) return fail("rt_multi_render: rank " + std::to_string(k) + ": " + err[k]);
    // ---- the one exchange step ----
    hipStream_t s0 = m->ctx[0]->stream;
    const int first = m->self_rccl ? 0 : 1;
    if (m->rccl) {
        NCCLOK(g_rccl.groupStart());
        for (int k = first; k < n; k++) {
            NCCLOK(g_rccl.recv(m->g8.p + (size_t)k * slot * 3, slot * 3, ncclUint8, k, m->comms[0], s0));
            NCCLOK(g_rccl.recv(m->gp.p + (size_t)k * slot, slot, ncclUint32, k, m->comms[0], s0));
            if (want_f) NCCLOK(g_rccl.recv(m->gf.p + (size_t)k * slot * 3, slot * 3, ncclFloat32, k, m->comms[0], s0));
            hipStream_t sk = m->ctx[k]->stream;
            NCCLOK(g_rccl.send(m->t8[k].p, slot * 3, ncclUint8, 0, m->comms[k], sk));
            NCCLOK(g_rccl.send(m->tp[k].p, slot, ncclUint32, 0, m->comms[k], sk));
            if (want_f) NCCLOK(g_rccl.send(m->tf[k].p, slot * 3, ncclFloat32, 0, m->comms[k], sk));
        }
        NCCLOK(g_rccl.groupEnd());
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
    if (rt_assemble_tiles_device(m->ctx[0], p->width, p->height, n, per, m->g8.p, want_f ? m->gf.p : nullptr, m->gp.p, m->o8.p,
                                 want_f ? m->of.p : nullptr, m->op.p, s0))
        return 1;
    if (out_rgb8) HIPOK(hipMemcpy(out_rgb8, m->o8.p, npix * 3, hipMemcpyDeviceToHost));
    if (out_rgbf) HIPOK(hipMemcpy(out_rgbf, m->of.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (out_primid) HIPOK(hipMemcpy(out_primid, m->op.p, npix * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (stats)
        for (int k = 0; k < n; k++) stats[k] = st[k];
    return 0;
}

}  // extern "C"
