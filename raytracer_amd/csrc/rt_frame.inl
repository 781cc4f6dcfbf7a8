// rt_frame.inl -- the frame pipeline behind rt_render* and rt_trace_batch: workspace carving, frame planning, the
// iteration loop.  Included by rt_abi.cpp inside its anonymous namespace (it uses rt_context, DevBuf, fail, HIPOK).
//
// A frame is rendered as a PIPELINE of chunks of whole 16x16 tiles (stands where `#pragma omp parallel for
// schedule(dynamic, 1)` kept every core busy whatever a tile cost, main.cpp:607): iteration j traces and shades, in ONE
// pass per kernel, level 0 of chunk j, level 1 of chunk j-1, ... level D of chunk j-D.  Every launch therefore holds the
// full mix of levels whatever the chunk size, a frame of C chunks takes C + D iterations instead of C x (D + 1) level
// passes, and the chunk size -- i.e. the workspace -- stops deciding the speed.  The ray queues ping-pong by iteration:
// k_shade of iteration j appends the bounce rays to the other queue, BEHIND the slots reserved for the primary rays of
// the chunk that enters at j + 1.  Sample accumulators form a ring of D + 1 chunk-sized slots; a chunk is resolved when
// its level D has been shaded and its last shadow rays traced (iteration j + D).  Between iterations nothing goes through
// the host: k_iter_ctl folds the iteration's 32-bit counters into 64-bit frame totals and prepares the next queue.

// bits of a ray's tag that hold its level (rt_types.h): 0 for depth 0, 2 for the reference's depth 3, 4 for depth 15
int levelBits(int max_depth) {
    int b = 0;
    while ((1 << b) <= max_depth) b++;
    return b;
}

// Worst-case entries per primary sample (of ONE chunk) in the ray queue of an iteration with `in_flight` chunks in the
// pipeline: their levels D, D-1, ... together, 2^l rays at level l -- every hit spawns a reflection / refraction pair
// (main.cpp:148-205).  Shadow rays: that times the lights.
double worstFan(int max_depth, size_t in_flight) {
    double f = 0.0;
    for (int l = max_depth; l >= 0 && (size_t)(max_depth - l) < in_flight; l--) f += (double)(1u << l);
    return f;
}

// Queue capacities for chunks of `samples` primary samples: what the rays need plus what k_shade's wave-private slabs
// may leave unused (the fast and the general instance each leave one partly filled slab per wave and queue; one slot per
// slab where a reflection / refraction pair met its end).
struct QueueSizes {
    size_t ray_cap = 0, sh_cap = 0;
    uint32_t slab = 64;
    bool fits32 = false;
};
QueueSizes queueSizes(const RtLaunchCfg &cfg, size_t samples, double ray_fan, double sh_fan, int nl) {
    QueueSizes z;
    const size_t waves = (size_t)rtShadeWaves(cfg);
    const size_t rays = (size_t)std::ceil((double)samples * ray_fan), shadows = (size_t)std::ceil((double)samples * sh_fan);
    z.slab = rtShadeSlab(cfg, rays);
    const size_t slabR = std::max<size_t>(z.slab, 128);
    z.ray_cap = rays + 2 * waves * slabR + rays / 64 + 128;
    z.sh_cap = shadows + 2 * waves * (size_t)nl * z.slab;
    // (k_shade's segment cursor and the traversal cursors run past the end of a queue by up to one fetch per wave)
    z.fits32 = z.ray_cap < (1ull << 32) - (1ull << 26) && z.sh_cap < (1ull << 32) - (1ull << 26);
    return z;
}

// Carves the wavefront workspace out of [base, ...): returns the bytes it takes.  base == nullptr only measures.
struct CarveSpec {
    size_t samples = 0;    // primary samples per chunk
    int ring = 1;          // chunks in flight = slots of the accumulator ring
    size_t ray_cap = 0, sh_cap = 0, split_cap = 0;
    size_t general_cap = 0;  // entries of the list of hits for the general k_shade (0 for scenes without such primitives)
    int nl = 1;
    bool l0 = false;       // level-0 shadow results stored, not added (RtQueues::occ0)
    int max_depth = 0;
};
size_t carveQueues(unsigned char *base, const CarveSpec &c, uint32_t *spill, RtQueues &q) {
    size_t off = 0;
    auto take = [&](size_t n) -> void * {
        void *p = base ? base + off : nullptr;
        off += (n + 255) / 256 * 256;
        return p;
    };
    q = RtQueues{};
    q.counters = (uint32_t *)take(RT_CNT_WORDS * sizeof(uint32_t));
    for (int k = 0; k < 2; k++) {
        q.rayA[k] = (float4 *)take(c.ray_cap * sizeof(float4));
        q.rayB[k] = (float4 *)take(c.ray_cap * sizeof(float4));
        q.rayR[k] = (float *)take(c.ray_cap * sizeof(float));
        q.split[k] = (uint2 *)take(c.split_cap * sizeof(uint2));
    }
    q.hits = (float2 *)take((c.ray_cap + 1) * sizeof(float2));
    q.shA = (float4 *)take(c.sh_cap * sizeof(float4));
    q.shC = (float4 *)take(c.sh_cap * sizeof(float4));
    q.shSlabLight = (uint8_t *)take(c.sh_cap / 64 + 1);  // (slabs hold at least 64 slots)
    q.sh_slab_shift = 6;                                 // (rtSetSlab: the frame's slab size)
    q.accum = (long long *)take((size_t)c.ring * c.samples * 3 * sizeof(long long));
    q.general = (uint32_t *)take(c.general_cap * sizeof(uint32_t));
    q.general_cap = (uint32_t)c.general_cap;
    if (c.l0) {
        q.occ0 = (uint8_t *)take(c.samples * (size_t)c.nl);
        q.col0 = (float4 *)take(c.samples * (size_t)c.nl * sizeof(float4));
        q.l0_stride = (uint32_t)c.samples;
    }
    q.spill = spill;
    q.ray_cap = (uint32_t)c.ray_cap;
    q.shadow_cap = (uint32_t)c.sh_cap;
    q.split_cap = (uint32_t)c.split_cap;
    const int bits = levelBits(c.max_depth);
    q.level_shift = (uint32_t)(30 - bits);
    q.level_mask = (1u << bits) - 1u;
    q.sample_mask = (1u << q.level_shift) - 1u;
    q.tile_stride = 1;  // (set per frame: enqueueFrame)
    return off;
}

// slots per slab of the shadow queue (a power of two >= 64: rtShadeSlab) as every kernel of the frame sees it
void rtSetSlab(RtQueues &q, uint32_t slab) {
    uint32_t sh = 0;
    while ((1u << sh) < slab) sh++;
    q.sh_slab_shift = sh;
}

// A frame's primary rays: (origin, weight) is what k_gen_primary would store for every one of them (RtQueues::primA)
void rtImplyPrimaryOrigin(RtQueues &q, const RtFrameDev &f, float reach) {
    q.primA = make_float4(f.origin[0], f.origin[1], f.origin[2], 1.0f);
    q.prim_implied = 1u;
    q.prim_far = std::fmax(std::fmax(std::fabs(f.origin[0]), std::fabs(f.origin[1])), std::fabs(f.origin[2])) > 4.0f * reach ? 1u : 0u;
}

// How a frame of `n_local` tiles is cut into chunks and what its queues hold.
struct FramePlan {
    size_t tiles = 0, samples = 0, nchunks = 0;  // tiles / primary sample slots per chunk; chunks of the frame
    int lanes = 1;                               // independent pipelines (chunk c runs in lane c % lanes), each on a stream of its own
    int ring = 1;                                // chunks in flight per lane (min(chunks of a lane, depth + 1))
    bool optimistic = false;                     // queues sized from the fan-out observed on the scene, not for the worst case
    double ray_fan = 0.0, sh_fan = 0.0;
    QueueSizes z;
    CarveSpec spec;
    size_t bytes = 0;
};
struct PlanInput {
    size_t n_local = 0, per_tile = 0;
    int max_depth = 0, nl = 1;
    int64_t chunk_samples = 0;  // caller-given chunk size (taken as is, worst-case queues) or 0
    bool worst = false;         // size for the worst case even where a fan-out has been observed
    bool l0 = false;
    size_t split_cap = 0;
    double margin = 1.25;       // queue capacity over the fullest queue seen on the scene so far
    size_t budget = 0;          // bytes the workspace may take (ignored for a caller-given chunk size)
    int lanes = 1;              // pipelines to run side by side where the frame has at least that many chunks
};
// The fewest (hence largest, and equal) chunks whose workspace fits the budget.  Returns false if not even one tile per
// chunk fits.
bool planFrame(const rt_context *ctx, const RtLaunchCfg &cfg, const PlanInput &in, FramePlan &pl) {
    const int D = in.max_depth;
    const size_t ring_samples = (size_t)1 << (30 - levelBits(D));
    const bool optimistic = in.chunk_samples <= 0 && !in.worst && ctx->obs_valid && D <= ctx->obs_depth;
    auto eval = [&](size_t tiles, FramePlan &o) -> bool {
        o = FramePlan{};
        o.tiles = tiles;
        o.nchunks = (in.n_local + tiles - 1) / tiles;
        o.lanes = (int)std::min<size_t>((size_t)std::max(in.lanes, 1), o.nchunks);
        o.ring = (int)std::min<size_t>((o.nchunks + o.lanes - 1) / o.lanes, (size_t)D + 1);
        o.samples = tiles * in.per_tile;
        o.optimistic = optimistic;
        const double wf = worstFan(D, (size_t)o.ring);
        o.ray_fan = optimistic ? std::min(wf, std::max(1.0, in.margin * ctx->obs_ray)) : wf;
        o.sh_fan = optimistic ? std::min(wf * in.nl, std::max(1.0, in.margin * ctx->obs_sh)) : wf * in.nl;
        o.z = queueSizes(cfg, o.samples, o.ray_fan, o.sh_fan, in.nl);
        if (!o.z.fits32 || (size_t)o.ring * o.samples > ring_samples) return false;
        o.spec.samples = o.samples;
        o.spec.ring = o.ring;
        o.spec.ray_cap = o.z.ray_cap;
        o.spec.sh_cap = o.z.sh_cap;
        o.spec.general_cap = ctx->dev.has_general ? o.z.ray_cap : 64;  // (dev: what rt_commit put on THIS device, also in a replica of rt_multi)
        o.spec.split_cap = in.split_cap;
        o.spec.nl = in.nl;
        o.spec.l0 = in.l0;
        o.spec.max_depth = D;
        RtQueues dummy;
        o.bytes = (size_t)o.lanes * carveQueues(nullptr, o.spec, nullptr, dummy);
        return true;
    };
    if (in.n_local == 0) return eval(1, pl);
    if (in.chunk_samples > 0) {
        size_t tiles = std::min(std::max<size_t>(1, (size_t)in.chunk_samples / in.per_tile), in.n_local);
        const size_t nc = (in.n_local + tiles - 1) / tiles;
        tiles = (in.n_local + nc - 1) / nc;  // equal chunks: no small straggler at the end of the frame
        return eval(tiles, pl);
    }
    size_t prev = 0;
    for (size_t nc = 1; nc <= in.n_local; nc++) {
        size_t tiles = (in.n_local + nc - 1) / nc;
        if (tiles == prev) continue;
        prev = tiles;
        // (lanes that get the same number of chunks finish together: skip counts that would not divide, where that is possible)
        if (in.lanes > 1 && nc > 1 && ((in.n_local + tiles - 1) / tiles) % (size_t)in.lanes != 0 && tiles > 1) continue;
        if (eval(tiles, pl) && pl.bytes <= in.budget) return true;
        if (tiles == 1) break;
    }
    return false;
}

struct KernelTimer {  // optional per-kernel-class timing with HIP events on the launch stream
    bool on = false;
    hipStream_t stream = nullptr;
    struct Span { hipEvent_t a, b; int cls; };
    std::vector<Span> spans;
    ~KernelTimer() { discard(); }
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
    void collect(double *ms /*[7]*/) {
        for (Span &s : spans) {
            float t = 0;
            (void)hipEventElapsedTime(&t, s.a, s.b);
            ms[s.cls] += t;
        }
        discard();
    }
    void discard() {  // (the spans of an attempt that overflowed and is repeated do not count)
        for (Span &s : spans) {
            (void)hipEventDestroy(s.a);
            (void)hipEventDestroy(s.b);
        }
        spans.clear();
    }
};
struct EventPair {  // released on every exit path
    hipEvent_t a = nullptr, b = nullptr;
    int create() {
        HIPOK(hipEventCreate(&a));
        HIPOK(hipEventCreate(&b));
        return 0;
    }
    ~EventPair() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

// device counters of a finished frame / batch -> rt_stats (adds)
void addCounters(rt_stats *st, const uint32_t *c) {
    auto u64at = [&](int w) { return ((uint64_t)c[w + 1] << 32) | c[w]; };
    auto tot = [&](int k) { return u64at(RT_CNT_TOTALS + 2 * k); };
    st->rays_closest += tot(RT_TOT_RAYS) - tot(RT_TOT_INVALID);  // slots of pixels outside the image and unused slab slots are not rays
    st->rays_shadow += tot(RT_TOT_SHADOW) - tot(RT_TOT_SHADOW_HOLES);
    st->shadow_culled += tot(RT_TOT_CULLED);
    st->shadow_zero += tot(RT_TOT_ZERO);
    st->shadow_unoccluded += u64at(RT_CNT_UNOCCLUDED);
    st->split_items += tot(RT_TOT_SPLIT);
    st->hits_shaded += tot(RT_TOT_SHADED);
    st->queue_overflow += tot(RT_TOT_OVERFLOW);
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
    for (int k = 0; k < 8; k++) st->prof_packet_closest[k] += u64at(RT_CNT_PROF_PACKET + 2 * k);  // (prof_packet_any: no shadow packets)
}

// RT_TAILPROF=1 (diagnosis): the launches of a frame record when their waves start and end (RtQueues::tailprof)
struct TailProf {
    DevBuf<unsigned long long> buf;
    struct Rec { int cls; uint32_t waves; };
    std::vector<Rec> recs;
    bool on = false;
    uint32_t next(int cls, uint32_t waves) {
        recs.push_back(Rec{cls, waves});
        return (uint32_t)recs.size() - 1;
    }
};

// Enqueues a whole frame: no host synchronisation, no allocation.  Launch order of a pipeline (S = k_shade, A = shadow
// rays, C = bounce rays, P = primary packets, the index is the iteration = the ray queue):
//     ctl(0) gen(0) [C|P](0)   then per iteration j:   ctl(j) S(j) gen(j+1) [A(j)|C(j+1)|P(j+1)] splits(j) fold0(j) resolve(j-D)
// A(j), C(j+1) and P(j+1) only depend on S(j) / gen(j+1), so they share ONE persistent launch (k_trace) whose waves go
// from one to the next at their own pace.  `fuse` = false launches them one after the other (per-kernel timing, RT_FUSE=0).
// What is left is a chain -- S(j) needs the hits of k_trace(j-1), which needs the rays of S(j-1) -- and every link ends
// with its slowest wave (~0.3 ms of a mostly idle chip per launch).  So a frame of several chunks runs as pl.lanes
// INDEPENDENT pipelines (chunk c in lane c % lanes, each lane with queues and counters of its own: q0[lane]) on streams
// of their own: every kernel fetches its work dynamically, so the blocks of one lane's launch fill the CUs that the
// other lane's launch leaves idle at its end.  Lane 0 runs on cfg.stream, lane 1 on ctx->stream2, forked from and joined
// to cfg.stream by events.
int enqueueFrame(rt_context *ctx, const RtLaunchCfg &cfg0, const RtFrameDev &f, const FramePlan &pl, const RtQueues *q0, int max_depth,
                 size_t n_local, size_t per_tile, uint8_t *d_rgb8, float *d_rgbf, uint32_t *d_primid, int stats, bool fuse, KernelTimer &kt,
                 rt_stats *st, TailProf *tp = nullptr) {
    const int lanes = pl.lanes;
    auto probe = [&](RtQueues &q, int cls, uint32_t waves) {
        if (!tp || !tp->on) return;
        q.tailprof = tp->buf.p;
        q.launch_id = tp->next(cls, waves);
    };
    if (lanes > 2) return fail("enqueueFrame: at most two lanes");
    hipStream_t streams[2] = {cfg0.stream, ctx->stream2};
    // the workspace keeps the last frame's contents: counters, work cursors and totals start at zero; queues and sample
    // accumulators are written before they are read (level-0 k_shade stores every sample slot of a chunk)
    for (int l = 0; l < lanes; l++) {
        HIPOK(hipMemsetAsync(q0[l].counters, 0, RT_CNT_WORDS * sizeof(uint32_t), streams[0]));
        if (q0[l].occ0) HIPOK(hipMemsetAsync(q0[l].occ0, 0, pl.samples * (size_t)pl.spec.nl, streams[0]));  // (k_fold0 clears what it folds)
    }
    if (d_primid && n_local) HIPOK(hipMemsetAsync(d_primid, 0xFF, n_local * RT_TILE_PIXELS * sizeof(uint32_t), streams[0]));
    if (n_local == 0) return 0;
    if (lanes > 1) {
        HIPOK(hipEventRecord(ctx->ev_fork, streams[0]));
        HIPOK(hipStreamWaitEvent(streams[1], ctx->ev_fork, 0));
    }
    const size_t nchunks = pl.nchunks;
    const uint32_t root = ctx->bvh.root_ref;
    const int nlights = (int)ctx->lights.size();
    const uint64_t max_inputs = (uint64_t)std::ceil((double)pl.samples * pl.ray_fan);
    // chunk c = local tiles c, c + nchunks, c + 2 nchunks, ...: every chunk a sample of the whole image (rt_kernels.hip)
    auto ntilesOf = [&](size_t c) { return (uint32_t)((n_local - c + nchunks - 1) / nchunks); };
    // lane-local view: iteration j of lane l brings in chunk c = l + j * lanes
    struct Lane { size_t nk, T; RtLaunchCfg cfg; };
    Lane L[2];
    size_t Tmax = 0;
    for (int l = 0; l < lanes; l++) {
        L[l].nk = (nchunks - (size_t)l + (size_t)lanes - 1) / (size_t)lanes;
        L[l].T = L[l].nk + (size_t)max_depth;
        L[l].cfg = cfg0;
        L[l].cfg.stream = streams[l];
        Tmax = std::max(Tmax, L[l].T);
    }
    auto chunkOf = [&](int l, size_t j) { return (size_t)l + j * (size_t)lanes; };
    auto primN = [&](int l, size_t j) { return j < L[l].nk ? ntilesOf(chunkOf(l, j)) * (uint32_t)per_tile : 0u; };
    auto sample0 = [&](size_t j) { return (uint32_t)((j % (size_t)pl.ring) * pl.samples); };
    // queue j's primary rays (if a chunk enters at j) and closest hits: everything k_shade(j) will read
    auto enter = [&](int l, size_t j, int par_any) {  // par_any: the shadow queue walked next to it (-1: none)
        const RtLaunchCfg &cfg = L[l].cfg;
        kt.stream = cfg.stream;
        RtQueues q = q0[l];
        q.tile_stride = (uint32_t)nchunks;
        q.prim_n = primN(l, j);
        q.prim_sample0 = j < L[l].nk ? sample0(j) : 0u;
        q.l0_sample0 = j >= 1 && j - 1 < L[l].nk ? sample0(j - 1) : 0u;  // (the shadow rays are those of iteration j - 1)
        const int par = (int)(j & 1);
        if (q.prim_n) {
            kt.begin(0);
            rtLaunchGenPrimary(cfg, f, q, par, (uint32_t)chunkOf(l, j), ntilesOf(chunkOf(l, j)));
            kt.end();
        }
        const bool pk = cfg.packets && q.prim_n > 0, queue_j = j < L[l].T;
        const uint32_t first = pk ? q.prim_n : 0u;
        if (fuse) {
            if (par_any < 0 && !queue_j) return;
            kt.begin(6);
            probe(q, 6, (uint32_t)(cfg.num_cus * cfg.trace_blocks_per_cu * 4));
            rtLaunchTrace(cfg, ctx->dev, root, q, par_any, queue_j ? par : -1, first, pk, stats);
            kt.end();
        } else {
            if (par_any >= 0) {
                kt.begin(3);
                rtLaunchAny(cfg, ctx->dev, root, q, par_any, stats);
                kt.end();
            }
            if (queue_j && (j > 0 || !pk)) {
                kt.begin(1);
                rtLaunchClosest(cfg, ctx->dev, root, q, par, first, stats);
                kt.end();
            }
            if (pk) {
                kt.begin(5);
                rtLaunchPacket(cfg, ctx->dev, root, q, par, stats);
                kt.end();
            }
        }
        if (st) {  // (launches that walk a queue; a fused launch walks one of each kind)
            if (queue_j) st->launches_closest++;
            if (par_any >= 0) st->launches_any++;
        }
    };
    for (int l = 0; l < lanes; l++) {
        rtLaunchIterCtl(L[l].cfg, q0[l], 1, primN(l, 1), true, primN(l, 0));
        enter(l, 0, -1);
    }
    for (size_t j = 0; j < Tmax; j++) {
        for (int l = 0; l < lanes; l++) {  // (the lanes' launches are enqueued alternately: both streams stay fed)
            if (j >= L[l].T) continue;
            const RtLaunchCfg &cfg = L[l].cfg;
            const int par = (int)(j & 1);
            RtQueues q = q0[l];
        q.tile_stride = (uint32_t)nchunks;
            q.prim_n = primN(l, j);
            q.prim_sample0 = j < L[l].nk ? sample0(j) : 0u;
            q.l0_sample0 = q.prim_sample0;
            if (j > 0) rtLaunchIterCtl(cfg, q, par ^ 1, primN(l, j + 1), false, 0u);
            kt.stream = cfg.stream;
            kt.begin(2);
            probe(q, 2, (uint32_t)rtShadeWaves(cfg));
            rtLaunchShade(cfg, ctx->dev, q, par, max_depth, f.spp, (uint32_t)chunkOf(l, j), d_primid, stats >= 2, max_inputs, 1);
            probe(q, 7, (uint32_t)rtShadeWaves(cfg));
            rtLaunchShade(cfg, ctx->dev, q, par, max_depth, f.spp, (uint32_t)chunkOf(l, j), d_primid, stats >= 2, max_inputs, 2);
            kt.end();
            enter(l, j + 1, par);
            kt.stream = cfg.stream;
            kt.begin(3);
            rtLaunchSplitRounds(cfg, ctx->dev, root, q, par, stats);
            if (q.occ0 && q.prim_n) rtLaunchFold0(cfg, q, nlights);
            kt.end();
            if (j >= (size_t)max_depth) {  // the chunk whose deepest level was just shaded and whose last shadow rays were traced
                const size_t jc = j - (size_t)max_depth, c = chunkOf(l, jc);
                kt.begin(4);
                rtLaunchResolve(cfg, f, q, sample0(jc), (uint32_t)c, ntilesOf(c), d_rgb8, d_rgbf);
                kt.end();
            }
        }
    }
    for (int l = 0; l < lanes; l++) rtLaunchIterCtl(L[l].cfg, q0[l], (int)((L[l].T - 1) & 1), 0u, false, 0u);  // the last iteration's counters -> totals
    if (lanes > 1) {
        HIPOK(hipEventRecord(ctx->ev_join, streams[1]));
        HIPOK(hipStreamWaitEvent(streams[0], ctx->ev_join, 0));
    }
    kt.stream = streams[0];
    return 0;
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

// the arena holds at least `bytes` (growing frees first: the old contents are of no use to anybody)
int ensureArena(rt_context *ctx, size_t bytes) {
    if (ctx->d_arena.n >= bytes && ctx->d_arena.p) return 0;
    const double t0 = nowMs();
    if (ctx->d_arena.alloc(bytes)) return fail("workspace allocation failed (workspace released): " + g_err);
    if (getenv("RT_VERBOSE")) std::fprintf(stderr, "[rt] workspace: %.2f GB in %.0f ms\n", (double)bytes / 1e9, nowMs() - t0);
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
    if (per_tile > ((size_t)1 << (30 - levelBits(p->max_depth)))) return fail("spp too large for one tile per chunk");
    const int nl = std::max<int>(1, (int)ctx->lights.size());
    const size_t nlights = ctx->lights.size();

    RtLaunchCfg cfg = ctx->cfg;
    cfg.stream = stream;
    const int stats_on = (p->collect_stats & 4) ? 2 : (p->collect_stats & 1) ? 1 : 0;  // counters / + wave-time profile
    KernelTimer kt;
    kt.on = (p->collect_stats & 2) != 0;
    kt.stream = stream;
    rt_stats local{};
    rt_stats *st = stats ? stats : &local;
    std::memset(st, 0, sizeof *st);
    EventPair ev;
    if (ev.create()) return 1;

    // Planning.  The workspace is one arena of at most RT_WORKSPACE_GB (30 GiB; never more than 7/8 of what is free on the
    // device, RT_WORKSPACE_FRACTION): the frame is cut into the fewest equal chunks whose queues fit it.  Queue capacities
    // per sample: the WORST case (every hit spawns a pair, every light faces every hit) for the first frame after
    // rt_commit -- 5-20 x what scenes produce, so that frame takes more, smaller chunks, which the pipeline makes cheap --
    // afterwards 1.25 x the fan-out observed on the scene.  A frame that overflows its queues after all (the kernels drop and
    // count what does not fit; nothing is written out of bounds) is rendered again with worst-case queues: every output
    // pixel is written by the repeat.  A caller-given chunk size is taken as is, with worst-case queues.
    int l0_mode = -1;  // level-0 shadow results stored instead of added: -1 = where it pays (below), 0 = never, 1 = always
    if (const char *e = getenv("RT_L0_STORE")) l0_mode = atoi(e) != 0 ? 1 : 0;
    PlanInput in;
    in.n_local = (size_t)n_local;
    in.per_tile = per_tile;
    in.max_depth = p->max_depth;
    in.nl = nl;
    in.chunk_samples = p->chunk_samples;
    in.split_cap = (size_t)1 << 22;  // items per round; a wave that finds no room keeps walking
    if (const char *e = getenv("RT_SPLIT_CAP")) in.split_cap = (size_t)std::max(64, atoi(e));  // (test knob: forces the no-room path)
    if (const char *e = getenv("RT_CHUNK_MARGIN")) in.margin = atof(e);  // (test knob: < 1 forces the overflow-and-repeat path)
    double ws_gb = 30.0, ws_frac = 0.875;
    if (const char *e = getenv("RT_WORKSPACE_GB")) ws_gb = std::max(0.001, atof(e));
    if (const char *e = getenv("RT_WORKSPACE_FRACTION")) ws_frac = std::min(0.875, std::max(0.01, atof(e)));
    bool fuse = true;  // RT_FUSE=0: the traversal phases of an iteration as launches of their own (ms_any / ms_closest / ms_packet)
    if (const char *e = getenv("RT_FUSE")) fuse = atoi(e) != 0;
    // RT_LANES=2: two pipelines side by side.  Measured on the 4K competition frame: the second lane's launches do fill the
    // ends of the first one's (one lane with chunks of the same size is 10 % slower), but two lanes halve the chunk a workspace
    // can hold, which costs what the overlap gains: 446 vs 435 ms in 30 GB, 463 vs 452 in 16 GB, 500 vs 500 in 8 GB.
    // Two lanes where the tree is large (>= 8 Mi triangles): its walks leave L2 and run to hundreds of steps, a launch's tail
    // is then milliseconds of a mostly idle chip, and the other lane's launches are what fills it -- the 50 M-triangle frame
    // at 64 spp: 1089 -> 1019 ms in the same 30 GB (profiles/r03_lanes.md).
    in.lanes = ctx->info.n_triangles >= ((uint64_t)8 << 20) ? 2 : 1;
    if (const char *e = getenv("RT_LANES")) in.lanes = std::max(1, std::min(2, atoi(e)));
    std::vector<uint32_t> hc(2 * RT_CNT_WORDS, 0);
    bool started = false;  // ev.a is recorded once the workspace exists: ms_total is GPU time, not hipMalloc time
    for (int attempt = 0;; attempt++) {
        // Level-0 shadow results stored instead of added (RtQueues::occ0) where most of the scene's shadow rays are
        // level-0 rays: their three 64-bit atomics each are then what bounds k_any (bunny scene: 88 -> 27 ms), while a
        // frame whose shadow rays mostly come from deeper levels hides its atomics behind the walks and would only pay for
        // the extra pass.  Either way the sums are the same integers.  RT_L0_STORE=0 / 1 forces the choice.
        in.l0 = nlights > 0 && (l0_mode == 1 || (l0_mode < 0 && ctx->obs_valid && ctx->obs_l0_share >= 0.6));
        in.worst = attempt > 0;
        cfg.shade_slab = ctx->cfg.shade_slab;
        size_t free_b = 0, total_b = 0;
        HIPOK(hipMemGetInfo(&free_b, &total_b));
        const size_t cap = (size_t)((double)(free_b + ctx->d_arena.n) * ws_frac);  // (other users of the HBM: scene, outputs, torch, RCCL)
        size_t budget = std::min((size_t)(ws_gb * (double)(1ull << 30)), cap);
        if ((p->flags & RT_RENDER_SINGLE_FRAME) && ctx->d_arena.n) budget = ctx->d_arena.n;  // one image: nothing grows for it
        budget = std::max(budget, ctx->d_arena.n);
        in.budget = budget;
        FramePlan pl;
        if (!planFrame(ctx, cfg, in, pl)) {
            in.budget = cap;  // a deep bounce budget at many spp: one tile needs more than the default workspace
            if (in.chunk_samples > 0 || !planFrame(ctx, cfg, in, pl))
                return fail("the wavefront queues of one chunk do not fit 32-bit indices / the device memory (lower spp, max_depth or chunk_samples)");
        }
        // what to allocate: the whole budget once a frame needs several chunks (every later frame is then carved from the
        // same arena), what this frame needs (+ 1/8) otherwise
        size_t want = pl.bytes;
        if (want > ctx->d_arena.n) {
            if (in.chunk_samples <= 0 && pl.nchunks > 1) want = std::max(want, in.budget);
            else want = std::min(std::max(cap, want), want + want / 8);
        }
        if (ensureArena(ctx, want)) return 1;
        RtQueues q[2];
        for (int l = 0; l < pl.lanes; l++) {  // (every lane its own queues, counters and traversal spill area)
            carveQueues(ctx->d_arena.p + (size_t)l * (pl.bytes / (size_t)pl.lanes), pl.spec, ctx->d_spill.p + (size_t)l * ctx->spill_words, q[l]);
            rtImplyPrimaryOrigin(q[l], f, ctx->dev.reach);
            rtSetSlab(q[l], pl.z.slab);
        }
        cfg.shade_slab = pl.z.slab;
        if (getenv("RT_VERBOSE"))
            std::fprintf(stderr, "[rt] frame: %zu chunks of %zu samples in %d lane(s) (%d in flight per lane), %s queues %.2f rays + %.2f shadow rays per sample, %.2f GB%s\n",
                         pl.nchunks, pl.samples, pl.lanes, pl.ring, pl.optimistic ? "observed" : "worst-case", pl.ray_fan, pl.sh_fan,
                         (double)pl.bytes / 1e9, in.l0 ? ", level-0 shadow results stored" : "");
        if (!started) {
            HIPOK(hipEventRecord(ev.a, stream));
            started = true;
        }
        rt_stats launches{};
        TailProf tp;
        tp.on = getenv("RT_TAILPROF") != nullptr;
        if (tp.on) {
            const size_t slots = (size_t)RT_TAIL_WAVES * 4 * (3 * (pl.nchunks + 16) + 8);
            if (tp.buf.alloc(slots)) return 1;
            HIPOK(hipMemset(tp.buf.p, 0, slots * sizeof(unsigned long long)));
        }
        if (enqueueFrame(ctx, cfg, f, pl, q, p->max_depth, (size_t)n_local, per_tile, d_rgb8, d_rgbf, d_primid, stats_on, fuse, kt, &launches, &tp)) return 1;
        for (int l = 0; l < pl.lanes; l++)
            HIPOK(hipMemcpyAsync(hc.data() + (size_t)l * RT_CNT_WORDS, q[l].counters, RT_CNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIPOK(hipEventRecord(ev.b, stream));
        HIPOK(hipStreamSynchronize(stream));
        HIPOK(hipGetLastError());
        if (tp.on) {  // per launch: when the first / mean / last wave left each phase and ended, in us after the first wave's start
            std::vector<unsigned long long> h((size_t)RT_TAIL_WAVES * 4 * tp.recs.size());
            HIPOK(hipMemcpy(h.data(), tp.buf.p, h.size() * 8, hipMemcpyDeviceToHost));
            struct Acc { double dur = 0, idle = 0, first_end = 0; int n = 0; } acc[8];
            for (size_t k = 0; k < tp.recs.size(); k++) {
                const unsigned long long *r = h.data() + (size_t)RT_TAIL_WAVES * 4 * k;
                unsigned long long t0 = ~0ull;
                size_t waves = 0;
                for (size_t w = 0; w < RT_TAIL_WAVES; w++)
                    if (r[4 * w + 3]) {
                        t0 = std::min(t0, r[4 * w]);
                        waves++;
                    }
                if (!waves) continue;
                double mn[4], mx[4], sum[4];
                size_t cnt[4];
                for (int c = 0; c < 4; c++) { mn[c] = 1e30; mx[c] = 0; sum[c] = 0; cnt[c] = 0; }
                for (size_t w = 0; w < RT_TAIL_WAVES; w++) {
                    if (!r[4 * w + 3]) continue;
                    for (int c = 0; c < 4; c++) {
                        if (!r[4 * w + c]) continue;
                        const double t = (double)(r[4 * w + c] - t0) / 100.0;
                        mn[c] = std::min(mn[c], t); mx[c] = std::max(mx[c], t); sum[c] += t; cnt[c]++;
                    }
                }
                const int cls = tp.recs[k].cls;
                acc[cls].dur += mx[3];
                acc[cls].idle += mx[3] - sum[3] / (double)waves + sum[0] / (double)waves;
                acc[cls].first_end += mx[3] - mn[3];
                acc[cls].n++;
                if (getenv("RT_TAILPROF_LAUNCHES")) {
                    std::fprintf(stderr, "[rt]   launch %zu (class %d, %zu waves): last start %.0f us", k, cls, waves, mx[0]);
                    if (cnt[1]) std::fprintf(stderr, "; shadow phase left at %.0f / %.0f / %.0f", mn[1], sum[1] / cnt[1], mx[1]);
                    if (cnt[2]) std::fprintf(stderr, "; bounce phase left at %.0f / %.0f / %.0f", mn[2], sum[2] / cnt[2], mx[2]);
                    std::fprintf(stderr, "; ends at %.0f / %.0f / %.0f\n", mn[3], sum[3] / cnt[3], mx[3]);
                }
            }
            for (int c = 0; c < 8; c++)
                if (acc[c].n)
                    std::fprintf(stderr, "[rt] tailprof class %d: %d launches, mean duration %.0f us, first wave done %.0f us before the last, "
                                 "mean wave absent (late start + early end) for %.0f us of a launch\n",
                                 c, acc[c].n, acc[c].dur / acc[c].n, acc[c].first_end / acc[c].n, acc[c].idle / acc[c].n);
            tp.buf.release();
        }
        st->chunks += pl.nchunks;
        uint64_t tot[RT_TOT_WORDS];  // the lanes' totals together (sums; the fullest queue of either)
        for (int k = 0; k < RT_TOT_WORDS; k++) {
            tot[k] = 0;
            for (int l = 0; l < pl.lanes; l++) {
                const uint64_t v = reinterpret_cast<const uint64_t *>(hc.data() + (size_t)l * RT_CNT_WORDS + RT_CNT_TOTALS)[k];
                tot[k] = (k == RT_TOT_MAX_RAYS || k == RT_TOT_MAX_SHADOW) ? std::max(tot[k], v) : tot[k] + v;
            }
        }
        if (tot[RT_TOT_OVERFLOW] && pl.optimistic) {  // the scene out-grew what was seen so far: again, sized for the worst case
            st->chunk_retries += pl.nchunks;
            ctx->obs_valid = false;
            kt.discard();
            continue;
        }
        st->launches_closest = launches.launches_closest;
        st->launches_any = launches.launches_any;
        for (int l = 0; l < pl.lanes; l++) addCounters(st, hc.data() + (size_t)l * RT_CNT_WORDS);
        if (!tot[RT_TOT_OVERFLOW] && n_local > 0) {
            // Fan-out for the next frames: the frame's mean (every level of a sample is in flight at once in steady state) or
            // the fullest queue seen, whichever is larger -- without the slab slack that queueSizes adds back.
            const double frame_samples = (double)n_local * (double)per_tile, chunk = (double)pl.samples;
            const size_t waves = (size_t)rtShadeWaves(cfg);
            const double slack_r = 2.0 * (double)waves * (double)std::max<uint32_t>(pl.z.slab, 128u), slack_s = 2.0 * (double)waves * nl * (double)pl.z.slab;
            const double r = std::max((double)(tot[RT_TOT_RAYS] - tot[RT_TOT_INVALID]) / frame_samples, ((double)tot[RT_TOT_MAX_RAYS] - slack_r) / chunk);
            const double sh = std::max((double)(tot[RT_TOT_SHADOW] - tot[RT_TOT_SHADOW_HOLES]) / frame_samples, ((double)tot[RT_TOT_MAX_SHADOW] - slack_s) / chunk);
            const double sall = (double)(tot[RT_TOT_SHADOW] - tot[RT_TOT_SHADOW_HOLES]);
            ctx->obs_l0_share = sall > 0 ? (double)tot[RT_TOT_SHADOW_L0] / sall : 0.0;
            if (getenv("RT_VERBOSE"))
                std::fprintf(stderr, "[rt] frame done: %.2f rays + %.2f shadow rays per sample in flight, level-0 share of the shadow rays %.2f; "
                             "queue slots without a ray: %.1f %% of the ray queues, %.1f %% of the shadow queues\n", r, sh, ctx->obs_l0_share,
                             100.0 * (double)tot[RT_TOT_INVALID] / std::max(1.0, (double)tot[RT_TOT_RAYS]),
                             100.0 * (double)tot[RT_TOT_SHADOW_HOLES] / std::max(1.0, (double)tot[RT_TOT_SHADOW]));
            if (!ctx->obs_valid) ctx->obs_ray = ctx->obs_sh = 0.0;
            ctx->obs_ray = std::max(ctx->obs_ray, std::max(r, 1.0));
            ctx->obs_sh = std::max(ctx->obs_sh, sh);
            ctx->obs_depth = ctx->obs_valid ? std::max(ctx->obs_depth, p->max_depth) : p->max_depth;
            ctx->obs_valid = true;
        }
        break;
    }
    float ms = 0;
    HIPOK(hipEventElapsedTime(&ms, ev.a, ev.b));
    st->ms_total = ms;
    double cls[7] = {0, 0, 0, 0, 0, 0, 0};
    kt.collect(cls);
    st->ms_gen = cls[0]; st->ms_closest = cls[1]; st->ms_shade = cls[2]; st->ms_any = cls[3]; st->ms_resolve = cls[4];
    st->ms_packet = cls[5];
    st->ms_trace = cls[6];
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

// ---- rt_render_tiles_static: the same pipeline on the library's static scratch, allocating nothing -----------------
#define RT_STATIC_SHADE_BLOCKS 64   // k_shade grid of the static path: 256 waves, each owning (lights + 1) output slabs
#define RT_STATIC_SLAB 64u          // slots per shadow slab (ray slabs: 128)
#define RT_STATIC_SPLIT_CAP 65536u  // work items of split any-hit walks per round
int renderTilesStatic(rt_context *ctx, const rt_camera *cam, const rt_render_params *p, uint8_t *d_rgb8, float *d_rgbf,
                      uint32_t *d_primid, hipStream_t stream) {
    RtFrameDev f{};
    int world = 1;
    if (makeFrame(ctx, cam, p, f, world)) return 1;
    if (bind(ctx)) return 1;
    if (!ctx->scratch) return fail("rt_render_tiles_static: the static scratch was not resolved on this device");
    int64_t n_local = 0;
    rt_tile_counts(p->width, p->height, p->tile_rank, world, &n_local, nullptr);
    RtLaunchCfg cfg = ctx->cfg;
    cfg.stream = stream;
    cfg.shade_blocks = RT_STATIC_SHADE_BLOCKS;
    cfg.shade_slab = RT_STATIC_SLAB;
    PlanInput in;
    in.n_local = (size_t)n_local;
    in.per_tile = (size_t)RT_TILE_PIXELS * p->spp;
    in.max_depth = p->max_depth;
    in.nl = std::max<int>(1, (int)ctx->lights.size());
    in.worst = true;  // (no read-back, hence no repeat: the queues must hold the worst case)
    in.split_cap = RT_STATIC_SPLIT_CAP;
    in.budget = ctx->scratch_bytes;
    FramePlan pl;
    if (!planFrame(ctx, cfg, in, pl))
        return fail("rt_render_tiles_static: one 16x16 tile at " + std::to_string(p->spp) + " spp, depth " + std::to_string(p->max_depth) +
                    " with " + std::to_string(in.nl) + " lights does not fit the " + std::to_string(ctx->scratch_bytes >> 20) +
                    " MiB static scratch (lower spp or max_depth, or use rt_render_tiles_device)");
    RtQueues q;
    carveQueues((unsigned char *)ctx->scratch, pl.spec, ctx->d_spill.p, q);
    rtImplyPrimaryOrigin(q, f, ctx->dev.reach);
    rtSetSlab(q, pl.z.slab);
    KernelTimer kt;  // (off)
    if (enqueueFrame(ctx, cfg, f, pl, &q, p->max_depth, (size_t)n_local, in.per_tile, d_rgb8, d_rgbf, d_primid, 0, true, kt, nullptr)) return 1;
    HIPOK(hipGetLastError());
    return 0;
}

// ---- rt_trace_batch: trace_ray (main.cpp:221-233) for arbitrary rays -- a batch is a "chunk" of n one-sample pixels ----
// count: the statistics build of the traversal kernels (records fetched, longest walk -> rt_stats), as rt_render's collect_stats & 1
int traceBatch(rt_context *ctx, int64_t n, const float *rays, int32_t depth, float *out_color, rt_hit *out_hits, uint8_t *out_shadow,
               rt_stats *st, bool count) {
    const int sl = count ? 1 : 0;
    const int nl = std::max<int>(1, (int)ctx->lights.size());
    RtLaunchCfg cfg = ctx->cfg;
    cfg.stream = ctx->stream;
    // rays per pass: 1 Mi, halved until the worst case of the deepest level (2^depth rays per input ray) fits 32-bit
    // queue indices and the free memory
    size_t chunk = std::min<size_t>(std::max<int64_t>(n, 1), (size_t)1 << 20);
    size_t free_b = 0, total_b = 0;
    HIPOK(hipMemGetInfo(&free_b, &total_b));
    const size_t budget = (free_b + ctx->d_arena.n) / 8 * 7;
    PlanInput in;
    in.per_tile = 1;  // (a "tile" of one ray)
    in.max_depth = depth;
    in.nl = nl;
    in.worst = true;
    in.split_cap = (size_t)1 << 22;
    if (const char *e = getenv("RT_SPLIT_CAP")) in.split_cap = (size_t)std::max(64, atoi(e));
    FramePlan pl;
    for (;; chunk /= 2) {
        in.n_local = chunk;
        in.chunk_samples = (int64_t)chunk;  // one chunk: the levels run one after the other
        if (planFrame(ctx, cfg, in, pl) && pl.bytes <= budget) break;
        if (chunk <= 1) return fail("rt_trace_batch: the queues of a single ray at this depth do not fit the device memory");
    }
    if (ensureArena(ctx, pl.bytes)) return 1;
    RtQueues q0;
    carveQueues(ctx->d_arena.p, pl.spec, ctx->d_spill.p, q0);
    cfg.shade_slab = pl.z.slab;
    rtSetSlab(q0, pl.z.slab);
    q0.prim_far = 1u;  // (rays of unknown origin: always the reference's arithmetic)
    DevBuf<float> d_rays, d_col;
    DevBuf<rt_hit> d_hits;
    DevBuf<uint8_t> d_sh;
    int rc = 0;
    std::vector<uint32_t> hc(RT_CNT_WORDS);
    for (int64_t off = 0; off < n && !rc; off += (int64_t)chunk) {
        const uint32_t m = (uint32_t)std::min<int64_t>(chunk, n - off);
        rc = d_rays.alloc((size_t)m * 6) || d_col.alloc((size_t)m * 3) || d_hits.alloc(m) || d_sh.alloc((size_t)m * 4);
        if (rc) break;
        hipError_t e = hipMemcpyAsync(d_rays.p, rays + 6 * off, (size_t)m * 6 * sizeof(float), hipMemcpyHostToDevice, cfg.stream);
        if (e == hipSuccess) e = hipMemsetAsync(q0.counters, 0, RT_CNT_WORDS * sizeof(uint32_t), cfg.stream);
        if (e != hipSuccess) { rc = fail(hipGetErrorString(e)); break; }
        for (int j = 0; j <= depth; j++) {  // one chunk: the levels one after the other, every phase a launch of its own
            const int par = j & 1;
            RtQueues q = q0;
            q.prim_n = j == 0 ? m : 0u;
            q.prim_sample0 = 0;
            q.l0_sample0 = 0;
            rtLaunchIterCtl(cfg, q, par ^ 1, 0u, j == 0, q.prim_n);
            if (j == 0) {
                rtLaunchRaysFromHost(cfg, d_rays.p, m, q);
                if (cfg.packets) rtLaunchPacket(cfg, ctx->dev, ctx->bvh.root_ref, q, 0, sl);
                else rtLaunchClosest(cfg, ctx->dev, ctx->bvh.root_ref, q, 0, 0u, sl);
                // the closest hits of the batch are reported from this pass
                if (out_hits || out_shadow) rtLaunchHitInfo(cfg, ctx->dev, ctx->bvh.root_ref, q, m, out_hits ? d_hits.p : nullptr, out_shadow ? d_sh.p : nullptr);
            } else {
                rtLaunchClosest(cfg, ctx->dev, ctx->bvh.root_ref, q, par, 0u, sl);
            }
            rtLaunchShade(cfg, ctx->dev, q, par, depth, 1, 0, nullptr, false, (uint64_t)m << j);
            rtLaunchAny(cfg, ctx->dev, ctx->bvh.root_ref, q, par, sl);
            rtLaunchSplitRounds(cfg, ctx->dev, ctx->bvh.root_ref, q, par, sl);
            st->launches_closest++;
            st->launches_any++;
        }
        rtLaunchIterCtl(cfg, q0, depth & 1, 0u, false, 0u);
        rtLaunchBatchColors(cfg, q0, m, d_col.p);
        e = hipStreamSynchronize(cfg.stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(hc.data(), q0.counters, RT_CNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_color) e = hipMemcpy(out_color + 3 * off, d_col.p, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_hits) e = hipMemcpy(out_hits + off, d_hits.p, (size_t)m * sizeof(rt_hit), hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_shadow) e = hipMemcpy(out_shadow + 4 * off, d_sh.p, (size_t)m * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = fail(hipGetErrorString(e)); break; }
        addCounters(st, hc.data());
        if (out_hits) {  // slot -> triangle index inside its mesh
            for (uint32_t k = 0; k < m; k++) {
                rt_hit &h = out_hits[off + k];
                if (h.hit && h.triangle >= 0) h.triangle = (int32_t)((int64_t)h.prim - ctx->mesh_first[h.object]);
            }
        }
    }
    d_rays.release(); d_col.release(); d_hits.release(); d_sh.release();
    return rc;
}
