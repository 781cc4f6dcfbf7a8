// rt_kernels.hip -- the wavefront pipeline: hand-written HIP kernels for gfx950 (wave64).
//
//   k_gen_primary   SSAA primary rays of a chunk of tiles           (reference main.cpp:620-642)
//   k_closest       closest hit: analytic prims + linear BVH        (closest main.cpp:59-71 and every intersect())
//   k_shade         PhongModel minus its nested calls; emits shadow rays with their pending radiance and the
//                   reflection / refraction pair of the next level  (main.cpp:81-213)
//   k_any           shadow rays: early-out any-hit, adds the pending radiance when unoccluded (is_shadowed :41-57)
//   k_resolve       per-sample clamp, mean, tone map, 8-bit truncation (main.cpp:232,646,240-256; Image.h:91-95)
//   k_assemble      tile-major rank buffers -> row-major image (multi-GPU gather epilogue)
//   k_hitinfo / k_eval_texture / k_eval_perlin   test hooks of the C ABI
//
// Queues are compacted with 64-wide ballots (mbcnt gives the lane offset inside the wave); k_shade's waves reserve
// their output slots in slabs (one atomic per RT_SLAB_MAX slots), the unused remainder of a slab is flagged as holes,
// and they shade hits only (compacted through an LDS ring).  The traversal kernels are persistent: lanes fetch rays
// dynamically (8 cursors per queue), keep a per-lane stack in LDS laid out [entry][lane] (conflict-free push/pop),
// and export shadow rays whose walks get long as work items for follow-up rounds ("Long any-hit walks").
#include <hip/hip_runtime.h>
#include "rt_device.h"
#include "rt_kernels.h"


// ---------------------------------------------------------------------------------------------------------------
// wave-level helpers
RT_DEV uint32_t laneId() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
RT_DEV uint32_t maskRankBelow(unsigned long long m) {  // number of set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// all 64 lanes must call this together; `total` is wave-uniform.  Returns the wave's base slot (uniform).
RT_DEV uint32_t waveReserve(uint32_t *counter, uint32_t total) {
    uint32_t base = 0;
    if (laneId() == 0 && total) base = atomicAdd(counter, total);
    return __builtin_amdgcn_readfirstlane(base);
}

// RT_TAILPROF: when the waves of a launch start, leave a phase and end (RtQueues::tailprof).  Every wave writes its own four
// words [t0, left phase 0, left phase 1, t1] (100 MHz ticks) -- nothing shared: a probe that folded its times into one
// line per launch with atomics queued 6 x 8192 of them on one address at the end of a launch, and measured mostly itself.
RT_DEV unsigned long long *rtTailSlot(const RtQueues &q) {
    return q.tailprof + ((size_t)q.launch_id * RT_TAIL_WAVES + (size_t)blockIdx.x * (RT_BLOCK / 64) + (threadIdx.x >> 6)) * 4;
}
RT_DEV unsigned long long rtTailBegin(const RtQueues &q) { return q.tailprof ? wall_clock64() : 0ull; }
RT_DEV void rtTailMark(const RtQueues &q, int k) {  // k: 0 = the shadow rays, 1 = the bounce rays of the fused traversal launch
    if (q.tailprof && laneId() == 0 && blockIdx.x * (RT_BLOCK / 64) < RT_TAIL_WAVES) rtTailSlot(q)[1 + k] = wall_clock64();
}
RT_DEV void rtTailEnd(const RtQueues &q, unsigned long long t0) {
    if (q.tailprof && laneId() == 0 && blockIdx.x * (RT_BLOCK / 64) < RT_TAIL_WAVES) {
        unsigned long long *p = rtTailSlot(q);
        p[0] = t0;
        p[3] = wall_clock64();
    }
}

// Adds (x, y, z) to the fixed-point accumulator of `sample` for every lane with `on`.  What these atomics cost is the
// L2 atomic units' time per lane operation, and the rays of one sample (a reflection/refraction pair, their children,
// the shadow rays those cast towards one light) sit in neighbouring queue slots, hence mostly in neighbouring lanes:
// runs of equal samples inside a group of 8 lanes are summed first -- in fixed point, so the accumulator ends up with
// the same bits as with separate atomics -- and only the first lane of a run issues the atomics.
// All 64 lanes must call this together.
RT_DEV void accumulateRuns(long long *accum, bool on, uint32_t sample, V3 c) {
    const uint32_t lane = laneId();
    long long fx = on ? rtToFixed(c.x) : 0, fy = on ? rtToFixed(c.y) : 0, fz = on ? rtToFixed(c.z) : 0;
    const uint32_t key = on ? sample : (RT_SAMPLE_MASK + 1u + lane);  // never equal to a neighbour's
    const uint32_t pkey = __shfl_up(key, 1);
    const bool head = (lane & 7u) == 0 || pkey != key;  // first lane of a run
    const unsigned long long heads = __ballot(head);
    for (int d = 1; d < 8; d <<= 1) {
        const long long ox = __shfl_down(fx, d), oy = __shfl_down(fy, d), oz = __shfl_down(fz, d);
        // lane + d belongs to this lane's run iff no run starts in (lane, lane + d]
        if ((lane & 7u) + d < 8u && ((heads >> (lane + 1u)) & ((1ull << d) - 1ull)) == 0) {
            fx += ox;
            fy += oy;
            fz += oz;
        }
    }
    if (on && head) {
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(accum) + 3ull * sample;
        atomicAdd(acc + 0, (unsigned long long)fx);
        atomicAdd(acc + 1, (unsigned long long)fy);
        atomicAdd(acc + 2, (unsigned long long)fz);
    }
}

// Work cursors of the kernels that deal out a queue dynamically (k_shade's segments, the packet batches; the per-lane
// walks have the same scheme inline).  The queue [first, n) is cut into RT_WORK_PARTS contiguous parts with a cursor each,
// on its own 128-byte line: a returning atomic on ONE address retires every ~10-15 ns chip-wide, which a launch of a few
// million entries fetched 64-1024 at a time reaches.  A wave starts in the part of its XCD (workgroups go round-robin over
// the 8 XCDs) and moves on when a part is used up.  All 64 lanes call these together; everything is wave-uniform.
struct RtPartCursor {
    uint32_t *cursors;
    uint32_t first, n, part_len, part, parts_done;
    uint32_t seen;  // how far the current part had got at this wave's last fetch
};
RT_DEV RtPartCursor rtPartCursor(uint32_t *cursors, uint32_t first, uint32_t n, uint32_t gran) {
    RtPartCursor c;
    c.cursors = cursors;
    c.first = first;
    c.n = n;
    const uint32_t span = n > first ? n - first : 0u;
    c.part_len = ((span + RT_WORK_PARTS - 1) / RT_WORK_PARTS + gran - 1) / gran * gran;
    c.part = blockIdx.x % RT_WORK_PARTS;
    c.parts_done = 0;
    c.seen = 0;
    return c;
}
// entries of the current part that nobody had claimed at this wave's last fetch (guides the size of the next one: towards
// the end of a queue a wave takes less, so that a launch does not end with a few waves working off large last helpings)
RT_DEV uint32_t rtPartLeft(const RtPartCursor &c) {
    const uint32_t lo = c.first + c.part * c.part_len, hi = min(lo + c.part_len, c.n);
    return hi > lo + c.seen ? hi - lo - c.seen : 0u;
}
RT_DEV bool rtPartFetch(RtPartCursor &c, uint32_t want, uint32_t &base, uint32_t &end) {
    while (c.parts_done < RT_WORK_PARTS) {
        const uint32_t lo = c.first + c.part * c.part_len, hi = min(lo + c.part_len, c.n);
        uint32_t b = 0;
        if (laneId() == 0) b = atomicAdd(c.cursors + c.part * RT_CNT_STRIDE, want);
        b = __builtin_amdgcn_readfirstlane(b);
        if (lo < c.n && lo + b < hi) {
            base = lo + b;
            end = min(base + want, hi);
            c.seen = b + want;
            return true;
        }
        c.part = c.part + 1 == RT_WORK_PARTS ? 0u : c.part + 1;  // this part is used up: on to the next
        c.parts_done++;
        c.seen = 0;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------------------------
// BVH traversal: 4-wide quantised nodes and triangle records in ONE array of 64-byte records (rt_types.h).  A step
// fetches the record the lane's `cur` refers to -- four 16-byte loads issued by every lane of the wave together,
// whether the lane is at a node or at a triangle -- and then either tests the four child boxes (near-first
// descent, the other hit children pushed far-first) or runs the reference's exact triangle test.
struct RtTraversalStats { uint32_t nodes, tris; };

typedef __attribute__((address_space(3))) uint32_t RtLdsWord;  // typed LDS pointer: pops stay ds_read, never flat loads
// The stack's bases are WAVE-uniform (SGPRs); the lane's own column is re-derived at each push/pop with two mbcnt
// instructions.  Holding per-lane base pointers (1 + 2 VGPRs) across the walk made the allocator spill them at 64
// VGPRs and reload them from scratch before every LDS access.
RT_DEV uint32_t laneNow() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
struct RtStack {          // per-lane stack: the first `lds_entries` live in LDS ([entry][lane], conflict free),
    RtLdsWord *lds;       // deeper ones in a global spill area ([entry][grid thread], coalesced); both point at the
    uint32_t *spill;      // wave's lane 0
    uint32_t spill_stride;
    int lds_entries;
};
RT_DEV void rtPush(const RtStack &s, int &sp, uint32_t v) {
    if (sp < s.lds_entries) s.lds[sp * RT_BLOCK + laneNow()] = v;
    else s.spill[(size_t)(sp - s.lds_entries) * s.spill_stride + laneNow()] = v;
    sp++;
}
RT_DEV uint32_t rtPop(const RtStack &s, int &sp) {  // requires sp > 0
    --sp;
    return sp < s.lds_entries ? s.lds[sp * RT_BLOCK + laneNow()] : s.spill[(size_t)(sp - s.lds_entries) * s.spill_stride + laneNow()];
}
// the same with the lane's column handed in (a node step pushes up to three times: one mbcnt pair for all of them)
RT_DEV void rtPushL(const RtStack &s, int &sp, uint32_t v, uint32_t ln) {
    if (sp < s.lds_entries) s.lds[sp * RT_BLOCK + ln] = v;
    else s.spill[(size_t)(sp - s.lds_entries) * s.spill_stride + ln] = v;
    sp++;
}
RT_DEV uint32_t rtPopL(const RtStack &s, int &sp, uint32_t ln) {  // requires sp > 0
    --sp;
    return sp < s.lds_entries ? s.lds[sp * RT_BLOCK + ln] : s.spill[(size_t)(sp - s.lds_entries) * s.spill_stride + ln];
}
RT_DEV RtStack rtMakeStack(uint32_t *lds_base, uint32_t *spill, int lds_entries) {
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u);  // first thread of this wave
    RtStack s;
    s.lds = (RtLdsWord *)(lds_base + wave0);
    s.spill = spill + (size_t)blockIdx.x * RT_BLOCK + wave0;
    s.spill_stride = gridDim.x * RT_BLOCK;
    s.lds_entries = lds_entries;
    return s;
}
RT_DEV float safeInv(float d) {
    // 1/d with |1/d| capped so that o*inv stays finite (a zero component must not turn the fma into inf-inf)
    float inv = 1.0f / d;
    const float cap = 1e30f;
    if (!(fabsf(inv) <= cap)) inv = (__float_as_uint(d) >> 31) ? -cap : cap;
    return inv;
}

struct RtWalk {  // per-lane traversal state
    RtRay r;
    V3 inv, oi, err;   // 1/d, o/d and the per-axis error bound of a plane distance (set-up only: analytic pre-tests, packets)
    V3 oiN, oiF;       // o/d with the error bound folded in: oi + err for near planes, oi - err for far planes (node steps)
    float best_t;      // closest hit so far (closest) or the light distance (any-hit)
    uint32_t best_id;
    int best_obj;      // object and reference candidate rank of the best hit: the exact-tie rule needs them
    uint32_t best_rank;
    uint32_t cur;      // record reference: node index, RT_NODE_LEAF | (triangles left - 1) << 28 | record index, or DONE
    uint32_t pend;     // postponed leaf (same encoding) whose triangles are still to be tested, or DONE = none
    int sp;
};

// Plane distances are t = fma(q, scale/d, fma(origin, 1/d, -o/d)): the product q*scale is exact, so the absolute
// error is at most 2^-24 (2|o/d| + |origin/d| + 2|t|).  err covers the first two terms with margin (applied per axis:
// near planes move towards the origin, far planes away), the factor on tfar covers the relative one.
RT_DEV void rtInitWalk(const RtSceneDev &sc, RtWalk &w, uint32_t root_ref) {
    w.inv = v3(safeInv(w.r.d.x), safeInv(w.r.d.y), safeInv(w.r.d.z));
    w.oi = v3(w.r.o.x * w.inv.x, w.r.o.y * w.inv.y, w.r.o.z * w.inv.z);
    const float u = 4.76837158e-07f;  // 2^-21
    w.err = v3((fabsf(w.oi.x) + sc.bound * fabsf(w.inv.x)) * u, (fabsf(w.oi.y) + sc.bound * fabsf(w.inv.y)) * u,
               (fabsf(w.oi.z) + sc.bound * fabsf(w.inv.z)) * u);
    w.oiN = w.oi + w.err;
    w.oiF = w.oi - w.err;
    w.sp = 0;
    w.cur = sc.n_tris ? root_ref : RT_NODE_DONE;
    w.pend = RT_NODE_DONE;
}

RT_DEV float ubyteF(uint32_t v, int c) { return (float)((v >> (8 * c)) & 0xFFu); }  // v_cvt_f32_ubyte{c}

// A traversal step is split in two so that a wave can run them as separate phases (traceQueue): rtNodeStep tests the
// four child boxes of the node `cur` refers to, rtTriStep runs the reference's exact test on the next triangle of
// the postponed leaf `pend`.  Both fetch one 64-byte record with exactly four 16-byte loads per lane; the empty asm
// pins them as whole registers (left alone, the compiler re-slices the record into six narrower loads: 50 % more L1
// accesses).  The kernels are VALU bound, so pushes and pops are conditional (skipped work costs only scalar branches).
typedef float v4f __attribute__((ext_vector_type(4)));
#define RT_LOAD_RECORD(ref)                                                                                    \
    const v4f *rec_ = reinterpret_cast<const v4f *>(sc.nodes) + 4 * (size_t)((ref) & RT_NODE_INDEX_MASK);      \
    v4f q0 = rec_[0], q1 = rec_[1], q2 = rec_[2], q3 = rec_[3];                                                \
    asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));

// Returns true when an any-hit ray is occluded.  Afterwards w.pend is the leaf's next triangle or RT_NODE_DONE (none).
template <bool ANY, bool STATS>
RT_DEV bool rtTriStep(const RtSceneDev &sc, RtWalk &w, RtTraversalStats &st) {
    // (an any-hit test never reads the record's fourth word -- object, candidate rank, ids: three loads, a quarter less L1 traffic)
    const v4f *rec_ = reinterpret_cast<const v4f *>(sc.nodes) + 4 * (size_t)(w.pend & RT_NODE_INDEX_MASK);
    v4f q0 = rec_[0], q1 = rec_[1], q2 = rec_[2], q3 = ANY ? v4f{0, 0, 0, 0} : rec_[3];
    if (ANY) asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2));
    else asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
    if (STATS) st.tris++;
    RtTri T;
    T.A = v3(q0.x, q0.y, q0.z);
    T.B = v3(q1.x, q1.y, q1.z);
    T.C = v3(q2.x, q2.y, q2.z);
    T.n = v3(q0.w, q1.w, q2.w);
    float t = 0.0f;
    const bool hit = rtTriHit(T, w.r, w.best_t, t);
    if (ANY) {
        if (hit) return true;
    } else {
        const int obj = (int)__float_as_uint(q3.x);
        const uint32_t rank = __float_as_uint(q3.y);
        // strict minimum; on an exact tie the earlier object wins (main.cpp:67) and inside a mesh the earlier
        // candidate of the reference's order (MeshLoader.h:399-404).  A miss (distance INFINITY) never loses to inf.
        // (ONE set of selects after the test: updating the best hit inside the test's nested early exits made the
        // compiler copy the four unchanged values on every exit edge, 20 v_mov per triangle step)
        const bool take = hit && (t < w.best_t || (t == w.best_t && w.best_id != RT_ID_MISS &&
                                                   (obj < w.best_obj || (obj == w.best_obj && rank < w.best_rank))));
        w.best_t = take ? t : w.best_t;
        w.best_id = take ? (((w.pend & RT_NODE_INDEX_MASK) - (uint32_t)sc.n_nodes) | __float_as_uint(q3.w)) : w.best_id;  // (| RT_ID_GENERAL flag)
        w.best_obj = take ? obj : w.best_obj;
        w.best_rank = take ? rank : w.best_rank;
    }
    if ((w.pend >> 28) & 7u) w.pend += 1u - 0x10000000u;  // next triangle of the leaf
    else w.pend = RT_NODE_DONE;
    return false;
}

// Afterwards w.cur is the nearest hit child (a node or a leaf reference), the next stack entry, or RT_NODE_DONE.
template <bool ANY, bool STATS>
RT_DEV void rtNodeStep(const RtSceneDev &sc, RtWalk &w, const RtStack &stk, RtTraversalStats &st) {
    RT_LOAD_RECORD(w.cur)
    const uint32_t ln = laneNow();  // the lane's stack column, once for every push / pop of this step
    if (STATS) st.nodes++;
    const V3 m = v3(q0.w * w.inv.x, q3.z * w.inv.y, q3.w * w.inv.z);  // scale / d (scale is a power of two: exact)
    // plane distance = fma(q, scale/d, fma(origin, 1/d, -(o/d +- err))): the error bound rides in the second operand
    // (rounding oi +- err once at set-up stays far inside the bound's 8x margin)
    const float anx = __builtin_fmaf(q0.x, w.inv.x, -w.oiN.x), any_ = __builtin_fmaf(q0.y, w.inv.y, -w.oiN.y),
                anz = __builtin_fmaf(q0.z, w.inv.z, -w.oiN.z);
    const float afx = __builtin_fmaf(q0.x, w.inv.x, -w.oiF.x), afy = __builtin_fmaf(q0.y, w.inv.y, -w.oiF.y),
                afz = __builtin_fmaf(q0.z, w.inv.z, -w.oiF.z);
    const bool negx = w.inv.x < 0.0f, negy = w.inv.y < 0.0f, negz = w.inv.z < 0.0f;
    const uint32_t lox = __float_as_uint(q2.x), loy = __float_as_uint(q2.y), loz = __float_as_uint(q2.z);
    const uint32_t hix = __float_as_uint(q2.w), hiy = __float_as_uint(q3.x), hiz = __float_as_uint(q3.y);
    const uint32_t nrx = negx ? hix : lox, frx = negx ? lox : hix;
    const uint32_t nry = negy ? hiy : loy, fry = negy ? loy : hiy;
    const uint32_t nrz = negz ? hiz : loz, frz = negz ? loz : hiz;
    uint32_t ref[4] = {__float_as_uint(q1.x), __float_as_uint(q1.y), __float_as_uint(q1.z), __float_as_uint(q1.w)};
    uint32_t key[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float tnx = __builtin_fmaf(ubyteF(nrx, c), m.x, anx), tfx = __builtin_fmaf(ubyteF(frx, c), m.x, afx);
        const float tny = __builtin_fmaf(ubyteF(nry, c), m.y, any_), tfy = __builtin_fmaf(ubyteF(fry, c), m.y, afy);
        const float tnz = __builtin_fmaf(ubyteF(nrz, c), m.z, anz), tfz = __builtin_fmaf(ubyteF(frz, c), m.z, afz);
        const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
        const float tf = fminf(fminf(tfx, tfy), fminf(tfz, w.best_t)) * 1.000001f;
        const bool hit = tn <= tf && ref[c] != 0xFFFFFFFFu;
        key[c] = hit ? __float_as_uint(tn) : 0xFFFFFFFFu;  // tn >= 0: bit order == value order
    }
    if (ANY && !sc.any_ordered) {
        // any-hit: order is irrelevant to the result -- enter the first hit child, push the others (no sort).  In a
        // large scene the order matters for the WORK: see any_ordered in rt_abi.cpp
        uint32_t next = RT_NODE_DONE;
#pragma unroll
        for (int c = 3; c >= 0; c--)
            if (key[c] != 0xFFFFFFFFu) {
                if (next != RT_NODE_DONE) rtPushL(stk, w.sp, next, ln);
                next = ref[c];
            }
        w.cur = next != RT_NODE_DONE ? next : (w.sp ? rtPopL(stk, w.sp, ln) : RT_NODE_DONE);
        return;
    }
    // sort (entry distance, reference) pairs: 5 comparators
#define RT_CSWAP(i, j)                                                            \
    {                                                                             \
        const bool sw = key[j] < key[i];                                          \
        const uint32_t ki = sw ? key[j] : key[i], kj = sw ? key[i] : key[j];      \
        const uint32_t ri = sw ? ref[j] : ref[i], rj = sw ? ref[i] : ref[j];      \
        key[i] = ki; key[j] = kj; ref[i] = ri; ref[j] = rj;                       \
    }
    RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
#undef RT_CSWAP
    if (key[0] == 0xFFFFFFFFu) {
        w.cur = w.sp ? rtPopL(stk, w.sp, ln) : RT_NODE_DONE;
    } else {  // descend into the nearest child, push the others far-first
        if (key[3] != 0xFFFFFFFFu) rtPushL(stk, w.sp, ref[3], ln);
        if (key[2] != 0xFFFFFFFFu) rtPushL(stk, w.sp, ref[2], ln);
        if (key[1] != 0xFFFFFFFFu) rtPushL(stk, w.sp, ref[1], ln);
        w.cur = ref[0];
    }
}

// moves a leaf reference from cur to the (empty) pend slot and takes the next stack entry
RT_DEV void rtPostponeLeaf(RtWalk &w, const RtStack &stk) {
    if (w.pend == RT_NODE_DONE && (w.cur & RT_NODE_LEAF) && w.cur != RT_NODE_DONE) {
        w.pend = w.cur;
        w.cur = w.sp ? rtPop(stk, w.sp) : RT_NODE_DONE;
    }
}

// Analytic primitives are staged once per block into LDS (a few hundred bytes; every lane of every ray reads all of
// them, so they are the hottest scene data) behind the traversal stack: [stack][spheres][planes].
struct RtAnalytic {
    const RtSphere *spheres;
    const RtPlane *planes;
};
RT_DEV RtAnalytic stageAnalytic(const RtSceneDev &sc, uint32_t *lds_after_stack) {
    uint32_t *dst = lds_after_stack;
    const uint32_t nsw = (uint32_t)sc.n_spheres * (sizeof(RtSphere) / 4), npw = (uint32_t)sc.n_planes * (sizeof(RtPlane) / 4);
    const uint32_t *ss = reinterpret_cast<const uint32_t *>(sc.spheres), *ps = reinterpret_cast<const uint32_t *>(sc.planes);
    for (uint32_t k = threadIdx.x; k < nsw; k += blockDim.x) dst[k] = ss[k];
    for (uint32_t k = threadIdx.x; k < npw; k += blockDim.x) dst[nsw + k] = ps[k];
    __syncthreads();
    return RtAnalytic{reinterpret_cast<const RtSphere *>(dst), reinterpret_cast<const RtPlane *>(dst + nsw)};
}
// The box pre-test of the unit primitives is only sound where the reference's own test is accurate.  That test works in the
// primitive's object space, in floats: `cdotc - cdotd * cdotd` (Objects.h:44-49) cancels once the origin is many radii away, and
// from thousands of radii it accepts every other ray (seen from 1 300 scene sizes the reference draws a 0.03 sphere as a blob a
// third of the frame wide: scripts/gpu_r3_far_camera.py).  A frame whose CAMERA stands farther out than four times the scene's
// reach therefore traces its primary rays without the pre-test (RtQueues::prim_far, a launch argument: the packet phase) and
// gets the reference's arithmetic, whatever it says.  (Deciding per ray in the per-lane walks -- a bounce off the horizon of
// an infinite plane starts as far out -- cost the 4K frame 6 to 11 ms in registers and code: not done)
// padded world box of a sphere: a ray that misses it cannot hit the sphere (plain slab test, no quantisation)
RT_DEV bool sphereBoxHit(const RtSphere &sp, V3 inv, V3 oi) {
    const float ax = __builtin_fmaf(sp.box_mn[0], inv.x, -oi.x), bx = __builtin_fmaf(sp.box_mx[0], inv.x, -oi.x);
    const float ay = __builtin_fmaf(sp.box_mn[1], inv.y, -oi.y), by = __builtin_fmaf(sp.box_mx[1], inv.y, -oi.y);
    const float az = __builtin_fmaf(sp.box_mn[2], inv.z, -oi.z), bz = __builtin_fmaf(sp.box_mx[2], inv.z, -oi.z);
    const float tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return tmin <= tmax * 1.0000005f;
}

// No triangle of the scene is farther from a ray's origin than this, in units of the ray's parameter: every corner lies within
// sc.bound of the coordinate origin on each axis, and 1 / max |d_k| >= 1 / |d| (rays handed in by a host need not be unit).
// Twice what the geometry allows.
RT_DEV float rtFarLimit(const RtSceneDev &sc, const RtWalk &w) {
    const float reach = 2.0f * (fabsf(w.r.o.x) + fabsf(w.r.o.y) + fabsf(w.r.o.z)) + 4.0f * sc.bound;
    return reach * fminf(fminf(fabsf(w.inv.x), fabsf(w.inv.y)), fabsf(w.inv.z));
}

// closest() over the analytic primitives, in scene order (ties resolved by object order, main.cpp:67).
// A ray that meets none of them starts its walk with best_t = rtFarLimit instead of infinity (best_id stays MISS, and a miss is
// reported with an infinite distance): the reference's triangle test accepts points millions of scene sizes away when a ray
// is almost parallel to a triangle's plane (there cross(B - P, C - P) is exactly zero) -- IF its walk happens to test that
// triangle, so which of these a ray "hit" depended on the tree, on the walk (packet boxes are exact, per-lane boxes 8-bit)
// and on the 63 rays sharing its packet.  With the limit a ray's result is a function of the ray and the scene alone
// (what that costs in agreement with the reference -- its own such hits, ~1 per 10^8 rays -- and buys).
template <bool CULL>
RT_DEV void analyticClosestT(const RtSceneDev &sc, const RtAnalytic &an, RtWalk &w) {
    w.best_t = INFINITY;
    w.best_id = RT_ID_MISS;
    w.best_obj = 0x7FFFFFFF;
    w.best_rank = 0;
    for (int k = 0; k < sc.n_spheres; k++) {
        const RtSphere &sp = an.spheres[k];
        if (CULL && !sphereBoxHit(sp, w.inv, w.oi)) continue;
        float dist;
        V3 ni, P;
        if (rtUnitPrimHit(sp, w.r, dist, ni, P) && (dist < w.best_t || (dist == w.best_t && w.best_id != RT_ID_MISS && sp.object < w.best_obj))) {
            w.best_t = dist;
            w.best_id = RT_ID_SPHERE | (uint32_t)k;
            w.best_obj = sp.object;
        }
    }
    for (int k = 0; k < sc.n_planes; k++) {
        const RtPlane &pl = an.planes[k];
        float t;
        if (rtPlaneHit(pl, w.r, t) && (t < w.best_t || (t == w.best_t && w.best_id != RT_ID_MISS && pl.object < w.best_obj))) {
            w.best_t = t;
            w.best_id = RT_ID_PLANE | (uint32_t)k | (((sc.plane_general >> k) & 1ull) ? RT_ID_GENERAL : 0u);
            w.best_obj = pl.object;
        }
    }
    if (w.best_id == RT_ID_MISS) w.best_t = rtFarLimit(sc, w);
}
RT_DEV void analyticClosest(const RtSceneDev &sc, const RtAnalytic &an, RtWalk &w) { analyticClosestT<true>(sc, an, w); }
// is_shadowed() after its back-face test, analytic part: any primitive with hit.distance <= dist (main.cpp:50-55)
template <bool CULL>
RT_DEV bool analyticOccludedT(const RtSceneDev &sc, const RtAnalytic &an, const RtWalk &w, float dist) {
    for (int k = 0; k < sc.n_planes; k++) {
        float t;
        if (rtPlaneHit(an.planes[k], w.r, t) && t <= dist) return true;
    }
    for (int k = 0; k < sc.n_spheres; k++) {
        const RtSphere &sp = an.spheres[k];
        if (CULL && !sphereBoxHit(sp, w.inv, w.oi)) continue;
        float d;
        V3 ni, P;
        if (rtUnitPrimHit(sp, w.r, d, ni, P) && d <= dist) return true;
    }
    return false;
}
RT_DEV bool analyticOccluded(const RtSceneDev &sc, const RtAnalytic &an, const RtWalk &w, float dist) {
    return analyticOccludedT<true>(sc, an, w, dist);
}

// whole-ray forms (test hook k_hitinfo)
RT_DEV bool occluded(const RtSceneDev &sc, const RtAnalytic &an, uint32_t root_ref, const RtRay &r, float dist,
                     const RtStack &stk, RtTraversalStats &st) {
    RtWalk w;
    w.r = r;
    rtInitWalk(sc, w, root_ref);
    if (analyticOccluded(sc, an, w, dist)) return true;
    w.best_t = dist;
    w.best_id = RT_ID_MISS;
    w.best_obj = 0;
    w.best_rank = 0;
    for (;;) {
        rtPostponeLeaf(w, stk);
        if (w.pend != RT_NODE_DONE) {
            if (rtTriStep<true, false>(sc, w, st)) return true;
        } else if (w.cur != RT_NODE_DONE) {
            rtNodeStep<true, false>(sc, w, stk, st);
        } else {
            return false;
        }
    }
}

RT_DEV void flushStats(uint32_t *counters, const RtTraversalStats &st, int nodes_word) {
    unsigned long long n = st.nodes, t = st.tris;
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_down(n, off);
        t += __shfl_down(t, off);
    }
    if (laneId() == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(counters + nodes_word), n);
        atomicAdd(reinterpret_cast<unsigned long long *>(counters + nodes_word + 2), t);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// tile / sample indexing.  A chunk covers the local tiles tile0, tile0 + stride, tile0 + 2 stride, ... (ntiles of them;
// stride = q.tile_stride = the frame's number of chunks: every chunk is a sample of the WHOLE image, so all chunks hold the
// same mix of cheap and expensive tiles, every iteration of the pipeline carries the same load, and the fan-out seen on one
// frame sizes the queues of any other cut of it).  Sample index inside the chunk, k = 0 .. ntiles - 1:
//   (k * 256 + m) * spp + s,   m = Morton index of the pixel inside its 16x16 tile
// so that a wave's 64 lanes are 64/spp neighbouring pixels (a 4x4 block at 4 spp) x spp samples.
RT_DEV uint32_t compact1By1(uint32_t x) {
    x &= 0x55555555u;
    x = (x ^ (x >> 1)) & 0x33333333u;
    x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
    x = (x ^ (x >> 4)) & 0x00ff00ffu;
    return x;
}

// Level 0 needs no compaction: slot == sample index inside the chunk; the slots [0, total) of the iteration's ray queue
// are reserved for them (k_iter_ctl starts the queue's counter at `total`, k_shade appends behind).  Slots of pixels
// outside the image (ragged edge tiles) are flagged invalid; k_closest records a miss for them and k_shade ignores them.
__global__ __launch_bounds__(RT_BLOCK) void k_gen_primary(RtFrameDev f, RtQueues q, int par, uint32_t tile0, uint32_t ntiles) {
    const uint32_t total = ntiles * RT_TILE_PIXELS * (uint32_t)f.spp;
    uint32_t invalid = 0;
    for (uint32_t idx = blockIdx.x * RT_BLOCK + threadIdx.x; idx < total; idx += gridDim.x * RT_BLOCK) {
        const uint32_t s = idx % (uint32_t)f.spp, pm = idx / (uint32_t)f.spp;
        const uint32_t m = pm & 255u, lt = tile0 + (pm >> 8) * q.tile_stride;
        const uint32_t gt = lt * (uint32_t)f.tile_world + (uint32_t)f.tile_rank;
        const int i = (int)((gt % (uint32_t)f.tiles_x) * RT_TILE + compact1By1(m));
        const int j = (int)((gt / (uint32_t)f.tiles_x) * RT_TILE + compact1By1(m >> 1));
        const uint32_t tag = q.prim_sample0 + idx;  // ring sample, level 0
        float4 A = make_float4(0, 0, 0, 0), B = make_float4(0, 0, 1, __uint_as_float(tag | RT_FLAG_INVALID));
        if (i < f.width && j < f.height) {
            float jx, jy;
            if (f.use_table) {  // jitterMatrix, main.cpp:601-606 (selects, not an indexed array: no scratch)
                const uint32_t k = s & 3u;
                jx = k == 0 ? (float)(-1.0 / 4.0) : (k == 1 ? (float)(3.0 / 4.0) : (k == 2 ? (float)(-3.0 / 4.0) : (float)(1.0 / 4.0)));
                jy = k == 0 ? (float)(3.0 / 4.0) : (k == 1 ? (float)(1.0 / 3.0) : (k == 2 ? (float)(-1.0 / 4.0) : (float)(-3.0 / 4.0)));
            } else {
                rtHashJitter((uint32_t)(j * f.width + i), s, f.seed, jx, jy);
            }
            const float dx = f.X + ((float)i + jx) * f.s + f.s / 2;  // main.cpp:629-630
            const float dy = f.Y - ((float)j + jy) * f.s - f.s / 2;
            const V3 d = normalize(mulPoint(f.rot, v3(dx, dy, f.dz), 0.0f));  // :637-641
            A = make_float4(f.origin[0], f.origin[1], f.origin[2], 1.0f);
            B = make_float4(d.x, d.y, d.z, __uint_as_float(tag));
        } else {
            invalid++;
        }
        // ((origin, weight) is the same for every primary ray: RtQueues::primA; the Fresnel word is only read for reflection
        // rays with a refraction sibling)
        if (!q.prim_implied) q.rayA[par][idx] = A;
        q.rayB[par][idx] = B;
    }
    if (__ballot(invalid != 0)) {
        for (int off = 32; off > 0; off >>= 1) invalid += __shfl_down(invalid, off);
        if (laneId() == 0 && invalid) atomicAdd(q.counters + RT_CNT_INVALID, invalid);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Traversal kernels: persistent lanes with dynamic ray fetch.  A wave reserves RT_FETCH_BATCH_* consecutive queue
// entries at a time (one returning atomic per batch); a lane whose ray has terminated takes the next entry of the
// wave's batch instead of idling until the slowest ray of its wave is done.  The walk leaves its loop to refill as
// soon as fewer than RT_REFILL_BELOW lanes hold a ray.  Ray order does not affect results: hit records go to the
// ray's own slot and radiance is accumulated in fixed point.
// Batch sizes (overridable for parameter sweeps, scripts/build_variants.sh; measured on the 4K/64 spp bench): the
// work counter is ONE address, and at 256 shadow rays per atomic its returning atomics saturate (k_any 332 ms; 512:
// 298 ms; 1024: 300 ms); closest-hit queues hold fewer, longer walks and prefer the better balance of 256 (234 ms;
// 1024: 247 ms).  Refill threshold: 16 lanes is 4 ms better than 24 or 32.
// Swept again at the end of round 3 (fused launch, 8 cursors per queue, sorted batches, pre-split tree; frame 387.2 ms):
// shadow batches of 256 / 1024 / 2048 / 4096: +8.9 / -2.4 / -3.3 / -2.3 ms; closest 512 (unsorted): +11; refill 12 / 24: +2.
#ifndef RT_FETCH_BATCH_ANY
#define RT_FETCH_BATCH_ANY 2048u
#endif
#ifndef RT_FETCH_BATCH_CLOSEST
#define RT_FETCH_BATCH_CLOSEST 256u
#endif
#ifndef RT_REFILL_BELOW
#define RT_REFILL_BELOW 16
#endif

// Shadow-queue record -> the Ray of is_shadowed() (main.cpp:49): the record holds the shaded point and the light's
// index; direction, light distance and the epsilon-offset origin are recomputed with the expressions k_shade uses
// (same operations, same order => same bits), which keeps the record at 32 bytes instead of 48.
RT_DEV void rtShadowRay(const RtSceneDev &sc, float4 A, uint32_t light, RtRay &r, float &dist, uint32_t &sample) {
    sample = __float_as_uint(A.w);  // ring sample | RT_SHADOW_L0
    const V3 P = v3(A.x, A.y, A.z), lp = ld3(sc.lights[light].pos);
    const V3 ld = normalize(lp - P);  // main.cpp:97
    dist = length(lp - P);            // :98
    r.o = P + RT_EPSILON * ld;        // :49
    r.d = ld;
}

//
// Long any-hit walks.  A launch cannot end before its longest ray, and a shadow ray can be very long: one that leaves a
// point 10^7 units away (a horizon pixel on an infinite plane) carries a slab-test error bound many units wide and walks
// through everything near its line (profiles/r01_configs.md: 712 724 steps for one ray, 0.57 s).  The subtrees on an
// any-hit ray's stack are independent -- the ray is occluded iff ANY of them holds a hit -- so a lane whose ray has been
// walking for more than `budget` wave iterations EXPORTS it: every stack entry (and cur / pend) becomes a work item
// (ray index, subtree) in q.split, and the lane takes the next ray.  Up to RT_SPLIT_ROUNDS follow-up launches
// (MODE 2) walk the items with all lanes, exporting again what is still long; the last one has no budget.  The ray's
// join word (shC.w as u32) counts its outstanding items; an item that finds a hit sets RT_JOIN_OCCLUDED; the item
// that brings the count to zero adds the pending radiance if the flag is clear.  Same sums, same bits as unsplit.
#define RT_MODE_QUEUE 0   // the iteration's queue (closest-hit: entries [first, n) -- the primary rays before them walk as packets)
#define RT_MODE_SPLIT 2   // work items of split any-hit walks (round `round` of the iteration)

// What a traversal block keeps in LDS: the per-lane stacks, one row with the wave iteration at which each lane took its ray
// (any-hit walks), and the staged analytic primitives.  Set up ONCE per block (stageAnalytic has the block's only
// barrier): the phases of the fused kernel (k_trace) share it, and a wave moves from one phase to the next on its own.
struct RtTraceLds {
    RtStack stk;
    RtLdsWord *age;   // 64 words per wave: any-hit walks keep the lanes' ages here, closest-hit walks the order of their batch (256 bytes)
    RtLdsWord *aux;   // 64 words per wave: bin counters of rtSortBatch
    RtAnalytic an;
};
RT_DEV RtTraceLds rtTraceSetup(const RtSceneDev &sc, const RtQueues &q, uint32_t *lds, int stack_entries) {
    RtTraceLds t;
    t.stk = rtMakeStack(lds, q.spill, stack_entries);
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u);
    t.age = (RtLdsWord *)(lds + stack_entries * RT_BLOCK + wave0);
    t.aux = (RtLdsWord *)(lds + (stack_entries + 1) * RT_BLOCK + wave0);
    t.an = stageAnalytic(sc, lds + (stack_entries + RT_TRACE_LDS_ROWS) * RT_BLOCK);
    return t;
}

// Order of consumption inside a batch of closest-hit rays.  A wave is as slow as its slowest lane, and how long a ray walks
// depends on where it goes: the <= 256 rays of a batch (consecutive queue entries, i.e. rays that left neighbouring hits) are
// consumed grouped by KIND (the refraction ray of a reflection / refraction pair dives into the object it entered, the
// reflection ray crosses the scene) and by the OCTANT of their direction, so that the rays a wave holds at any time walk
// alike.  Order is free -- results go to the ray's own slot.  A counting sort through LDS: one returning LDS add per ray
// gives its rank inside its bin (64 bins), a wave scan the bins' first slots; perm[k] (a byte) = the batch entry consumed k-th.
// All 64 lanes call this together; len <= 256.
#ifndef RT_SORT_BATCH
#define RT_SORT_BATCH 1
#endif
typedef __attribute__((address_space(3))) uint8_t RtLdsByte;
RT_DEV void rtSortBatch(const RtQueues &q, int buf, uint32_t pool_base, uint32_t len, RtLdsWord *permw, RtLdsWord *bins) {
    const uint32_t lane = laneId();
    RtLdsByte *perm = (RtLdsByte *)permw;
    bins[lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t kr[4];        // bin | rank inside the bin << 8, per sub-block of 64 entries
    uint32_t carry = 0u;   // the entry in front of the sub-block's first is a reflection ray with a refraction sibling
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t idx = (uint32_t)k * 64u + lane;
        const bool valid = idx < len;
        float4 B = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(RT_FLAG_INVALID));
        if (valid) B = q.rayB[buf][pool_base + idx];
        const uint32_t tag = __float_as_uint(B.w);
        const uint32_t sib = (tag & RT_FLAG_HAS_REFR_SIBLING) ? 1u : 0u;
        uint32_t prev = __shfl_up(sib, 1);
        if (lane == 0u) prev = carry;
        carry = __builtin_amdgcn_readlane(sib, 63);
        const uint32_t oct = (__float_as_uint(B.x) >> 31) | ((__float_as_uint(B.y) >> 31) << 1) | ((__float_as_uint(B.z) >> 31) << 2);
#ifndef RT_SORT_KEY
#define RT_SORT_KEY 0
#endif
#if RT_SORT_KEY == 0
        const uint32_t bin = (tag & RT_FLAG_INVALID) ? 63u : (oct | (prev << 3));
#elif RT_SORT_KEY == 1
        const uint32_t bin = (tag & RT_FLAG_INVALID) ? 63u : oct;
#elif RT_SORT_KEY == 2
        const float ax = fabsf(B.x), ay = fabsf(B.y), az = fabsf(B.z);
        const uint32_t major = ax >= ay && ax >= az ? 0u : (ay >= az ? 1u : 2u);
        const uint32_t bin = (tag & RT_FLAG_INVALID) ? 63u : (major | (oct << 2) | (prev << 5));
#elif RT_SORT_KEY == 3
        const uint32_t lvl = min((tag >> q.level_shift) & q.level_mask, 3u);
        const uint32_t bin = (tag & RT_FLAG_INVALID) ? 63u : (oct | (prev << 3) | (lvl << 4));
#elif RT_SORT_KEY == 4
        const uint32_t bin = (tag & RT_FLAG_INVALID) ? 63u : prev;
#endif
        uint32_t rank = 0u;
        if (valid) rank = __hip_atomic_fetch_add(bins + bin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        kr[k] = bin | (rank << 8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t cnt = bins[lane];
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += t;
    }
    __builtin_amdgcn_wave_barrier();
    bins[lane] = incl - cnt;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t idx = (uint32_t)k * 64u + lane;
        if (idx < len) perm[bins[kr[k] & 255u] + (kr[k] >> 8)] = (uint8_t)idx;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// L0: level-0 shadow results are stored, not added (RtQueues::occ0) -- an instance of its own, so that the usual one keeps
// its registers
// TAIL: the queue's last entries are fetched one generation at a time (a launch must not end with a few waves working off
// large last batches); off where another phase of the same launch follows, whose work fills the chip meanwhile.
template <bool ANY, int STATS, int MODE, bool L0 = false, bool TAIL = true>  // STATS: 0 none, 1 counters (records fetched, longest walk), 2 + wave-time profile
RT_DEV void traceQueue(const RtSceneDev &sc, uint32_t root_ref, const RtQueues &q, const RtTraceLds &tl, int par, uint32_t first, int round,
                       uint32_t budget) {
    constexpr bool COUNT = STATS >= 1, PROF = STATS >= 2;
    const RtStack &stk = tl.stk;
    RtLdsWord *const age = tl.age;  // one LDS row behind the stack: the wave iteration at which the lane took its ray
    const RtAnalytic &an = tl.an;
    const int buf = par;
    constexpr bool SPLIT = MODE == RT_MODE_SPLIT;
    constexpr uint32_t batch_max = ANY ? RT_FETCH_BATCH_ANY : RT_FETCH_BATCH_CLOSEST;
    static_assert(ANY || !SPLIT, "only any-hit walks are split");
    const uint32_t n = SPLIT ? min(q.counters[RT_CNT_SPLIT(par, round)], q.split_cap)
                             : (ANY ? min(q.counters[RT_CNT_SHADOW(par)], q.shadow_cap) : min(q.counters[RT_CNT_RAYS(par)], q.ray_cap));
    uint32_t *work = q.counters + RT_CNT_SPLIT(par, round) + 1;  // (split rounds: one cursor)
    // The queue is cut into RT_WORK_PARTS contiguous parts with a cursor each (on its own 128-byte line): one
    // cursor for the whole chip saturates -- a returning atomic on one address retires every ~10 ns.  A wave starts
    // in the part of its XCD (workgroups go round-robin over the 8 XCDs) and moves on when a part is used up.
    constexpr uint32_t PARTS = MODE == RT_MODE_QUEUE ? RT_WORK_PARTS : 1u;
    uint32_t *work_parts = q.counters + (ANY ? RT_CNT_WORKP_ANY(par, 0) : RT_CNT_WORKP_CLOSEST(par, 0));
    const uint32_t span = n > first ? n - first : 0u;     // entries [first, n) are this launch's
    // (shadow queues of small chunks -- a small workspace, one GPU's share of a frame -- take smaller batches, down to 512: a
    // wave should get ~4 of them, or the last ones keep a few waves busy while the chip idles)
    uint32_t batch = batch_max;
    if (ANY && MODE == RT_MODE_QUEUE)
        while (batch > 512u && batch * (gridDim.x * (RT_BLOCK / 64) * 4u) > span) batch >>= 1;
    const bool prim_here = !ANY && q.prim_implied && first < q.prim_n;  // primary rays in this queue walk: their origin is RtQueues::primA
    const uint32_t part_len = PARTS > 1 ? ((span + PARTS - 1) / PARTS + batch - 1) / batch * batch : span;
    uint32_t part = PARTS > 1 ? blockIdx.x % PARTS : 0u, parts_done = 0;  // wave-uniform
    const uint2 *items_in = q.split[round & 1];
    uint2 *items_out = q.split[(round + 1) & 1];          // the queue pass is round -1
    uint32_t *items_out_count = q.counters + RT_CNT_SPLIT(par, round + 1);
    uint32_t iter = 0;                                     // wave iterations of the walk loop so far (uniform)
    // the wave passes the top of its outer loop (where long lanes are looked for) at least every period + 1 iterations
    const uint32_t period = (budget && budget < 1024u) ? budget - 1u : 1023u;  // budget is a power of two
    bool can_export = true;                                // uniform: cleared when the round's item buffer is full
    bool occl = false;                                     // SPLIT: the finished item found a hit
    RtTraversalStats st{0, 0};
    uint32_t pool_next = 0, pool_end = 0, pool_base = 0;  // wave-uniform
    constexpr bool SORTED = RT_SORT_BATCH && !ANY && MODE == RT_MODE_QUEUE && batch_max <= 256u;  // rtSortBatch
    bool sorted = false;                                   // wave-uniform: the current batch has an order in LDS
    uint32_t part_seen = 0;                                // how far the current part had got at this wave's last fetch
#ifndef RT_TAIL_FACTOR
#define RT_TAIL_FACTOR 1u
#endif
    const uint32_t tail_len = RT_TAIL_FACTOR * (gridDim.x * (RT_BLOCK / 64) / PARTS * batch);  // RT_TAIL_FACTOR full batches per wave of the part
    bool more = true;                      // wave-uniform: the queue may still hold unclaimed rays
    bool has = false;
    uint32_t ray_i = 0, sample = 0;
    RtWalk w;
    w.r = RtRay{v3s(0.0f), v3s(0.0f)};
    w.inv = w.oi = w.err = w.oiN = w.oiF = v3s(0.0f);
    w.best_t = INFINITY;
    w.best_id = RT_ID_MISS;
    w.best_obj = 0x7FFFFFFF;
    w.best_rank = 0;
    w.cur = w.pend = RT_NODE_DONE;
    w.sp = 0;
    // A finished lane parks its result in registers (fin) and writes it when the wave refills, together with the
    // other lanes that finished meanwhile: written right away, the result code (a load, three fixed-point
    // conversions and atomics) would run once per ray with one or two lanes active.
    bool fin = false;
    uint32_t walk = 0;  // STATS: steps of the lane's current ray (the longest walk bounds a launch's tail)
    uint32_t unocc = 0; // STATS: shadow rays of this lane that reached their light
    // Occluder cache (any-hit queue pass): "occluded" only asks whether ANY triangle lies in front of the light, so the
    // order of the tests is free.  Neighbouring shadow rays (a slab holds rays from neighbouring hits towards one light)
    // are mostly blocked by the same triangle: a new ray starts with the last occluder found in this wave as its
    // postponed leaf, i.e. it is tested in the ray's first triangle phase.  A hit ends the walk before it began; a miss
    // costs one triangle step (4K frame: k_any 282 -> 259 ms).  The cached test skips the box tests on the way to the
    // triangle, so it is only as good as "boxes never cull what the exact test accepts": see the guard where the ray
    // is taken.  -DRT_DEBUG_OCC builds a checker that ignores every cached hit, walks on and prints the rays whose walk
    // finds no occluder (scripts/gpu_float_determinism.py compares float bits across runs); RT_OCCLUDER_CACHE=0
    // (rt_commit) turns the cache off.
    uint32_t last_occ = RT_NODE_DONE;  // single-triangle leaf reference
#ifdef RT_DEBUG_OCC
    uint32_t dbg_ref = RT_NODE_DONE;   // the cached reference this ray started on
    bool dbg_direct = false;           // ... and it hit, but the walk goes on to see whether it agrees
#endif
    bool occ_new = false;              // found since the wave's last refill
    constexpr bool l0store = ANY && L0;
    auto writeResult = [&](uint32_t ri, uint32_t smp, float4 C, uint32_t lightw) {
        bool add = true;
        if (SPLIT) {  // the item that brings the ray's outstanding count to zero speaks for the ray
            uint32_t *join = reinterpret_cast<uint32_t *>(q.shC + ri) + 3;
            if (occl) atomicOr(join, RT_JOIN_OCCLUDED);
            const uint32_t old = atomicSub(join, 1u);
            add = !occl && old == 1u;  // (old carries the flag if any earlier item set it)
            occl = false;
        }
        if (COUNT && ANY && add) unocc++;
        if (ANY && add && l0store && (smp & RT_SHADOW_L0)) {  // (level-0 hits of this iteration: ring samples from prim_sample0)
            const size_t at = (size_t)lightw * q.l0_stride + ((smp & q.sample_mask) - q.l0_sample0);
            q.col0[at] = C;
            q.occ0[at] = 1;
        } else if (ANY && add) {  // (summing runs of equal samples first, as k_shade does, finds too few runs here to pay)
            unsigned long long *acc = reinterpret_cast<unsigned long long *>(q.accum) + 3ull * (smp & q.sample_mask);
            atomicAdd(acc + 0, (unsigned long long)rtToFixed(C.x));
            atomicAdd(acc + 1, (unsigned long long)rtToFixed(C.y));
            atomicAdd(acc + 2, (unsigned long long)rtToFixed(C.z));
        }
        if (!ANY) q.hits[ri] = make_float2(w.best_id == RT_ID_MISS ? INFINITY : w.best_t, __uint_as_float(w.best_id));
        fin = false;
    };
    // STATS build only: where a wave's time goes (shader clock): [0] refill cycles, [1] node-phase cycles, [2] node
    // phases, [3] lanes active in them, [4] triangle-phase cycles, [5] triangle phases, [6] lanes, [7] wave lifetime,
    // [8] refills, [9] rays fetched
    unsigned long long prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tStart = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
    for (;;) {
        const unsigned long long tRefill = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
        if (ANY && budget && can_export) {  // uniform
            const bool longLane = has && iter - age[laneNow()] > budget;
            if (__ballot(longLane)) {
                const uint32_t lane = laneId();
                const uint32_t k = longLane ? (w.cur != RT_NODE_DONE ? 1u : 0u) + (w.pend != RT_NODE_DONE ? 1u : 0u) + (uint32_t)w.sp : 0u;
                uint32_t incl = k;  // inclusive scan over the wave (rare path)
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t t = __shfl_up(incl, d);
                    if (lane >= (uint32_t)d) incl += t;
                }
                const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
                const uint32_t base = waveReserve(items_out_count, total);
                if (base + total <= q.split_cap) {
                    if (longLane) {
                        uint2 *out = items_out + base + (incl - k);
                        if (w.cur != RT_NODE_DONE) *out++ = make_uint2(ray_i, w.cur);
                        if (w.pend != RT_NODE_DONE) *out++ = make_uint2(ray_i, w.pend);
                        while (w.sp) *out++ = make_uint2(ray_i, rtPop(stk, w.sp));
                        // a ray of the queue pass had no join count yet; an item was itself one outstanding unit
                        atomicAdd(reinterpret_cast<uint32_t *>(q.shC + ray_i) + 3, SPLIT ? k - 1u : k);
                        w.cur = w.pend = RT_NODE_DONE;
                        has = false;
                    }
                } else {  // no room: the rays walk on; the reserved slots inside the buffer must not hold garbage
                    for (uint32_t sl = base + lane; sl < min(base + total, q.split_cap); sl += 64) items_out[sl] = make_uint2(RT_SPLIT_NULL, RT_NODE_DONE);
                    // The item buffer of this round is full and stays full: this wave stops exporting, so it bumps
                    // the counter past the capacity at most once (retrying every `budget` iterations could wrap it)
                    can_export = false;
                }
            }
        }
        if (COUNT && !has && walk) {
            atomicMax(q.counters + (ANY ? RT_CNT_ANY_NODES : RT_CNT_NODES) + 4, walk);
            walk = 0;
        }
        const unsigned long long idle2 = __ballot(!has);
        if (idle2 && more) {
            while (pool_next >= pool_end && parts_done < PARTS) {
                const uint32_t lo = first + part * part_len, hi = min(lo + part_len, n);
                // (two batch sizes: one generation of rays once the part is nearly used up -- as far as this wave knows from
                // its previous fetch -- so that a launch does not end with a few waves walking 256 rays each)
                const uint32_t want = TAIL && lo + part_seen + tail_len >= hi ? 64u : batch;
                uint32_t base = 0;
                if (laneId() == 0) base = atomicAdd(PARTS > 1 ? work_parts + part * RT_CNT_STRIDE : work, want);
                base = __builtin_amdgcn_readfirstlane(base);
                if (lo + base < hi && lo < n) {
                    pool_next = lo + base;
                    pool_end = min(pool_next + want, hi);
                    pool_base = pool_next;
                    part_seen = base + want;
                    if (SORTED) {
                        sorted = pool_end - pool_base > 64u;  // (one generation of rays is taken as it comes)
                        if (sorted) rtSortBatch(q, buf, pool_base, pool_end - pool_base, tl.age, tl.aux);
                    }
                } else {  // this part is used up: on to the next
                    part = part + 1 == PARTS ? 0u : part + 1;
                    parts_done++;
                    part_seen = 0;
                }
            }
            if (parts_done >= PARTS) more = false;
            if (ANY && !SPLIT) {  // idle lanes adopt the freshest occluder of the wave (their own is ~100 queue entries old)
                const unsigned long long fresh = __ballot(occ_new);
                if (fresh) {
                    const uint32_t cand = __builtin_amdgcn_readlane(last_occ, 63 - __builtin_clzll(fresh));
                    if (!has) last_occ = cand;
                }
                occ_new = false;
            }
            const uint32_t avail = pool_end - pool_next, want = (uint32_t)__popcll(idle2);
            const uint32_t rank = maskRankBelow(idle2);
            // loads first (the new ray's record and, for a finished shadow ray, its pending colour): one round trip
            const bool take = !has && rank < avail;
            const uint32_t old_i = ray_i, old_sample = sample;
            float4 A = make_float4(0, 0, 0, 0), B = A, C = A;
            uint32_t item_ref = RT_NODE_DONE, light = 0;
            bool nothing = false;  // the slot holds no ray
            if (take) {
                if (SPLIT) {
                    const uint2 it = items_in[pool_next + rank];
                    ray_i = it.x;
                    item_ref = it.y;
                    nothing = ray_i == RT_SPLIT_NULL;
                } else {
                    uint32_t pos = pool_next + rank;
                    // Bounce rays sit in the queue as reflection / refraction PAIRS (k_shade keeps siblings adjacent), and
                    // the two kinds walk very differently: a refraction ray dives into the object it entered (short), a
                    // reflection ray crosses the scene (long).  A wave that takes them alternately always holds a mix, so
                    // its short rays drain and the long ones run in a half-empty wave.  The batch is therefore consumed
                    // even entries first, then the odd ones (order is free: results go to the ray's own slot): 39 -> 42 of
                    // 64 lanes per node phase, k_closest 179 -> 161 ms on the 4K frame.  (Unpaired rays shift the parity;
                    // that only weakens the separation.)
                    if (SORTED) {  // grouped by kind and direction octant (rtSortBatch)
                        if (sorted) pos = pool_base + ((const RtLdsByte *)tl.age)[pos - pool_base];
                    } else if (!ANY) {
                        const uint32_t off = pos - pool_base, len = pool_end - pool_base, half = (len + 1u) >> 1;
                        pos = pool_base + (off < half ? 2u * off : 2u * (off - half) + 1u);
                    }
                    ray_i = pos;
                }
                if (!nothing) A = ANY ? q.shA[ray_i] : q.rayA[buf][ray_i];
                if (!ANY && prim_here && ray_i < q.prim_n) A = q.primA;  // (wave-uniform prim_here: only where primary rays walk per lane)
                if (ANY && !nothing) light = q.shSlabLight[ray_i >> q.sh_slab_shift];  // (a slab holds rays towards ONE light)
                if (!ANY) B = q.rayB[buf][ray_i];
                if (ANY && !SPLIT) nothing = __float_as_uint(A.w) == RT_SHADOW_HOLE;  // unused slot of a k_shade slab
            }
            uint32_t old_light = 0;
            if (ANY && fin && !occl) C = q.shC[old_i];  // (an occluded ray adds nothing: 60 % of the competition frame's shadow rays)
            if (ANY && fin && l0store) old_light = q.shSlabLight[old_i >> q.sh_slab_shift];
            if (fin) writeResult(old_i, old_sample, C, old_light);
            if (take && !nothing) {
                float ldist = 0.0f;
                if (ANY) {
                    rtShadowRay(sc, A, light, w.r, ldist, sample);
                } else {
                    w.r.o = v3(A.x, A.y, A.z);
                    w.r.d = v3(B.x, B.y, B.z);
                    sample = __float_as_uint(B.w);
                }
                rtInitWalk(sc, w, root_ref);
                has = true;
                if (ANY) age[laneNow()] = iter;
                if (SPLIT) {
                    w.best_t = ldist;
                    w.cur = item_ref;  // the subtree (or leaf) this item stands for; the analytic primitives were tested in the queue pass
                } else if (ANY) {
                    w.best_t = ldist;  // light distance
                    if (analyticOccluded(sc, an, w, ldist)) has = false;  // occluded: nothing to add
                    // Not for rays that start far outside the scene (a horizon pixel on an infinite plane): there
                    // the exact test's distance t is off by up to ~10^3 units for a triangle the ray grazes (t is a
                    // quotient of two rounded dot products of |o|-sized terms), so it accepts triangles BEYOND the
                    // light that the walk, whose boxes end at the light, never reaches -- the answer would depend on
                    // what the cache holds (RT_DEBUG_OCC build: 7 such rays per stress frame, none with this guard).
                    const bool near_scene = fmaxf(fmaxf(fabsf(w.r.o.x), fabsf(w.r.o.y)), fabsf(w.r.o.z)) <= 4.0f * sc.bound;
                    if (sc.occluder_cache && near_scene) w.pend = last_occ;
#ifdef RT_DEBUG_OCC
                    dbg_ref = w.pend;
                    dbg_direct = false;
#endif
                } else if (sample & RT_FLAG_INVALID) {  // slot of a pixel outside the image: recorded as a miss
                    w.best_t = INFINITY;
                    w.best_id = RT_ID_MISS;
                    w.cur = RT_NODE_DONE;
                } else {
                    analyticClosest(sc, an, w);
                }
            }
            pool_next += min(want, avail);
            if (PROF) prof[0] += __builtin_amdgcn_s_memtime() - tRefill;
            if (COUNT) {  // (uniform counters: scalar adds)
                prof[8]++;
                prof[9] += min(want, avail);
            }
        }
        if (!__ballot(has)) {
            if (!more) break;
            continue;
        }
        const int leave_below = more ? RT_REFILL_BELOW : 1;
        // Phases: a lane that reaches a leaf postpones it (pend) and goes on with its next node; each iteration the
        // wave runs either a node step or a triangle step, whichever more lanes are ready for, so both bodies run
        // with most lanes active instead of alternating with partial masks.
        do {
            if (has) rtPostponeLeaf(w, stk);
            const bool atNode = has && !(w.cur & RT_NODE_LEAF), atTri = has && w.pend != RT_NODE_DONE;
            const int nNode = (int)__popcll(__ballot(atNode)), nTri = (int)__popcll(__ballot(atTri));
            const unsigned long long tPhase = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
#ifndef RT_VOTE_NODE_NUM  // node phase iff nNode * NUM >= nTri * DEN (1/1 = the larger group)
// (closest-hit walks: a triangle step is what shortens the ray and lets the next node steps cull, and since the large triangles
// are cut into pieces there are few of them per ray -- they go first unless twice as many lanes wait at a node: 1/1 387.2 ms per
// 4K frame, 3/4 384.9, 2/3 385.2, 1/2 383.9, 2/5 and 1/3 the same)
#define RT_VOTE_NODE_NUM 1
#define RT_VOTE_NODE_DEN 2
#endif
#ifndef RT_VOTE_ANY_NUM  // any-hit: a new ray's first triangle step is its cached occluder, so triangle phases get a head start
#define RT_VOTE_ANY_NUM 2   // (k_any on the 4K frame: 1/1 260 ms, 2/3 255, 1/2 257, 1/3 260, 1/4 266)
#define RT_VOTE_ANY_DEN 3
#endif
            if (nNode * (ANY ? RT_VOTE_ANY_NUM : RT_VOTE_NODE_NUM) >= nTri * (ANY ? RT_VOTE_ANY_DEN : RT_VOTE_NODE_DEN) && nNode > 0) {
                if (atNode) rtNodeStep<ANY, COUNT>(sc, w, stk, st);
                if (COUNT && atNode) walk++;
                if (PROF) prof[1] += __builtin_amdgcn_s_memtime() - tPhase;
                if (COUNT) {
                    prof[2]++;
                    prof[3] += nNode;
                }
            } else if (nTri > 0) {
                if (COUNT && atTri) walk++;
#ifdef RT_DEBUG_OCC
                bool dbg_first = ANY && !SPLIT && atTri && dbg_ref != RT_NODE_DONE && w.pend == dbg_ref;
                if (dbg_first) dbg_ref = RT_NODE_DONE;
                bool dbg_hit = atTri && rtTriStep<ANY, COUNT>(sc, w, st);
                if (dbg_hit && dbg_first) {  // ignore the direct hit, remember it, walk on
                    dbg_direct = true;
                    dbg_hit = false;
                    w.pend = RT_NODE_DONE;
                }
                if (dbg_hit) {
                    has = false;
#else
                if (atTri && rtTriStep<ANY, COUNT>(sc, w, st)) {  // occluded: the pending radiance is dropped
                    has = false;
#endif
                    if (ANY && !SPLIT) {
                        last_occ = w.pend & ~(7u << 28);  // the triangle alone, not the rest of its leaf
                        occ_new = true;
                    }
                    if (SPLIT) {  // ... by the ray's last item (writeResult)
                        fin = true;
                        occl = true;
                    }
                }
                if (PROF) prof[4] += __builtin_amdgcn_s_memtime() - tPhase;
                if (COUNT) {
                    prof[5]++;
                    prof[6] += nTri;
                }
            }
            if (has && w.cur == RT_NODE_DONE && w.pend == RT_NODE_DONE) {  // walk finished: result written at the next refill
                fin = true;
                has = false;
#ifdef RT_DEBUG_OCC
                if (ANY && !SPLIT && dbg_direct)
                    printf("OCCHOLE ray %u tri %u o %.9g %.9g %.9g d %.9g %.9g %.9g ldist %.9g\n", ray_i, last_occ & RT_NODE_INDEX_MASK,
                           w.r.o.x, w.r.o.y, w.r.o.z, w.r.d.x, w.r.d.y, w.r.d.z, w.best_t);
#endif
            }
            ++iter;
        } while ((int)__popcll(__ballot(has)) >= leave_below && (iter & period) != 0);
    }
    if (fin) {
        const float4 Cl = (ANY && !occl) ? q.shC[ray_i] : make_float4(0, 0, 0, 0);
        writeResult(ray_i, sample, Cl, l0store ? (uint32_t)q.shSlabLight[ray_i >> q.sh_slab_shift] : 0u);
    }
    if (COUNT) {
        if (walk) atomicMax(q.counters + (ANY ? RT_CNT_ANY_NODES : RT_CNT_NODES) + 4, walk);
        flushStats(q.counters, st, ANY ? RT_CNT_ANY_NODES : RT_CNT_NODES);
        if (ANY) {
            unsigned long long u = unocc;
            for (int off = 32; off > 0; off >>= 1) u += __shfl_down(u, off);
            if (laneId() == 0 && u) atomicAdd(reinterpret_cast<unsigned long long *>(q.counters + RT_CNT_UNOCCLUDED), u);
        }
        if (PROF) prof[7] = __builtin_amdgcn_s_memtime() - tStart;
        if (laneId() == 0) {  // counts of phases / lanes / refills / rays with COUNT, their clocks with PROF
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(q.counters + (ANY ? RT_CNT_PROF_ANY : RT_CNT_PROF_CLOSEST));
            for (int k = 0; k < 10; k++)
                if (prof[k]) atomicAdd(dst + k, prof[k]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Packet kernels: 64 queue-consecutive rays (one wave) walk the tree TOGETHER.  At 64 spp a wave's primary rays are
// the 64 samples of one pixel, its shadow rays leave neighbouring points towards one light: such rays visit nearly
// the same nodes, so the walk is made wave-uniform -- the node reference, the stack (one VGPR, entry k in lane k) and
// the order decisions live in scalar registers, nodes and triangles arrive through scalar loads (no L1 traffic, no
// per-lane stack), and every lane tests the same four boxes / the same triangle with all 64 lanes active.  A child is
// entered when ANY lane's box test passes, so results are those of the per-lane walk (boxes only cull; the exact
// triangle test decides).  Packets whose directions do not share one octant, or that exceed a step budget
// (incoherent rays: the union of their walks is large), are handed to the per-lane kernels through q.resid.
#ifndef RT_PACKET_BATCH
#define RT_PACKET_BATCH 8u       // packets a wave reserves per atomic (2: +18 ms, the cursor saturates; 4, 8, 16 equal)
#endif
#ifndef RT_PACKET_BUDGET
#define RT_PACKET_BUDGET 160     // node steps after which a packet is declared incoherent
#endif

typedef uint32_t u16v __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) u16v *RtScalarRecs;  // constant address space: uniform loads become s_load

template <bool STATS>
RT_DEV void tracePacket(const RtSceneDev &sc, const RtTraceLds &tl, uint32_t root_ref, const RtQueues &q, int par,
                        uint32_t pb, uint32_t n, unsigned long long *prof) {
    const RtAnalytic &an = tl.an;
    const uint32_t lane = laneId();
    const int buf = par;
    const uint32_t i = pb + lane;
    const bool valid = i < n;
    RtWalk w;
    float4 A = make_float4(0.0f, 0.0f, 0.0f, 0.0f), B = make_float4(0.0f, 0.0f, 1.0f, __uint_as_float(RT_FLAG_INVALID));
    if (valid) {
        A = q.prim_implied ? q.primA : q.rayA[buf][i];  // (packets are primary rays: i < prim_n)
        B = q.rayB[buf][i];
    }
    const uint32_t sample = __float_as_uint(B.w);
    w.r.o = v3(A.x, A.y, A.z);
    w.r.d = v3(B.x, B.y, B.z);
    rtInitWalk(sc, w, root_ref);
    bool alive = valid;
    w.best_t = INFINITY;
    w.best_id = RT_ID_MISS;
    w.best_obj = 0x7FFFFFFF;
    w.best_rank = 0;
    if (sample & RT_FLAG_INVALID) alive = false;  // slot of a pixel outside the image: recorded as a miss
    if (alive) {
        if (q.prim_far) analyticClosestT<false>(sc, an, w);  // (uniform)
        else analyticClosestT<true>(sc, an, w);
    }
    const unsigned long long am = __ballot(alive);
    bool toResid = false;
    // error bound of a plane distance folded into o/d: near planes move towards the origin, far planes away
    const V3 oiN = w.oi + w.err, oiF = w.oi - w.err;
    const uint32_t octant = (w.inv.x < 0.0f ? 1u : 0u) | (w.inv.y < 0.0f ? 2u : 0u) | (w.inv.z < 0.0f ? 4u : 0u);
    const RtScalarRecs nodesP = (RtScalarRecs)(const void *)sc.nodes_p;
    const RtScalarRecs recs = (RtScalarRecs)(const void *)sc.nodes;
    int steps = 0;
    // One walk per direction octant present in the packet (usually one): inside a group near / far planes are chosen
    // once, on the scalar unit.
    for (unsigned long long rem = sc.n_tris ? am : 0ull; rem && !toResid;) {
        const uint32_t oct = __builtin_amdgcn_readlane(octant, (int)__builtin_ctzll(rem));
        const bool act = alive && octant == oct;
        rem &= ~__ballot(octant == oct);
        if (STATS) prof[6]++;
        const bool negx = oct & 1u, negy = oct & 2u, negz = oct & 4u;
        uint32_t cur = __builtin_amdgcn_readfirstlane(root_ref), vstack = 0;
        int sp = 0;
        while (cur != RT_NODE_DONE) {
            if (!(cur & RT_NODE_LEAF)) {
                if (++steps > RT_PACKET_BUDGET) {
                    toResid = true;
                    if (STATS) prof[5]++;
                    break;
                }
                const u16v n0 = nodesP[2 * (size_t)cur], n1 = nodesP[2 * (size_t)cur + 1];
                unsigned long long m[4];
                float tnv[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float lx = __uint_as_float(n0[c]), ly = __uint_as_float(n0[4 + c]), lz = __uint_as_float(n0[8 + c]);
                    const float hx = __uint_as_float(n0[12 + c]), hy = __uint_as_float(n1[c]), hz = __uint_as_float(n1[4 + c]);
                    const float nx = negx ? hx : lx, fx = negx ? lx : hx;  // uniform selects
                    const float ny = negy ? hy : ly, fy = negy ? ly : hy;
                    const float nz = negz ? hz : lz, fz = negz ? lz : hz;
                    const float tnx = __builtin_fmaf(nx, w.inv.x, -oiN.x), tfx = __builtin_fmaf(fx, w.inv.x, -oiF.x);
                    const float tny = __builtin_fmaf(ny, w.inv.y, -oiN.y), tfy = __builtin_fmaf(fy, w.inv.y, -oiF.y);
                    const float tnz = __builtin_fmaf(nz, w.inv.z, -oiN.z), tfz = __builtin_fmaf(fz, w.inv.z, -oiF.z);
                    const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
                    const float tf = fminf(fminf(tfx, tfy), fminf(tfz, w.best_t)) * 1.000001f;
                    tnv[c] = tn;
                    m[c] = n1[8 + c] != 0xFFFFFFFFu ? __ballot(act && tn <= tf) : 0ull;
                }
                if (STATS) {
                    prof[1]++;
                    prof[2] += (unsigned long long)__popcll(m[0] | m[1] | m[2] | m[3]);
                }
                // order: entry distance of each child's first interested lane (scalar sort of four pairs)
                uint32_t key[4], ref[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    ref[c] = n1[8 + c];
                    key[c] = 0xFFFFFFFFu;
                    if (m[c]) key[c] = __builtin_amdgcn_readlane(__float_as_uint(tnv[c]), (int)__builtin_ctzll(m[c]));
                }
#define RT_CSWAP(a, b)                                                  \
    if (key[b] < key[a]) {                                              \
        const uint32_t tk = key[a], tr = ref[a];                        \
        key[a] = key[b]; ref[a] = ref[b]; key[b] = tk; ref[b] = tr;     \
    }
                RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
#undef RT_CSWAP
                if (key[0] == 0xFFFFFFFFu) {
                    cur = sp ? __builtin_amdgcn_readlane(vstack, --sp) : RT_NODE_DONE;
                } else {
#pragma unroll
                    for (int c = 3; c >= 1; c--)
                        if (key[c] != 0xFFFFFFFFu) {
                            vstack = ((int)lane == sp) ? ref[c] : vstack;
                            sp++;
                        }
                    cur = ref[0];
                }
            } else {  // leaf: every live lane of the group runs the exact test on each of its triangles
                const uint32_t first = cur & RT_NODE_INDEX_MASK, cnt = ((cur >> 28) & 7u) + 1u;
                for (uint32_t k = 0; k < cnt; k++) {
                    const u16v tr = recs[first + k];
                    if (STATS) {
                        prof[3]++;
                        prof[4] += (unsigned long long)__popcll(__ballot(act));
                    }
                    RtTri T;
                    T.A = v3(__uint_as_float(tr[0]), __uint_as_float(tr[1]), __uint_as_float(tr[2]));
                    T.B = v3(__uint_as_float(tr[4]), __uint_as_float(tr[5]), __uint_as_float(tr[6]));
                    T.C = v3(__uint_as_float(tr[8]), __uint_as_float(tr[9]), __uint_as_float(tr[10]));
                    T.n = v3(__uint_as_float(tr[3]), __uint_as_float(tr[7]), __uint_as_float(tr[11]));
                    float t = 0.0f;
                    const bool hit = act && rtTriHit(T, w.r, w.best_t, t);
                    // (selects after the test, as in rtTriStep)
                    const int obj = (int)tr[12];
                    const uint32_t rank = tr[13];
                    const bool take = hit && (t < w.best_t || (t == w.best_t && w.best_id != RT_ID_MISS &&
                                                               (obj < w.best_obj || (obj == w.best_obj && rank < w.best_rank))));
                    w.best_t = take ? t : w.best_t;
                    w.best_id = take ? ((first + k - (uint32_t)sc.n_nodes) | tr[15]) : w.best_id;  // (| RT_ID_GENERAL flag)
                    w.best_obj = take ? obj : w.best_obj;
                    w.best_rank = take ? rank : w.best_rank;
                }
                cur = sp ? __builtin_amdgcn_readlane(vstack, --sp) : RT_NODE_DONE;
            }
        }
    }
    if (toResid) {
        // Incoherent packet: its rays finish as per-lane walks right here, from the root, with the best hit found so far
        // (re-testing a triangle gives the same distance, and the tie rule never replaces a hit by itself).  A few per
        // cent of the packets; handing them to a follow-up launch cost a launch tail per iteration.
        if (STATS) prof[7] += (unsigned long long)__popcll(am);
        bool has = alive;
        w.cur = has ? root_ref : RT_NODE_DONE;
        w.pend = RT_NODE_DONE;
        w.sp = 0;
        RtTraversalStats st{0, 0};
        while (__ballot(has)) {
            if (has) rtPostponeLeaf(w, tl.stk);
            const bool atNode = has && !(w.cur & RT_NODE_LEAF), atTri = has && w.pend != RT_NODE_DONE;
            const int nNode = (int)__popcll(__ballot(atNode)), nTri = (int)__popcll(__ballot(atTri));
            if (nNode >= nTri && nNode > 0) {
                if (atNode) rtNodeStep<false, false>(sc, w, tl.stk, st);
            } else if (nTri > 0) {
                if (atTri) rtTriStep<false, false>(sc, w, st);
            }
            if (has && w.cur == RT_NODE_DONE && w.pend == RT_NODE_DONE) has = false;
        }
    }
    if (valid) q.hits[i] = make_float2(w.best_id == RT_ID_MISS ? INFINITY : w.best_t, __uint_as_float(w.best_id));
}

// the primary rays of the entering chunk: entries [0, q.prim_n) of ray queue `par`, 64 at a time
template <bool STATS>
RT_DEV void packetQueue(const RtSceneDev &sc, uint32_t root_ref, const RtQueues &q, const RtTraceLds &tl, int par) {
    const uint32_t n = min(q.prim_n, q.ray_cap);
    // STATS: [0] packets, [1] node steps, [2] lanes interested in a child per node step, [3] triangle steps, [4] live
    // lanes per triangle step, [5] packets over budget, [6] octant groups walked, [7] rays that finished as per-lane walks
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // Guided batches: a wave walks the packets of its batch one after the other, so large last batches would keep a few
    // waves busy while the chip idles: a wave takes up to RT_PACKET_BATCH packets per fetch, fewer once its part of the
    // queue runs out (half of what is left per wave of the part).
    RtPartCursor pc = rtPartCursor(q.counters + RT_CNT_WORKP_PACKET(par, 0), 0u, n, 64u * RT_PACKET_BATCH);
    const uint32_t waves_per_part = max(1u, gridDim.x * (RT_BLOCK / 64) / RT_WORK_PARTS);
    for (;;) {
        const uint32_t want = 64u * min((uint32_t)RT_PACKET_BATCH, max(1u, rtPartLeft(pc) / (128u * waves_per_part)));
        uint32_t base, end;
        if (!rtPartFetch(pc, want, base, end)) break;
        for (uint32_t pb = base; pb < end; pb += 64u) {
            if (STATS) prof[0]++;
            tracePacket<STATS>(sc, tl, root_ref, q, par, pb, n, prof);
        }
    }
    if (STATS && laneId() == 0) {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(q.counters + RT_CNT_PROF_PACKET);
        for (int k = 0; k < 8; k++) atomicAdd(dst + k, prof[k]);
    }
}

// The traversal kernels.  All of them use the same LDS layout (RtTraceLds) and 72 VGPRs (RT_TRACE_MIN_BLOCKS = 7 waves per SIMD).
// k_trace is the FUSED form the frame loop launches once per iteration: its waves work off, one phase after the other
// and each wave at its own pace, (1) the shadow queue of iteration j, (2) the bounce rays of iteration j + 1 (appended by
// k_shade(j), which ran before this launch), (3) the primary rays of the chunk that enters at j + 1, as packets.  The
// three are independent of each other, and a persistent launch ends with its longest walk: separate launches paid
// that tail three times per iteration (~0.3 ms each), here the tails of (1) and (2) are filled by the next phase.
// par_any / par_closest < 0: the phase has no work (first and last iterations).
template <int STATS, bool L0>
__global__ __launch_bounds__(RT_BLOCK, RT_TRACE_MIN_BLOCKS) void k_trace(RtSceneDev sc, uint32_t root_ref, RtQueues q, int par_any, int par_closest, uint32_t first,
                                                       int packets, int stack_entries, uint32_t budget) {
    extern __shared__ uint32_t s_lds[];
    const unsigned long long tp0 = rtTailBegin(q);
    const RtTraceLds tl = rtTraceSetup(sc, q, s_lds, stack_entries);
#ifndef RT_FUSED_TAIL   // (1: the phases of the fused launch keep their one-generation fetches at the end of their queues)
#define RT_FUSED_TAIL 0
#endif
    if (par_any >= 0) {
        if (par_closest >= 0 && !RT_FUSED_TAIL) traceQueue<true, STATS, RT_MODE_QUEUE, L0, false>(sc, root_ref, q, tl, par_any, 0u, -1, budget);
        else traceQueue<true, STATS, RT_MODE_QUEUE, L0, true>(sc, root_ref, q, tl, par_any, 0u, -1, budget);
    }
    rtTailMark(q, 0);
    if (par_closest >= 0) {
        if (packets && !RT_FUSED_TAIL) traceQueue<false, STATS, RT_MODE_QUEUE, false, false>(sc, root_ref, q, tl, par_closest, first, -1, 0u);
        else traceQueue<false, STATS, RT_MODE_QUEUE, false, true>(sc, root_ref, q, tl, par_closest, first, -1, 0u);
        rtTailMark(q, 1);
        if (packets) packetQueue<(STATS != 0)>(sc, root_ref, q, tl, par_closest);
    }
    rtTailEnd(q, tp0);
}
// the phases as launches of their own (per-kernel timing: collect_stats & 2, RT_FUSE=0; rt_trace_batch; the split rounds)
template <bool STATS>
__global__ __launch_bounds__(RT_BLOCK, RT_TRACE_MIN_BLOCKS) void k_packet(RtSceneDev sc, uint32_t root_ref, RtQueues q, int par, int stack_entries) {
    extern __shared__ uint32_t s_lds[];
    const RtTraceLds tl = rtTraceSetup(sc, q, s_lds, stack_entries);
    packetQueue<STATS>(sc, root_ref, q, tl, par);
}
template <int STATS>
__global__ __launch_bounds__(RT_BLOCK, RT_TRACE_MIN_BLOCKS) void k_closest(RtSceneDev sc, uint32_t root_ref, RtQueues q, int par, uint32_t first, int stack_entries) {
    extern __shared__ uint32_t s_lds[];
    const RtTraceLds tl = rtTraceSetup(sc, q, s_lds, stack_entries);
    traceQueue<false, STATS, RT_MODE_QUEUE>(sc, root_ref, q, tl, par, first, -1, 0u);
}
template <int STATS, int MODE, bool L0 = false>
__global__ __launch_bounds__(RT_BLOCK, RT_TRACE_MIN_BLOCKS) void k_any(RtSceneDev sc, uint32_t root_ref, RtQueues q, int par, int stack_entries, int round,
                                                     uint32_t budget) {
    if (MODE == RT_MODE_SPLIT && q.counters[RT_CNT_SPLIT(par, round)] == 0) return;  // the usual case: nothing was split
    extern __shared__ uint32_t s_lds[];
    const RtTraceLds tl = rtTraceSetup(sc, q, s_lds, stack_entries);
    traceQueue<true, STATS, MODE, L0>(sc, root_ref, q, tl, par, 0u, round, budget);
}

// ---------------------------------------------------------------------------------------------------------------
// PhongModel without its nested closest()/is_shadowed() calls (main.cpp:81-213).  One lane per traced ray of the
// iteration's queue (rays of every level, the level rides in the ray's tag); remaining bounce budget = max_depth - level.  A node's colour is linear in its children except for the
// Fresnel coupling (main.cpp:187-203), which is resolved here from the sibling's hit record.
// Colour-only math of PhongModel (its results feed radiance, never a ray): the reference's libm calls are not bit-exact
// on the device anyway (BASELINE tolerance: 8-bit RGB within 1, PSNR >= 50 dB), so these use the hardware's
// transcendental units instead of ocml's full-accuracy sequences.  Geometry (directions, distances, hit tests) keeps
// IEEE division and square root.
#ifndef RT_LIBM_COLOR_MATH  // (define it to get ocml's powf / cosf(acosf) / the double division back: +18 ms per frame, same 8-bit images)
// std::pow(x, y) as main.cpp:142 uses it, followed by max(0, .): negative bases give NaN (-> 0) unless y is an integer
RT_DEV float colorPowMax0(float x, float y) {
    if (y == 0.0f) return 1.0f;
    const float ax = fabsf(x);
    float r = __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(ax));  // v_exp_f32(y * v_log_f32(|x|))
    if (x < 0.0f) {
        const float fy = floorf(y);
        const bool isInt = fy == y, odd = isInt && (fy * 0.5f != floorf(fy * 0.5f));
        if (!isInt || odd) r = 0.0f;  // NaN or a negative value: max(0, .) is 0
    }
    return r == r ? r : 0.0f;
}
#define RT_COLOR_POW_MAX0(x, y) colorPowMax0((x), (y))
#define RT_COLOR_ATTENUATION(d) __builtin_amdgcn_rcpf((d) * (d))
#define RT_COS_OF_ANGLE(a, b) gclamp(dot((a), (b)), -1.0f, 1.0f)
#else
#define RT_COLOR_POW_MAX0(x, y) smax(0.0f, powf((x), (y)))
#define RT_COLOR_ATTENUATION(d) ((float)(1.0 / ((double)(d) * (double)(d))))
#define RT_COS_OF_ANGLE(a, b) cosf(rtAngle((a), (b)))
#endif

// Two instances.  FAST (GENERAL = false) scans the level's hit records and shades triangles and planes whose material
// needs neither a BMP map nor the anisotropic branch: without those and without the sphere / cone reconstruction
// (double-precision asin / atan2) the kernel fits 128 VGPRs with no scratch (the full body spills 95 registers).  Hits it
// cannot shade (RT_ID_GENERAL in the id, or a sphere / cone) go to a list (q.general) that the GENERAL instance -- the
// full body -- works off afterwards; it is launched only for scenes that hold such primitives.
template <bool PROF, bool GENERAL>
__global__ __launch_bounds__(RT_BLOCK, RT_SHADE_BLOCKS_PER_CU) void k_shade(RtSceneDev sc, RtQueues q, int par, int max_depth, int spp,
                                                   uint32_t tile0, uint32_t *primid_tiles, uint32_t slab) {
    constexpr bool FAST = !GENERAL;
    const unsigned long long tp0 = rtTailBegin(q);
    // Output slots come from WAVE-PRIVATE slabs: a wave reserves `slab` consecutive slots of the shadow queue per light
    // (and of the next level's ray queue) with one atomic and fills them over its next iterations -- no barrier, no
    // per-iteration atomic, and a slab holds rays towards ONE light, which keeps k_any's batches coherent.  Cursor and
    // end of the slab for light l live in lane l of curv/endv, those of the ray slab in lane RT_MAX_LIGHTS.  What a
    // wave leaves unused at the end is marked as holes (shadow: w = RT_SHADOW_HOLE; rays: RT_FLAG_INVALID), which the
    // trace kernels skip; holes are counted so that the ray statistics stay exact.
    uint32_t curv = 0, endv = 0, holesS = 0, holesR = 0;
    uint32_t gcur = 0, gend = 0;  // the wave's slab of the general list (wave-uniform)
    const uint32_t slabS = slab, slabR = max(slab, 128u);  // one iteration emits at most 64 shadow rays per light, 128 rays
    const int buf = par, nbuf = par ^ 1;
    const uint32_t n = GENERAL ? min(q.counters[RT_CNT_GENERAL(par)], q.general_cap) : min(q.counters[RT_CNT_RAYS(par)], q.ray_cap);
    // A wave shades SEGMENTS of consecutive queue entries (consecutive entries are neighbouring samples / hits), so the
    // rays it appends to a slab leave neighbouring points towards one light.  A wave takes its next segment from a
    // cursor (one atomic per segment): segments cost very differently (misses are only scanned), and dealing them out
    // in fixed order left the launch waiting for the waves that drew the expensive ones.  Their length adapts to the
    // queue so that every wave still gets several.
    const uint32_t nwaves = gridDim.x * (RT_BLOCK / 64);
    const uint32_t seg = min(max((n / (nwaves * 8u)) & ~63u, 64u), RT_SHADE_SEGMENT);
    RtPartCursor segs = rtPartCursor(q.counters + (GENERAL ? RT_CNT_WORKP_SHADEG(par, 0) : RT_CNT_WORKP_SHADE(par, 0)), 0u, n, 64u);
    auto nextSegment = [&](uint32_t &b, uint32_t &e) { return rtPartFetch(segs, seg, b, e); };
    const uint32_t lane = laneId();
    const int nlights = sc.n_lights;
    uint32_t culled = 0, shaded = 0, zero = 0, l0lit = 0;
    // PROF build: shader clocks per section, summed over waves: [0] loads + hit reconstruction, [1] material, light
    // culling, textures, ambient, bounce set-up, [2] the scan for hits, [3] direct light + shadow-ray emission,
    // [4] next-level emission, [5] iterations
    unsigned long long prof[6] = {0, 0, 0, 0, 0, 0}, tp = PROF ? __builtin_amdgcn_s_memtime() : 0ull, tn = 0;
#define RT_PROF_MARK(k) if (PROF) { tn = __builtin_amdgcn_s_memtime(); prof[k] += tn - tp; tp = tn; }
    // Only HITS are shaded, and a third of the traced rays of a frame miss: the wave first SCANS its segment (64 hit
    // records at a time: 8 bytes each), appends the indices of the hits to a ring in LDS, and shades 64 of them at a
    // time with every lane busy.  At level 0 the scan also stores the zero that initialises the accumulator of a
    // sample whose ray missed (see below).  Order is kept, so sibling pairs and runs of equal samples stay neighbours.
    __shared__ uint32_t s_ring[RT_BLOCK / 64][128];
    uint32_t *const ring = s_ring[threadIdx.x >> 6];
    uint32_t scan = 0, segEnd = 0, head = 0, count = 0;  // wave-uniform scan state
    bool moreSeg = nextSegment(scan, segEnd);
    while (moreSeg || count) {
        while (count < 64 && moreSeg) {
            const uint32_t j = scan + lane;
            const bool valid = j < segEnd;
            // FAST: entry j of the level's ray queue; GENERAL: entry j of the list the fast instance left (all hits)
            const uint32_t qi = GENERAL ? (valid ? q.general[j] : RT_GENERAL_HOLE) : j;
            const uint32_t hid = GENERAL ? 0u : __float_as_uint(q.hits[valid ? j : 0].y);
            bool hit = valid && (GENERAL ? qi != RT_GENERAL_HOLE : hid != RT_ID_MISS);
            if (!GENERAL && j < q.prim_n && valid && !hit) {  // primary ray that missed, or slot outside the image: the sample's first and only level-0 entry
                long long *acc = q.accum + 3ull * (q.prim_sample0 + j);
                acc[0] = 0;
                acc[1] = 0;
                acc[2] = 0;
            }
            if (!GENERAL) {  // what this instance cannot shade goes to the general one's list
                const bool gen = hit && ((hid & RT_ID_KIND_MASK) == RT_ID_SPHERE || (hid & RT_ID_GENERAL));
                const unsigned long long gm = __ballot(gen);
                if (gm) {  // wave-uniform.  List slots come from a wave-private slab too: in a scene of spheres nearly every
                           // scan step appends, and one returning atomic per step on ONE counter serialises the chip
                    const uint32_t c = (uint32_t)__popcll(gm), rank = maskRankBelow(gm), room = gend - gcur;
                    uint32_t slot = gcur + rank;
                    if (c > room) {  // the slab is used up to its last slot, the rest opens the next one
                        const uint32_t nbase = waveReserve(q.counters + RT_CNT_GENERAL(par), slabR);
                        if (rank >= room) slot = nbase + (rank - room);
                        gcur = nbase + (c - room);
                        gend = nbase + slabR;
                    } else {
                        gcur += c;
                    }
                    if (gen) {
                        if (slot < q.general_cap) q.general[slot] = j;
                        else atomicAdd(q.counters + RT_CNT_OVERFLOW, 1u);
                        hit = false;
                    }
                }
            }
            const unsigned long long hm = __ballot(hit);
            if (hit) ring[(head + count + maskRankBelow(hm)) & 127u] = qi;
            count += (uint32_t)__popcll(hm);
            scan += 64;
            if (scan >= segEnd) moreSeg = nextSegment(scan, segEnd);  // next segment of this wave
        }
        RT_PROF_MARK(2)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t takeN = min(count, 64u);
        bool active = lane < takeN;
        const uint32_t i = active ? ring[(head + lane) & 127u] : 0u;
        __builtin_amdgcn_wave_barrier();
        head = (head + takeN) & 127u;
        count -= takeN;
        float2 hrec = make_float2(0.0f, __uint_as_float(RT_ID_MISS));
        float4 A = make_float4(0, 0, 0, 0), B = make_float4(0, 0, 0, __uint_as_float(RT_FLAG_INVALID));
        if (active) {
            hrec = q.hits[i];
            A = (q.prim_implied && i < q.prim_n) ? q.primA : q.rayA[buf][i];
            B = q.rayB[buf][i];
        }
        const uint32_t id = __float_as_uint(hrec.y);
        const uint32_t sflags = __float_as_uint(B.w), sample = sflags & q.sample_mask;
        const uint32_t level = (sflags >> q.level_shift) & q.level_mask;   // per ray: the queue mixes the levels of D + 1 chunks
        const int remaining = max_depth - (int)level;
        if (active && level == 0 && primid_tiles && ((sample - q.prim_sample0) % (uint32_t)spp) == 0) {
            // primitive hit by sample 0 of the pixel, written tile-major at y*16+x (misses keep the buffer's RT_PRIM_MISS)
            const uint32_t pm = (sample - q.prim_sample0) / (uint32_t)spp, m = pm & 255u;
            const uint32_t px = compact1By1(m), py = compact1By1(m >> 1);
            const uint32_t kind = id & RT_ID_KIND_MASK;
            uint32_t prim;
            if (kind == RT_ID_SPHERE) prim = RT_PRIM_ANALYTIC | (uint32_t)sc.spheres[id & 0xFFFF].object;
            else if (kind == RT_ID_PLANE) prim = RT_PRIM_ANALYTIC | (uint32_t)sc.planes[id & 0xFFFF].object;
            else prim = sc.tri_shade[id & RT_ID_TRI_MASK].prim;
            primid_tiles[(size_t)(tile0 + (pm >> 8) * q.tile_stride) * RT_TILE_PIXELS + py * RT_TILE + px] = prim;
        }

        // per-lane state carried from the shading step to the (block-uniform) emission steps
        V3 P = v3s(0), ns = v3s(0), view = v3s(0), normal = v3s(0), dcol = v3s(0), mspec = v3s(0), hT = v3s(0), hB = v3s(0);
        float weight = 0.0f, shiny = 0.0f, alpha_x = 1.0f, alpha_y = 1.0f, directScale = 1.0f, directScale2 = 1.0f;
        bool aniso = false;
        bool emitRefl = false, emitRefr = false;
        V3 amb0 = v3s(0.0f);
        V3 reflDir = v3s(0), refrDir = v3s(0);
        float wRefl = 0, wRefr = 0, fresnelR = 1.0f;

        if (active) {
            shaded++;
            weight = A.w;
            if (sflags & RT_FLAG_HAS_REFR_SIBLING) {  // reflection *= R only if the refraction ray hit (main.cpp:187,201)
                const bool sibHit = __float_as_uint(q.hits[i + 1].y) != RT_ID_MISS;
                if (sibHit) weight = weight * q.rayR[buf][i];
            }
            RtRay r{v3(A.x, A.y, A.z), v3(B.x, B.y, B.z)};
            RtFullHit h;
            rtReconstructHit<FAST>(sc, r, hrec.x, id, h);
            RT_PROF_MARK(0)
            const rt_material m = sc.materials[h.material];
            P = h.P;
            hT = h.T;
            hB = h.B;
            view = normalize(-r.d);  // main.cpp:228
            normal = h.normal;
            ns = h.ns;
            if (dot(normal, view) < 0) normal = -normal;  // main.cpp:88-94
            if (dot(ns, view) < 0) ns = -ns;
            mspec = ld3(m.specular);
            aniso = !FAST && m.is_anisotropic != 0;
            alpha_x = m.alpha_x;
            alpha_y = m.alpha_y;
            // (1 - reflection)(1 - refraction) scaling of the direct term (main.cpp:153,172), only with budget left
            const bool bounce = remaining > 0;
            const bool hasRefl = bounce && m.reflection > 0, hasRefr = bounce && m.refraction > 0;
            if (hasRefl) directScale = 1 - m.reflection;
            if (hasRefr) directScale2 = 1 - m.refraction;

            // texture and roughness are light-independent: evaluated once (the reference re-evaluates them per light
            // that faces the surface; a hit no light faces does not use them) -- and not at all where the direct term
            // is multiplied by an exact zero (deadDirect below: no shadow ray is emitted, nothing reads dcol / shiny)
            if (!(directScale == 0.0f || directScale2 == 0.0f || weight == 0.0f)) {
                dcol = m.texture != RT_TEX_NONE ? rtEvalTexture<FAST>(sc, m.texture, h.uv) : ld3(m.diffuse);
                shiny = m.shininess;
                if (!FAST && m.has_img_texture) {  // main.cpp:116-117, double pow
                    const double rough = (double)rtEvalTexture(sc, m.roughness, h.uv).x;
                    const double r2 = rough * rough;
                    shiny = (float)((0.5 / (r2 * r2)) - 0.5);
                }
            }
            // ambient term, added at every node and never scaled (main.cpp:207-211)
            V3 amb;
            if (!FAST && m.has_img_texture) amb = ld3(sc.ambient) * 0.1f * rtEvalTexture(sc, m.occlusion, h.uv).x;
            else amb = ld3(sc.ambient) * ld3(m.ambient);
            amb0 = amb * weight;  // level 0: first contribution of the sample, stored below together with the zeros of the misses

            if (hasRefl) {  // main.cpp:152-166
                reflDir = rtReflect(-view, ns);
                wRefl = weight * m.reflection;
                emitRefl = true;
            }
            if (hasRefr) {  // main.cpp:171-203
                const bool entering = dot(ns, -view) < 0.0f;
                const float n1 = entering ? 1.0f : m.sigma, n2 = entering ? m.sigma : 1.0f;
                const float eta = n1 / n2;
                if (rtRefract(-view, entering ? ns : -ns, eta, refrDir)) {
                    const float O1 = RT_COS_OF_ANGLE(ns, view);
                    const float O2 = RT_COS_OF_ANGLE(-ns, refrDir);
                    const double ra = (double)((n1 * O1 - n2 * O2) / (n1 * O1 + n2 * O2));
                    const double rb = (double)((n1 * O2 - n2 * O1) / (n1 * O2 + n2 * O1));
                    fresnelR = (float)(0.5 * (ra * ra + rb * rb));
                    const float T = 1 - fresnelR;
                    wRefr = weight * m.refraction * T;
                    emitRefr = true;
                }
            }
        }

        // ambient term of the hit (levels >= 1); an exactly zero term (materials without ambient colour) adds nothing
        accumulateRuns(q.accum, active && level > 0 && !(amb0.x == 0.0f && amb0.y == 0.0f && amb0.z == 0.0f), sample, amb0);
        if (level == 0 && active) {
            // Level 0 holds exactly one entry per sample slot (hit, miss or outside the image), and it is the first to
            // touch the slot: a plain store initialises the accumulator -- no memset of the buffer, no atomics here.
            // (misses: the scan above)
            long long *acc = q.accum + 3ull * sample;
            acc[0] = rtToFixed(amb0.x);
            acc[1] = rtToFixed(amb0.y);
            acc[2] = rtToFixed(amb0.z);
        }
        RT_PROF_MARK(1)
        // ---- direct light: per light the pending radiance (main.cpp:102-145) rides on its shadow ray ----
        // A shadow ray whose pending radiance is exactly zero cannot change the image: whether it reaches the light or
        // not, zero is added.  The reference still casts it (is_shadowed runs before the colour is computed,
        // main.cpp:100-103) -- and it is common: `color *= 1 - refraction` (main.cpp:172) zeroes the whole direct term
        // of every material with refraction 1 (crystal, the refractive floor, glass) wherever bounces remain, i.e. all
        // their shadow rays above the last level.  Such rays are not emitted; they are counted (shadow_zero) so that
        // traced + zero equals the reference's count.  Exactly zero means: a factor of the product is exactly zero (the
        // product is then +-0 or, for an infinite / NaN direct term, NaN -- which rtToFixed drops: zero either way), or
        // the computed colour is +-0 in all three channels.  Results are bit-identical.
        const bool deadDirect = directScale == 0.0f || directScale2 == 0.0f || weight == 0.0f;
        for (int l = 0; l < nlights; l++) {
            // is_shadowed's back-face test (main.cpp:44-46): a light behind the surface casts no shadow ray
            const V3 lp = ld3(sc.lights[l].pos), toL = lp - P;
            const V3 ld = normalize(toL);                            // main.cpp:97
            const bool faces = active && !(dot(normal, ld) < 0);
            if (active && !faces) culled++;
            bool lit = faces && !deadDirect;
            V3 c3 = v3s(0.0f);
            if (lit) {
                const V3 lc = ld3(sc.lights[l].color);
                const float dl = length(toL);                        // :98
                const float diffuse = smax(0.0f, dot(ld, ns));       // :104
                const V3 hv = normalize(ld + view);                  // :106
                const float distance = smax(0.1f, dl);               // :109
                const float attenuation = RT_COLOR_ATTENUATION(distance);  // :110 (double pow)
                const V3 diffusion = attenuation * lc * dcol * diffuse;
                V3 spec = v3s(0.0f);
                if (!FAST && aniso) {  // main.cpp:123-139
                    const float NdotL = dot(ns, ld), NdotV = dot(ns, view);
                    if (NdotL > 0 && NdotV > 0) {
                        const float HdotT = dot(hv, hT), HdotB = dot(hv, hB), HdotN = dot(hv, ns);
                        const float e1 = HdotT / alpha_x, e2 = HdotB / alpha_y;
                        const float exponent = -2.0f * (powf(e1, 2.0f) * powf(e2, 2.0f)) / (1 + HdotN);
                        const float pi = 3.14159265358979323846264338327950288f;
                        spec = (mspec * NdotL * expf(exponent)) / (sqrtf(NdotL * NdotV) * 4 * pi * alpha_x * alpha_y);
                    }
                } else {  // :141-142; std::max(0.0f, NaN) == 0.0f
                    const float specular = RT_COLOR_POW_MAX0(dot(hv, ns), 4 * shiny);
                    spec = attenuation * lc * mspec * specular;
                }
                c3 = diffusion + spec;
                c3 = c3 * directScale;   // x * 1.0f is exact when no scaling applies
                c3 = c3 * directScale2;
                c3 = c3 * weight;
                if (c3.x == 0.0f && c3.y == 0.0f && c3.z == 0.0f) lit = false;  // (a light the shading normal faces away from: diffuse 0)
            }
            if (faces && !lit) zero++;
            const unsigned long long mask = __ballot(lit);
            if (!mask) continue;  // wave-uniform
            const uint32_t c = (uint32_t)__popcll(mask), rank = maskRankBelow(mask);
            uint32_t cur = __builtin_amdgcn_readlane(curv, l), end = __builtin_amdgcn_readlane(endv, l);
            const uint32_t room = end - cur;
            uint32_t slot = cur + rank;
            if (c > room) {  // the slab is used up to its last slot, the rest of the rays opens the next one
                const uint32_t sz = slabS;
                const uint32_t nbase = waveReserve(q.counters + RT_CNT_SHADOW(par), sz);
                // (every reservation on this counter is one slab: slabs are aligned, and slot >> log2(slab) names the slab --
                // its light is written once here instead of a byte per ray)
                if (lane == 0 && nbase < q.shadow_cap) q.shSlabLight[nbase >> q.sh_slab_shift] = (uint8_t)l;
                if (rank >= room) slot = nbase + (rank - room);
                cur = nbase + (c - room);
                end = nbase + sz;
            } else {
                cur += c;
            }
            if (lane == (uint32_t)l) {
                curv = cur;
                endv = end;
            }
            if (lit) {
                if (slot < q.shadow_cap) {  // 32-byte record: point + sample, pending colour (rtShadowRay rebuilds the ray; the light: shSlabLight)
                    q.shA[slot] = make_float4(P.x, P.y, P.z, __uint_as_float(sample | (level == 0 ? RT_SHADOW_L0 : 0u)));
                    if (level == 0) l0lit++;
                    q.shC[slot] = make_float4(c3.x, c3.y, c3.z, 0.0f);
                } else {
                    atomicAdd(q.counters + RT_CNT_OVERFLOW, 1u);
                }
            }
        }
        RT_PROF_MARK(3)
        // ---- next-level rays, lane-major so that a reflection/refraction pair stays adjacent (also across slabs) ----
        const unsigned long long mRefl = __ballot(emitRefl), mRefr = __ballot(emitRefr);
        if (mRefl | mRefr) {  // wave-uniform
            const uint32_t k = (emitRefl ? 1u : 0u) + (emitRefr ? 1u : 0u);
            const uint32_t off = maskRankBelow(mRefl) + maskRankBelow(mRefr);
            const uint32_t c = (uint32_t)(__popcll(mRefl) + __popcll(mRefr));
            uint32_t cur = __builtin_amdgcn_readlane(curv, RT_MAX_LIGHTS), end = __builtin_amdgcn_readlane(endv, RT_MAX_LIGHTS);
            const uint32_t room = end - cur;
            uint32_t slot = cur + off;
            if (c > room) {
                const uint32_t sz = slabR;
                const uint32_t nbase = waveReserve(q.counters + RT_CNT_RAYS(nbuf), sz);
                // first lane whose rays do not fit any more: it and the lanes after it move to the new slab
                const unsigned long long nofit = __ballot(k > 0 && off + k > room);
                const uint32_t f = (uint32_t)__ffsll((long long)nofit) - 1u;
                const uint32_t off_f = __builtin_amdgcn_readlane(off, f);
                if (lane >= f) slot = nbase + (off - off_f);
                if (lane == f && room > off_f) {  // a pair met the slab's last slot: that slot becomes a hole
                    if (end - 1 < q.ray_cap) q.rayB[nbuf][end - 1] = make_float4(0.0f, 0.0f, 1.0f, __uint_as_float(RT_FLAG_INVALID));
                    holesR++;
                }
                cur = nbase + (c - off_f);
                end = nbase + sz;
            } else {
                cur += c;
            }
            if (lane == RT_MAX_LIGHTS) {
                curv = cur;
                endv = end;
            }
            const uint32_t ntag = sample | ((level + 1u) << q.level_shift);  // the children are one level deeper
            if (emitRefl) {
                if (slot < q.ray_cap) {
                    const V3 ro = P + RT_EPSILON * reflDir;  // main.cpp:155
                    q.rayA[nbuf][slot] = make_float4(ro.x, ro.y, ro.z, wRefl);
                    q.rayB[nbuf][slot] = make_float4(reflDir.x, reflDir.y, reflDir.z,
                                                     __uint_as_float(ntag | (emitRefr ? RT_FLAG_HAS_REFR_SIBLING : 0u)));
                    if (emitRefr) q.rayR[nbuf][slot] = fresnelR;  // (only read for a reflection ray with a refraction sibling)
                } else {
                    atomicAdd(q.counters + RT_CNT_OVERFLOW, 1u);
                }
                slot++;
            }
            if (emitRefr) {
                if (slot < q.ray_cap) {
                    const V3 ro = P + RT_EPSILON * refrDir;  // main.cpp:181
                    q.rayA[nbuf][slot] = make_float4(ro.x, ro.y, ro.z, wRefr);
                    q.rayB[nbuf][slot] = make_float4(refrDir.x, refrDir.y, refrDir.z, __uint_as_float(ntag));
                } else {
                    atomicAdd(q.counters + RT_CNT_OVERFLOW, 1u);
                }
            }
        }
        RT_PROF_MARK(4)
        if (PROF) prof[5]++;
    }
    // ---- what is left of the wave's slabs becomes holes ----
    for (int l = 0; l < nlights; l++) {
        const uint32_t cur = __builtin_amdgcn_readlane(curv, l), end = min(__builtin_amdgcn_readlane(endv, l), q.shadow_cap);
        for (uint32_t sl = cur + lane; sl < end; sl += 64) q.shA[sl] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(RT_SHADOW_HOLE));
        if (lane == 0 && end > cur) holesS += end - cur;
    }
    {
        const uint32_t cur = __builtin_amdgcn_readlane(curv, RT_MAX_LIGHTS), end = min(__builtin_amdgcn_readlane(endv, RT_MAX_LIGHTS), q.ray_cap);
        for (uint32_t sl = cur + lane; sl < end; sl += 64) q.rayB[nbuf][sl] = make_float4(0.0f, 0.0f, 1.0f, __uint_as_float(RT_FLAG_INVALID));
        if (lane == 0 && end > cur) holesR += end - cur;
    }
    if (!GENERAL)  // unused slots of the general list's slab: holes the general instance's scan skips
        for (uint32_t sl = gcur + lane; sl < min(gend, q.general_cap); sl += 64) q.general[sl] = RT_GENERAL_HOLE;
    if (PROF && lane == 0) {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(q.counters + RT_CNT_PROF_SHADE);
        for (int k = 0; k < 6; k++) atomicAdd(dst + k, prof[k]);
    }
#undef RT_PROF_MARK
    // statistics: one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        culled += __shfl_down(culled, off);
        zero += __shfl_down(zero, off);
        l0lit += __shfl_down(l0lit, off);
        shaded += __shfl_down(shaded, off);
        holesS += __shfl_down(holesS, off);
        holesR += __shfl_down(holesR, off);
    }
    if (lane == 0) {
        if (culled) atomicAdd(q.counters + RT_CNT_CULLED, culled);
        if (zero) atomicAdd(q.counters + RT_CNT_ZERO, zero);
        if (l0lit) atomicAdd(q.counters + RT_CNT_SHADOW_L0, l0lit);
        if (shaded) atomicAdd(q.counters + RT_CNT_SHADED, shaded);
        if (holesS) atomicAdd(q.counters + RT_CNT_SHADOW_HOLES, holesS);
        if (holesR) atomicAdd(q.counters + RT_CNT_INVALID, holesR);  // ray-queue holes are invalid slots like those of k_gen_primary
    }
    rtTailEnd(q, tp0);
}

// Level-0 shadow results that k_any stored instead of adding (RtQueues::occ0): one lane per sample of the chunk that entered
// the pipeline in this iteration adds the radiance of its unoccluded level-0 shadow rays to the sample's accumulator and
// clears the flags for the next iteration.  Runs behind the iteration's k_any and its split rounds, before anything of the
// chunk's level 1 touches the accumulators; no other kernel is active, so plain read-modify-write.
__global__ __launch_bounds__(RT_BLOCK) void k_fold0(RtQueues q, int nlights) {
    const uint32_t samples = q.prim_n;
    for (uint32_t s = blockIdx.x * RT_BLOCK + threadIdx.x; s < samples; s += gridDim.x * RT_BLOCK) {
        unsigned long long ax = 0, ay = 0, az = 0;
        bool any = false;
        for (int l = 0; l < nlights; l++) {
            const size_t at = (size_t)l * q.l0_stride + s;
            if (q.occ0[at]) {
                const float4 C = q.col0[at];
                ax += (unsigned long long)rtToFixed(C.x);
                ay += (unsigned long long)rtToFixed(C.y);
                az += (unsigned long long)rtToFixed(C.z);
                any = true;
                q.occ0[at] = 0;
            }
        }
        if (any) {
            unsigned long long *acc = reinterpret_cast<unsigned long long *>(q.accum) + 3ull * (q.prim_sample0 + s);
            acc[0] += ax;
            acc[1] += ay;
            acc[2] += az;
        }
    }
}

// Between two iterations of the pipeline (one thread; runs at the START of iteration j, before anything else of it):
// moves the 32-bit counters of iteration j - 1 (parity `prev`) into the frame's 64-bit totals and prepares the counters of
// that parity for iteration j + 1 -- its ray queue starts behind the `prim_next` slots of the primary rays that will enter
// then.  No host round trip, no memset between iterations.
__global__ void k_iter_ctl(RtQueues q, int prev, uint32_t prim_next, int set_cur, uint32_t prim_cur) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t *c = q.counters;
    unsigned long long *tot = reinterpret_cast<unsigned long long *>(c + RT_CNT_TOTALS);
    const uint32_t rays = c[RT_CNT_RAYS(prev)], sh = c[RT_CNT_SHADOW(prev)];
    tot[RT_TOT_RAYS] += rays;
    tot[RT_TOT_SHADOW] += sh;
    if (rays > tot[RT_TOT_MAX_RAYS]) tot[RT_TOT_MAX_RAYS] = rays;
    if (sh > tot[RT_TOT_MAX_SHADOW]) tot[RT_TOT_MAX_SHADOW] = sh;
    tot[RT_TOT_CULLED] += c[RT_CNT_CULLED];
    tot[RT_TOT_SHADED] += c[RT_CNT_SHADED];
    tot[RT_TOT_OVERFLOW] += c[RT_CNT_OVERFLOW];
    tot[RT_TOT_INVALID] += c[RT_CNT_INVALID];
    tot[RT_TOT_SHADOW_HOLES] += c[RT_CNT_SHADOW_HOLES];
    tot[RT_TOT_ZERO] += c[RT_CNT_ZERO];
    tot[RT_TOT_SHADOW_L0] += c[RT_CNT_SHADOW_L0];
    c[RT_CNT_CULLED] = c[RT_CNT_SHADED] = c[RT_CNT_OVERFLOW] = c[RT_CNT_INVALID] = 0;
    c[RT_CNT_SHADOW_HOLES] = c[RT_CNT_ZERO] = c[RT_CNT_SHADOW_L0] = 0;
    for (int r = 0; r <= RT_SPLIT_ROUNDS; r++) {
        tot[RT_TOT_SPLIT] += min(c[RT_CNT_SPLIT(prev, r)], q.split_cap);
        c[RT_CNT_SPLIT(prev, r)] = 0;
        c[RT_CNT_SPLIT(prev, r) + 1] = 0;
    }
    c[RT_CNT_RAYS(prev)] = prim_next;
    c[RT_CNT_SHADOW(prev)] = 0;
    c[RT_CNT_GENERAL(prev)] = 0;
    for (int k = 0; k < RT_WORK_PARTS; k++) {
        c[RT_CNT_WORKP_CLOSEST(prev, k)] = 0;
        c[RT_CNT_WORKP_ANY(prev, k)] = 0;
        c[RT_CNT_WORKP_PACKET(prev, k)] = 0;
        c[RT_CNT_WORKP_SHADE(prev, k)] = 0;
        c[RT_CNT_WORKP_SHADEG(prev, k)] = 0;
    }
    if (set_cur) c[RT_CNT_RAYS(prev ^ 1)] = prim_cur;
}

// ---------------------------------------------------------------------------------------------------------------
// trace_ray's clamp per SAMPLE (main.cpp:232), mean over samples (:646), toneMapping (:240-256),
// Image::setPixel truncation (Image.h:91-95).  One lane per pixel slot of the chunk that leaves the pipeline; outputs are tile-major.
__global__ __launch_bounds__(RT_BLOCK) void k_resolve(RtFrameDev f, RtQueues q, uint32_t sample0, uint32_t tile0, uint32_t ntiles,
                                                     uint8_t *rgb8_tiles, float *rgbf_tiles) {
    const uint32_t total = ntiles * RT_TILE_PIXELS;
    for (uint32_t pm = blockIdx.x * RT_BLOCK + threadIdx.x; pm < total; pm += gridDim.x * RT_BLOCK) {
        const uint32_t m = pm & 255u;
        const uint32_t px = compact1By1(m), py = compact1By1(m >> 1);
        V3 sum = v3s(0.0f);
        const long long *a0 = q.accum + 3ull * ((size_t)sample0 + (size_t)pm * f.spp);  // sample0: the chunk's slot in the accumulator ring
        int s = 0;
        // A lane's samples are 24 * spp contiguous bytes, 1536 B away from its neighbour's at 64 spp: fetched 8 bytes at a time
        // the 64 lines a wave touches per load fall out of the L1 before their other samples are read (counter traffic 2.8 x
        // the buffer).  Four samples = six 16-byte loads issued together use every line while it is there.  The samples are
        // still added one after the other, in order, as the reference's loop does (main.cpp:646): same float bits.
        if ((f.spp & 3) == 0) {
            for (; s < f.spp; s += 4) {
                const longlong2 *a = reinterpret_cast<const longlong2 *>(a0 + 3 * s);
                const longlong2 v0 = a[0], v1 = a[1], v2 = a[2], v3 = a[3], v4 = a[4], v5 = a[5];
                sum.x += gclamp(rtFromFixed(v0.x), 0.0f, 1.0f); sum.y += gclamp(rtFromFixed(v0.y), 0.0f, 1.0f); sum.z += gclamp(rtFromFixed(v1.x), 0.0f, 1.0f);
                sum.x += gclamp(rtFromFixed(v1.y), 0.0f, 1.0f); sum.y += gclamp(rtFromFixed(v2.x), 0.0f, 1.0f); sum.z += gclamp(rtFromFixed(v2.y), 0.0f, 1.0f);
                sum.x += gclamp(rtFromFixed(v3.x), 0.0f, 1.0f); sum.y += gclamp(rtFromFixed(v3.y), 0.0f, 1.0f); sum.z += gclamp(rtFromFixed(v4.x), 0.0f, 1.0f);
                sum.x += gclamp(rtFromFixed(v4.y), 0.0f, 1.0f); sum.y += gclamp(rtFromFixed(v5.x), 0.0f, 1.0f); sum.z += gclamp(rtFromFixed(v5.y), 0.0f, 1.0f);
            }
        }
        for (; s < f.spp; s++) {
            const long long *a = a0 + 3 * s;
            sum.x += gclamp(rtFromFixed(a[0]), 0.0f, 1.0f);
            sum.y += gclamp(rtFromFixed(a[1]), 0.0f, 1.0f);
            sum.z += gclamp(rtFromFixed(a[2]), 0.0f, 1.0f);
        }
        sum = sum / (float)f.spp;
        const size_t o = ((size_t)(tile0 + (pm >> 8) * q.tile_stride) * RT_TILE_PIXELS + py * RT_TILE + px) * 3;
        if (rgbf_tiles) {
            rgbf_tiles[o] = sum.x;
            rgbf_tiles[o + 1] = sum.y;
            rgbf_tiles[o + 2] = sum.z;
        }
        if (rgb8_tiles) {
            rgb8_tiles[o] = (uint8_t)(int)(255 * rtToneMap1(sum.x));
            rgb8_tiles[o + 1] = (uint8_t)(int)(255 * rtToneMap1(sum.y));
            rgb8_tiles[o + 2] = (uint8_t)(int)(255 * rtToneMap1(sum.z));
        }
    }
}

// rank-ordered tile-major buffers (what an RCCL gather delivers) -> row-major image
__global__ __launch_bounds__(RT_BLOCK) void k_assemble(int width, int height, int world, long long tiles_per_rank,
                                                      const uint8_t *t8, const float *tf, const uint32_t *tp,
                                                      uint8_t *o8, float *of, uint32_t *op) {
    const int tiles_x = (width + RT_TILE - 1) / RT_TILE;
    const size_t npix = (size_t)width * height;
    for (size_t p = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x; p < npix; p += (size_t)gridDim.x * RT_BLOCK) {
        const int x = (int)(p % width), y = (int)(p / width);
        const int gt = (y / RT_TILE) * tiles_x + (x / RT_TILE);
        const int rank = gt % world;
        const long long lt = gt / world;
        const size_t src = ((size_t)rank * tiles_per_rank + lt) * RT_TILE_PIXELS + (y % RT_TILE) * RT_TILE + (x % RT_TILE);
        if (o8 && t8) {
            o8[3 * p] = t8[3 * src];
            o8[3 * p + 1] = t8[3 * src + 1];
            o8[3 * p + 2] = t8[3 * src + 2];
        }
        if (of && tf) {
            of[3 * p] = tf[3 * src];
            of[3 * p + 1] = tf[3 * src + 1];
            of[3 * p + 2] = tf[3 * src + 2];
        }
        if (op && tp) op[p] = tp[src];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// test hooks
__global__ __launch_bounds__(RT_BLOCK) void k_hitinfo(RtSceneDev sc, uint32_t root_ref, RtQueues q, uint32_t n,
                                                     rt_hit *out_hits, uint8_t *out_shadow, int stack_entries) {
    extern __shared__ uint32_t s_lds[];
    const RtStack stk = rtMakeStack(s_lds, q.spill, stack_entries);
    const RtAnalytic an = stageAnalytic(sc, s_lds + stack_entries * RT_BLOCK);
    RtTraversalStats st{0, 0};
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += gridDim.x * RT_BLOCK) {
        const float4 A = q.rayA[0][i], B = q.rayB[0][i];
        const float2 hrec = q.hits[i];
        const uint32_t id = __float_as_uint(hrec.y);
        const uint32_t sample = __float_as_uint(B.w) & q.sample_mask;
        RtRay r{v3(A.x, A.y, A.z), v3(B.x, B.y, B.z)};
        rt_hit o;
        memset(&o, 0, sizeof o);
        o.object = -1;
        o.triangle = -1;
        o.prim = RT_PRIM_MISS;
        o.distance = hrec.x;
        uint8_t sh[4] = {0, 0, 0, 0};
        if (id != RT_ID_MISS) {
            RtFullHit h;
            rtReconstructHit(sc, r, hrec.x, id, h);
            o.hit = 1;
            o.object = h.object;
            o.prim = h.prim;
            o.triangle = h.triangle;  // slot; converted to the index inside the mesh on the host
            o.intersection[0] = h.P.x; o.intersection[1] = h.P.y; o.intersection[2] = h.P.z;
            o.normal[0] = h.normal.x; o.normal[1] = h.normal.y; o.normal[2] = h.normal.z;
            o.normal_shading[0] = h.ns.x; o.normal_shading[1] = h.ns.y; o.normal_shading[2] = h.ns.z;
            o.uv[0] = h.uv.x; o.uv[1] = h.uv.y;
            o.tangent[0] = h.T.x; o.tangent[1] = h.T.y; o.tangent[2] = h.T.z;
            o.bitangent[0] = h.B.x; o.bitangent[1] = h.B.y; o.bitangent[2] = h.B.z;
            if (out_shadow) {
                const V3 view = normalize(-r.d);
                V3 normal = h.normal;
                if (dot(normal, view) < 0) normal = -normal;
                for (int l = 0; l < sc.n_lights && l < 4; l++) {
                    const V3 lp = ld3(sc.lights[l].pos);
                    const V3 ld = normalize(lp - h.P);
                    const float dl = length(lp - h.P);
                    bool s = true;
                    if (!(dot(normal, ld) < 0)) {
                        RtRay sr{h.P + RT_EPSILON * ld, ld};
                        s = occluded(sc, an, root_ref, sr, dl, stk, st);
                    }
                    sh[l] = s ? 1 : 0;
                }
            }
        }
        if (out_hits) out_hits[sample] = o;
        if (out_shadow)
            for (int l = 0; l < 4; l++) out_shadow[4 * (size_t)sample + l] = sh[l];
    }
}

__global__ void k_rays_from_host(const float *rays, uint32_t n, RtQueues q) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        q.rayA[0][i] = make_float4(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2], 1.0f);
        q.rayB[0][i] = make_float4(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5], __uint_as_float(i));
        q.rayR[0][i] = 1.0f;  // (the queue's length is set by k_iter_ctl: the batch is the "chunk" that enters at iteration 0)
    }
}

__global__ void k_batch_colors(RtQueues q, uint32_t n, float *out) {  // trace_ray's clamp, main.cpp:232
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int c = 0; c < 3; c++) out[3 * (size_t)i + c] = gclamp(rtFromFixed(q.accum[3 * (size_t)i + c]), 0.0f, 1.0f);
}

__global__ void k_eval_texture(RtSceneDev sc, int id, long long n, const float *uv, float *out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        V3 c = rtEvalTexture(sc, id, V2{uv[2 * i], uv[2 * i + 1]});
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}

__global__ void k_eval_perlin(long long n, const float *p, float *out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = rtPerlin3(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
}

// ---------------------------------------------------------------------------------------------------------------
// launchers (host).  Grids are sized for the chip (256 CUs), not for the queue: queue lengths live in device memory
// and the kernels grid-stride over them, so a frame needs no host round trip between levels.
static inline int gridFor(const RtLaunchCfg &cfg, int blocks_per_cu) { return cfg.num_cus * blocks_per_cu; }

void rtLaunchIterCtl(const RtLaunchCfg &cfg, const RtQueues &q, int prev, uint32_t prim_next, bool set_cur, uint32_t prim_cur) {
    hipLaunchKernelGGL(k_iter_ctl, dim3(1), dim3(64), 0, cfg.stream, q, prev, prim_next, set_cur ? 1 : 0, prim_cur);
}
void rtLaunchGenPrimary(const RtLaunchCfg &cfg, const RtFrameDev &f, const RtQueues &q, int par, uint32_t tile0, uint32_t ntiles) {
    hipLaunchKernelGGL(k_gen_primary, dim3(gridFor(cfg, 8)), dim3(RT_BLOCK), 0, cfg.stream, f, q, par, tile0, ntiles);
}
// (stats: 0 = plain, 1 = counters, 2 = counters + wave-time profile)
#define RT_BY_STATS(CALL) do { if (stats >= 2) { CALL(2); } else if (stats == 1) { CALL(1); } else { CALL(0); } } while (0)
static inline size_t traceLds(const RtLaunchCfg &cfg) { return (size_t)(cfg.stack_entries + RT_TRACE_LDS_ROWS) * RT_BLOCK * sizeof(uint32_t) + cfg.analytic_bytes; }
static inline uint32_t splitBudget(const RtLaunchCfg &cfg, const RtQueues &q) { return q.split_cap ? (uint32_t)cfg.split_budget : 0u; }
// The fused traversal launch of an iteration (k_trace): shadow queue `par_any`, bounce rays [first, n) of ray queue
// `par_closest`, its primary rays [0, q.prim_n) as packets.  A parity < 0 skips the phase.
void rtLaunchTrace(const RtLaunchCfg &cfg, const RtSceneDev &sc, uint32_t root_ref, const RtQueues &q, int par_any, int par_closest,
                   uint32_t first, bool packets, int stats) {
    const size_t lds = traceLds(cfg);
    const dim3 grid(gridFor(cfg, cfg.trace_blocks_per_cu)), block(RT_BLOCK);
    const uint32_t budget = splitBudget(cfg, q);
#define RT_T(S)                                                                                                                                        \
    do {                                                                                                                                               \
        if (q.occ0) hipLaunchKernelGGL((k_trace<S, true>), grid, block, lds, cfg.stream, sc, root_ref, q, par_any, par_closest, first, packets ? 1 : 0, cfg.stack_entries, budget);  \
        else hipLaunchKernelGGL((k_trace<S, false>), grid, block, lds, cfg.stream, sc, root_ref, q, par_any, par_closest, first, packets ? 1 : 0, cfg.stack_entries, budget);       \
    } while (0)
    RT_BY_STATS(RT_T);
#undef RT_T
}
// the same phases as launches of their own
void rtLaunchPacket(const RtLaunchCfg &cfg, const RtSceneDev &sc, uint32_t root_ref, const RtQueues &q, int par, int stats) {
    const dim3 grid(gridFor(cfg, cfg.trace_blocks_per_cu)), block(RT_BLOCK);
    if (stats) hipLaunchKernelGGL((k_packet<true>), grid, block, traceLds(cfg), cfg.stream, sc, root_ref, q, par, cfg.stack_entries);
    else hipLaunchKernelGGL((k_packet<false>), grid, block, traceLds(cfg), cfg.stream, sc, root_ref, q, par, cfg.stack_entries);
}
void rtLaunchClosest(const RtLaunchCfg &cfg, const RtSceneDev &sc, uint32_t root_ref, const RtQueues &q, int par, uint32_t first, int stats) {
    const size_t lds = traceLds(cfg);
    const dim3 grid(gridFor(cfg, cfg.trace_blocks_per_cu)), block(RT_BLOCK);
#define RT_C_QUEUE(S) hipLaunchKernelGGL((k_closest<S>), grid, block, lds, cfg.stream, sc, root_ref, q, par, first, cfg.stack_entries)
    RT_BY_STATS(RT_C_QUEUE);
#undef RT_C_QUEUE
}
#define RT_LAUNCHA(S, M, G, R, B)                                                                                                     \
    do {                                                                                                                              \
        if (q.occ0) hipLaunchKernelGGL((k_any<S, M, true>), G, block, lds, cfg.stream, sc, root_ref, q, par, cfg.stack_entries, R, B);  \
        else hipLaunchKernelGGL((k_any<S, M, false>), G, block, lds, cfg.stream, sc, root_ref, q, par, cfg.stack_entries, R, B);       \
    } while (0)
void rtLaunchAny(const RtLaunchCfg &cfg, const RtSceneDev &sc, uint32_t root_ref, const RtQueues &q, int par, int stats) {
    const size_t lds = traceLds(cfg);
    const dim3 grid(gridFor(cfg, cfg.trace_blocks_per_cu)), block(RT_BLOCK);
    const uint32_t budget = splitBudget(cfg, q);
#define RT_A_QUEUE(S) RT_LAUNCHA(S, RT_MODE_QUEUE, grid, -1, budget)
    RT_BY_STATS(RT_A_QUEUE);
#undef RT_A_QUEUE
}
// follow-up rounds over the work items of split shadow walks of queue `par` (almost always without work: their waves read
// a zero count and leave); a quarter of the grid is plenty, the last round walks whatever is left to the end
void rtLaunchSplitRounds(const RtLaunchCfg &cfg, const RtSceneDev &sc, uint32_t root_ref, const RtQueues &q, int par, int stats) {
    const uint32_t budget = splitBudget(cfg, q);
    if (!budget) return;
    const size_t lds = traceLds(cfg);
    const dim3 block(RT_BLOCK), sgrid(gridFor(cfg, cfg.trace_blocks_per_cu >= 4 ? cfg.trace_blocks_per_cu / 4 : 1));
    for (int r = 0; r < RT_SPLIT_ROUNDS; r++) {
        // items get an eighth of the budget: a round then lasts ~0.15 ms and the items multiply faster
        const uint32_t b = r == RT_SPLIT_ROUNDS - 1 ? 0u : (budget >= 64u ? budget / 8u : budget);
#define RT_A_SPLIT(S) RT_LAUNCHA(S, RT_MODE_SPLIT, sgrid, r, b)
        RT_BY_STATS(RT_A_SPLIT);
#undef RT_A_SPLIT
    }
}
#undef RT_LAUNCHA
#undef RT_BY_STATS
void rtLaunchShade(const RtLaunchCfg &cfg, const RtSceneDev &sc, const RtQueues &q, int par, int max_depth, int spp,
                   uint32_t tile0, uint32_t *primid_tiles, bool prof, uint64_t max_inputs, int which) {
    const uint32_t slab = rtShadeSlab(cfg, max_inputs);
    const dim3 grid(rtShadeBlocks(cfg)), block(RT_BLOCK);
    // the fast instance first (it fills the general list), then -- only for scenes that hold spheres / cones, BMP-textured
    // or anisotropic materials -- the general instance over that list.  Both reserve slabs: the queue slack covers two
    // partly filled slabs per wave and queue (rt_abi.cpp: queueSizes).
    if (which != 2) {
        if (prof) hipLaunchKernelGGL((k_shade<true, false>), grid, block, 0, cfg.stream, sc, q, par, max_depth, spp, tile0, primid_tiles, slab);
        else hipLaunchKernelGGL((k_shade<false, false>), grid, block, 0, cfg.stream, sc, q, par, max_depth, spp, tile0, primid_tiles, slab);
    }
    if (sc.has_general && which != 1) {
        if (prof) hipLaunchKernelGGL((k_shade<true, true>), grid, block, 0, cfg.stream, sc, q, par, max_depth, spp, tile0, primid_tiles, slab);
        else hipLaunchKernelGGL((k_shade<false, true>), grid, block, 0, cfg.stream, sc, q, par, max_depth, spp, tile0, primid_tiles, slab);
    }
}
void rtLaunchFold0(const RtLaunchCfg &cfg, const RtQueues &q, int nlights) {
    hipLaunchKernelGGL(k_fold0, dim3(gridFor(cfg, 8)), dim3(RT_BLOCK), 0, cfg.stream, q, nlights);
}
void rtLaunchResolve(const RtLaunchCfg &cfg, const RtFrameDev &f, const RtQueues &q, uint32_t sample0, uint32_t tile0, uint32_t ntiles,
                     uint8_t *rgb8_tiles, float *rgbf_tiles) {
    hipLaunchKernelGGL(k_resolve, dim3(gridFor(cfg, 4)), dim3(RT_BLOCK), 0, cfg.stream, f, q, sample0, tile0, ntiles, rgb8_tiles, rgbf_tiles);
}
void rtLaunchAssemble(const RtLaunchCfg &cfg, int width, int height, int world, long long tiles_per_rank, const void *t8,
                      const void *tf, const void *tp, void *o8, void *of, void *op) {
    hipLaunchKernelGGL(k_assemble, dim3(gridFor(cfg, 4)), dim3(RT_BLOCK), 0, cfg.stream, width, height, world, tiles_per_rank,
                       (const uint8_t *)t8, (const float *)tf, (const uint32_t *)tp, (uint8_t *)o8, (float *)of, (uint32_t *)op);
}
void rtLaunchHitInfo(const RtLaunchCfg &cfg, const RtSceneDev &sc, uint32_t root_ref, const RtQueues &q, uint32_t n,
                     rt_hit *out_hits, uint8_t *out_shadow) {
    const size_t lds = (size_t)cfg.stack_entries * RT_BLOCK * sizeof(uint32_t) + cfg.analytic_bytes;
    hipLaunchKernelGGL(k_hitinfo, dim3(gridFor(cfg, 2)), dim3(RT_BLOCK), lds, cfg.stream, sc, root_ref, q, n, out_hits, out_shadow, cfg.stack_entries);
}
void rtLaunchRaysFromHost(const RtLaunchCfg &cfg, const float *d_rays, uint32_t n, const RtQueues &q) {
    hipLaunchKernelGGL(k_rays_from_host, dim3(gridFor(cfg, 2)), dim3(RT_BLOCK), 0, cfg.stream, d_rays, n, q);
}
void rtLaunchBatchColors(const RtLaunchCfg &cfg, const RtQueues &q, uint32_t n, float *d_out) {
    hipLaunchKernelGGL(k_batch_colors, dim3(gridFor(cfg, 2)), dim3(RT_BLOCK), 0, cfg.stream, q, n, d_out);
}
void rtLaunchEvalTexture(const RtLaunchCfg &cfg, const RtSceneDev &sc, int id, long long n, const float *d_uv, float *d_out) {
    hipLaunchKernelGGL(k_eval_texture, dim3(gridFor(cfg, 2)), dim3(RT_BLOCK), 0, cfg.stream, sc, id, n, d_uv, d_out);
}
void rtLaunchEvalPerlin(const RtLaunchCfg &cfg, long long n, const float *d_p, float *d_out) {
    hipLaunchKernelGGL(k_eval_perlin, dim3(gridFor(cfg, 2)), dim3(RT_BLOCK), 0, cfg.stream, n, d_p, d_out);
}

// ---------------------------------------------------------------------------------------------------------------
// Static scratch of rt_render_tiles_static (include/rt_hip.h): the wavefront workspace of a caller that cannot
// allocate.  In host code the array's NAME is only a stand-in for the symbol, never a device address: the address on
// the current device comes from hipGetSymbolAddress, once per device (rt_create), and host code passes that pointer
// -- to kernels (inside RtQueues), to hipMemsetAsync -- wherever it needs the array's memory.
static __device__ __align__(256) unsigned char rt_static_scratch[RT_STATIC_SCRATCH_BYTES];

int rtScratchResolve(void **ptr, size_t *bytes, const char **err) {
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(rt_static_scratch));
    if (e != hipSuccess || !p) {
        if (err) *err = e != hipSuccess ? hipGetErrorString(e) : "hipGetSymbolAddress returned a null address";
        return 1;
    }
    *ptr = p;
    *bytes = RT_STATIC_SCRATCH_BYTES;
    return 0;
}
