// rt_bvh.cpp -- host-side builder of the linear BVH the traversal kernels walk (rt_kernels.hip: traverseBvh).
//
// Replaces the reference's per-mesh pointer tree (bvh_node, MeshLoader.h:211-285: split at the mean vertex
// coordinate, axis cycling with depth, leaves of <= 100 triangles that are copied by value per visit) with ONE
// binned-SAH binary tree over every triangle of the scene, collapsed into a 4-wide tree of 64-byte nodes whose
// child boxes are quantised to 8 bits per plane (rt_types.h: RtNode4):
//   * one node = one cache line = one fetch per traversal step, and about half the steps of the binary tree;
//   * quantisation rounds outwards in exact arithmetic (scale is a power of two), so boxes stay conservative;
//   * leaves hold <= RT_LEAF_MAX triangles that are contiguous in the reordered triangle array.
// The reference's tree is not reproduced -- only its observable result is (closest / any hit), plus the ORDER in
// which it hands out candidates, because that order decides exact distance ties (MeshLoader.h:399-404):
// rtReferenceOrder() recomputes that order so the kernel can break ties identically.
#include "rt_bvh.h"

#ifndef RT_SAH_STEP_COST
#define RT_SAH_STEP_COST 1.0f  // cost of a traversal step in triangle tests (leaf-vs-split decision)
#endif

#include <algorithm>
#include <cmath>
#include <emmintrin.h>
#include <cstring>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <thread>

namespace {

struct Box {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    void grow(const float *p) {
        for (int k = 0; k < 3; k++) {
            mn[k] = std::fmin(mn[k], p[k]);
            mx[k] = std::fmax(mx[k], p[k]);
        }
    }
    void grow(const Box &b) {
        for (int k = 0; k < 3; k++) {
            mn[k] = std::fmin(mn[k], b.mn[k]);
            mx[k] = std::fmax(mx[k], b.mx[k]);
        }
    }
    float halfArea() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (!(dx >= 0) || !(dy >= 0) || !(dz >= 0)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

#ifndef RT_SAH_BINS
#define RT_SAH_BINS 16
#endif
constexpr int kBins = RT_SAH_BINS;

// Host threads of the build (the 50 M-triangle scene).  RT_BUILD_THREADS overrides; OMP_NUM_THREADS
// is honoured because the GPU boxes set it to their CPU share.
int buildThreads() {
    for (const char *name : {"RT_BUILD_THREADS", "OMP_NUM_THREADS"})
        if (const char *e = std::getenv(name)) {
            const int v = std::atoi(e);
            if (v > 0) return std::min(v, 64);
        }
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::min(std::max(hw, 1u), 16u);
}

// fn(tid, begin, end) over [lo, hi) split into contiguous pieces, one thread each
template <class F>
void parallelRange(size_t lo, size_t hi, int threads, F fn) {
    const size_t n = hi - lo;
    threads = (int)std::min<size_t>((size_t)std::max(threads, 1), std::max<size_t>(n / 65536, 1));
    if (threads <= 1) return fn(0, lo, hi);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([=, &fn] { fn(t, lo + n * t / threads, lo + n * (t + 1) / threads); });
    for (std::thread &th : pool) th.join();
}

double nowSeconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// What the builder knows of a triangle.  The records themselves are partitioned (not an index array into side arrays):
// every pass over a node is then a sequential stream instead of a gather per triangle, which is what bounds the build
// of a multi-million-triangle scene.
struct Prim {
    Box box;
    float c[3];    // centroid of the box
    uint32_t idx;  // triangle number in the caller's order
};
static_assert(sizeof(Prim) == 40 && sizeof(Box) == 24, "the passes below read a Prim as three overlapping 16-byte loads");

// The passes over the records (bounds, binning) are the build of a multi-million-triangle scene, and std::fmin / std::fmax
// on three floats at a time is what they spend it on: each becomes a compare-unordered / min / blend sequence.  Four lanes at a
// time instead -- (mn.xyz, mx.x), (mx.xyz, c.x), (c.xyz, idx): lane 3 rides along and is never read.  _mm_min_ps(v, acc) is
// `v < acc ? v : acc`: a NaN coordinate leaves the accumulator alone exactly as fmin does (the accumulators start at +-inf
// and can never become NaN), so the boxes -- and the tree -- are what the scalar passes gave.
struct Box4 {
    __m128 mn = _mm_set1_ps(INFINITY), mx = _mm_set1_ps(-INFINITY);
    float halfArea() const {  // Box::halfArea, same operations in the same order
        const __m128 d = _mm_sub_ps(mx, mn);
        if ((_mm_movemask_ps(_mm_cmpge_ps(d, _mm_setzero_ps())) & 7) != 7) return 0.0f;
        float t[4];
        _mm_storeu_ps(t, _mm_mul_ps(d, _mm_shuffle_ps(d, d, _MM_SHUFFLE(3, 0, 2, 1))));  // dx dy, dy dz, dz dx
        return t[0] + t[1] + t[2];
    }
    void store(Box &b) const {
        float t[4];
        _mm_storeu_ps(t, mn);
        b.mn[0] = t[0]; b.mn[1] = t[1]; b.mn[2] = t[2];
        _mm_storeu_ps(t, mx);
        b.mx[0] = t[0]; b.mx[1] = t[1]; b.mx[2] = t[2];
    }
};

struct Bounds { Box box, cbox; };  // of a node's references: their boxes, their centroids
struct BuildTask {
    uint32_t self, lo, hi;
    int depth;
    bool in_tmp, have_bounds;  // see Builder::build
    Bounds bounds;
};
constexpr uint32_t kStableFrom = 1u << 17;  // nodes of at least this many references partition stably, out of place, on all threads

// One builder = one node array.  The scene-level builder walks the top of the tree (its big nodes bin in parallel)
// and, below `cutoff` triangles, defers subtrees as tasks; worker threads build those into private node arrays that
// are appended afterwards (disjoint ranges of `prims`, so the partitioning needs no locks).
struct Builder {
    Prim *prims = nullptr;            // the permutation being partitioned, as whole records
    Prim *tmp = nullptr;              // scratch of the same size (the big nodes partition out of place), or null
    RtRawVec<RtNode> nodes;
    int maxDepth = 0;
    uint32_t cutoff = 0;              // 0: build everything here
    int threads = 1;
    std::vector<BuildTask> tasks;

    void setNodeBox(RtNode &n, const Box &b) {
        // pad: the slab test in the kernel uses fma(mn, 1/d, -o/d); its rounding scales with |coordinate|
        for (int k = 0; k < 3; k++) {
            float e = 2e-6f * std::fmax(std::fmax(std::fabs(b.mn[k]), std::fabs(b.mx[k])), 1e-3f);
            n.mn[k] = b.mn[k] - e;
            n.mx[k] = b.mx[k] + e;
        }
    }

    // Builds the subtree of the references [lo, hi) into node `self`.  The references stand in prims -- or, below a node that
    // partitioned out of place (the huge nodes at the top), in tmp (in_tmp): such a node's children read where it wrote, and the
    // first descendant small enough to partition in place copies its range back.  pre: the bounds of the range if the parent's
    // partition pass gathered them on the way (same sets, so the same minima and maxima as a pass of its own).
    void build(uint32_t self, uint32_t lo, uint32_t hi, int depth, bool in_tmp = false, const Bounds *pre = nullptr) {
        const uint32_t count = hi - lo;
        if (cutoff && count <= cutoff) {
            tasks.push_back(BuildTask{self, lo, hi, depth, in_tmp, pre != nullptr, pre ? *pre : Bounds()});
            return;
        }
        maxDepth = std::max(maxDepth, depth);
        if (in_tmp && !(tmp && count >= kStableFrom)) {
            std::memcpy(prims + lo, tmp + lo, (size_t)count * sizeof(Prim));
            in_tmp = false;
        }
        Prim *const src = in_tmp ? tmp : prims;  // where this node reads; `cur`, below: where its children do
        Prim *cur = src;
        const int nt = count >= (1u << 17) ? threads : 1;  // only the few huge nodes at the top are worth threads
        Box box, cbox;
        auto boundsOf = [&](size_t a, size_t e, Box &b, Box &c) {
            Box4 b4, c4;
            for (size_t i = a; i < e; i++) {
                const float *p = reinterpret_cast<const float *>(src + i);
                const __m128 cc = _mm_loadu_ps(p + 6);
                b4.mn = _mm_min_ps(_mm_loadu_ps(p), b4.mn);
                b4.mx = _mm_max_ps(_mm_loadu_ps(p + 3), b4.mx);
                c4.mn = _mm_min_ps(cc, c4.mn);
                c4.mx = _mm_max_ps(cc, c4.mx);
            }
            b4.store(b);
            c4.store(c);
        };
        if (pre) {
            box = pre->box;
            cbox = pre->cbox;
        } else if (nt == 1) {  // (nearly every node: no heap traffic)
            boundsOf(lo, hi, box, cbox);
        } else {
            std::vector<Box> pb(nt), pc(nt);
            parallelRange(lo, hi, nt, [&](int t, size_t a, size_t e) { boundsOf(a, e, pb[t], pc[t]); });
            for (int t = 0; t < nt; t++) {
                box.grow(pb[t]);
                cbox.grow(pc[t]);
            }
        }
        setNodeBox(nodes[self], box);
        auto makeLeaf = [&]() {
            nodes[self].a = lo;
            nodes[self].b = count;
        };
        if (count <= 1) return makeLeaf();

        // binned SAH over the three axes, binned in one pass over the records
        float bestCost = INFINITY;
        int bestAxis = -1, bestSplit = -1;
        struct Bins {
            float mn[3][kBins][4], mx[3][kBins][4];  // (four lanes per box: see Box4)
            uint32_t cnt[3][kBins];
            Bins() { clear(); }  // (parallelRange may start fewer threads than there are partial sets: every set starts empty)
            void clear() {
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < kBins; b++) {
                        _mm_storeu_ps(mn[a][b], _mm_set1_ps(INFINITY));
                        _mm_storeu_ps(mx[a][b], _mm_set1_ps(-INFINITY));
                    }
                std::memset(cnt, 0, sizeof cnt);
            }
        };
        float cmin3[3], scale3[3];
        for (int axis = 0; axis < 3; axis++) {
            const float cext = cbox.mx[axis] - cbox.mn[axis];
            cmin3[axis] = cbox.mn[axis];
            scale3[axis] = cext > 0 ? kBins / cext : 0.0f;  // a flat axis lands in bin 0 and is skipped below
        }
        Bins all;
        std::vector<Bins> part(nt > 1 ? nt : 0);  // (per piece of the range; the stable partition below reads its counts again)
        {
            // bin = clamp((int)((c - cmin) * scale), 0, kBins - 1): the clamp is done on the float (same bin for every
            // finite product; a NaN product lands in bin 0 either way)
            const __m128 cmin4 = _mm_setr_ps(cmin3[0], cmin3[1], cmin3[2], 0.0f), scale4 = _mm_setr_ps(scale3[0], scale3[1], scale3[2], 0.0f);
            const __m128 zero4 = _mm_setzero_ps(), top4 = _mm_set1_ps((float)(kBins - 1));
            const __m128 lanes3 = _mm_castsi128_ps(_mm_setr_epi32(-1, -1, -1, 0));
            parallelRange(lo, hi, nt, [&](int tid, size_t a, size_t e) {
                Bins &p = nt > 1 ? part[tid] : all;
                for (size_t i = a; i < e; i++) {
                    const float *f = reinterpret_cast<const float *>(src + i);
                    const __m128 bmn = _mm_loadu_ps(f), bmx = _mm_loadu_ps(f + 3), cc = _mm_and_ps(_mm_loadu_ps(f + 6), lanes3);  // (lane 3 = idx: a denormal as a float)
                    const __m128 pos = _mm_min_ps(_mm_max_ps(_mm_mul_ps(_mm_sub_ps(cc, cmin4), scale4), zero4), top4);
                    alignas(16) int bin[4];
                    _mm_store_si128(reinterpret_cast<__m128i *>(bin), _mm_cvttps_epi32(pos));
                    for (int axis = 0; axis < 3; axis++) {
                        const int b = bin[axis];
                        _mm_storeu_ps(p.mn[axis][b], _mm_min_ps(bmn, _mm_loadu_ps(p.mn[axis][b])));
                        _mm_storeu_ps(p.mx[axis][b], _mm_max_ps(bmx, _mm_loadu_ps(p.mx[axis][b])));
                        p.cnt[axis][b]++;
                    }
                }
            });
            for (size_t tid = 0; tid < part.size(); tid++)
                for (int axis = 0; axis < 3; axis++)
                    for (int b = 0; b < kBins; b++) {
                        _mm_storeu_ps(all.mn[axis][b], _mm_min_ps(_mm_loadu_ps(part[tid].mn[axis][b]), _mm_loadu_ps(all.mn[axis][b])));
                        _mm_storeu_ps(all.mx[axis][b], _mm_max_ps(_mm_loadu_ps(part[tid].mx[axis][b]), _mm_loadu_ps(all.mx[axis][b])));
                        all.cnt[axis][b] += part[tid].cnt[axis][b];
                    }
        }
        // The sweeps visit the non-empty bins only: across an empty bin the boxes, the counts and so the cost are those of the
        // bin before, and a strict `<` never takes an equal cost -- the split found is the one a sweep over all bins finds.
        // Most nodes of the tree are small, and most of a small node's 3 x kBins bins are empty.
        for (int axis = 0; axis < 3; axis++) {
            if (!(cbox.mx[axis] - cbox.mn[axis] > 0)) continue;
            const uint32_t *cnt = all.cnt[axis];
            int full[kBins], m = 0;
            for (int b = 0; b < kBins; b++)
                if (cnt[b]) full[m++] = b;
            float rightArea[kBins];  // of the bins full[j..m)
            uint32_t rightCnt[kBins];
            Box4 acc;
            uint32_t c = 0;
            for (int j = m - 1; j > 0; j--) {
                const int b = full[j];
                acc.mn = _mm_min_ps(_mm_loadu_ps(all.mn[axis][b]), acc.mn);
                acc.mx = _mm_max_ps(_mm_loadu_ps(all.mx[axis][b]), acc.mx);
                c += cnt[b];
                rightArea[j] = acc.halfArea();
                rightCnt[j] = c;
            }
            acc = Box4();
            c = 0;
            for (int j = 0; j + 1 < m; j++) {
                const int b = full[j];
                acc.mn = _mm_min_ps(_mm_loadu_ps(all.mn[axis][b]), acc.mn);
                acc.mx = _mm_max_ps(_mm_loadu_ps(all.mx[axis][b]), acc.mx);
                c += cnt[b];
                const float cost = acc.halfArea() * c + rightArea[j + 1] * rightCnt[j + 1];
                if (cost < bestCost) {
                    bestCost = cost;
                    bestAxis = axis;
                    bestSplit = b;
                }
            }
        }
        // leaf when allowed and not worse than splitting (traversal step ~ 1 triangle test)
        if (count <= RT_LEAF_MAX) {
            const float leafCost = box.halfArea() * count;
            if (bestAxis < 0 || bestCost + RT_SAH_STEP_COST * box.halfArea() >= leafCost) return makeLeaf();
        }
        uint32_t mid;
        Bounds kidBounds[2];
        bool haveKidBounds = false;
        if (bestAxis >= 0) {
            const float cmin = cbox.mn[bestAxis], scale = kBins / (cbox.mx[bestAxis] - cbox.mn[bestAxis]);
            auto pred = [&](const Prim &t) {
                int b = (int)((t.c[bestAxis] - cmin) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                return b <= bestSplit;
            };
            if (tmp && count >= kStableFrom) {
                // The few huge nodes at the top: std::partition is one thread walking gigabytes (8 levels of the 50 M-triangle
                // build: most of its 2.4 s).  A STABLE partition out of place instead, on all threads, whose result does not
                // depend on how the range is cut into pieces, so that the tree stays the same for any thread count (which
                // algorithm a node gets depends on its size alone).  ONE pass: how many of a piece go left is already known
                // (the binning pass cut the range into the same pieces, and pred is "bin <= bestSplit" with the same
                // arithmetic); the children read the other array instead of a copy back; and their bounds are gathered while
                // the records pass through.  These levels are bound by memory: 6 passes over the array per level became 3.
                const int T = std::max(1, (int)std::min<size_t>((size_t)std::max(nt, 1), std::max<size_t>(count / 65536, 1)));
                std::vector<uint32_t> left((size_t)T + 1, 0);
                for (int t = 0; t < T; t++) {
                    const uint32_t *cnt = nt > 1 ? part[(size_t)t].cnt[bestAxis] : all.cnt[bestAxis];
                    uint32_t c = 0;
                    for (int b = 0; b <= bestSplit; b++) c += cnt[b];
                    left[(size_t)t + 1] = left[(size_t)t] + c;
                }
                const uint32_t L = left[(size_t)T];
                Prim *const dst = in_tmp ? prims : tmp;
                std::vector<Bounds> kb(2 * (size_t)T);
                parallelRange(lo, hi, nt, [&](int t, size_t a, size_t e) {
                    Prim *dl = dst + lo + left[(size_t)t], *dr = dst + lo + L + ((a - lo) - left[(size_t)t]);
                    Box4 lb, lc, rb, rc;
                    for (size_t i = a; i < e; i++) {
                        const float *f = reinterpret_cast<const float *>(src + i);
                        const __m128 bmn = _mm_loadu_ps(f), bmx = _mm_loadu_ps(f + 3), cc = _mm_loadu_ps(f + 6);
                        if (pred(src[i])) {
                            *dl++ = src[i];
                            lb.mn = _mm_min_ps(bmn, lb.mn); lb.mx = _mm_max_ps(bmx, lb.mx);
                            lc.mn = _mm_min_ps(cc, lc.mn); lc.mx = _mm_max_ps(cc, lc.mx);
                        } else {
                            *dr++ = src[i];
                            rb.mn = _mm_min_ps(bmn, rb.mn); rb.mx = _mm_max_ps(bmx, rb.mx);
                            rc.mn = _mm_min_ps(cc, rc.mn); rc.mx = _mm_max_ps(cc, rc.mx);
                        }
                    }
                    lb.store(kb[2 * (size_t)t].box); lc.store(kb[2 * (size_t)t].cbox);
                    rb.store(kb[2 * (size_t)t + 1].box); rc.store(kb[2 * (size_t)t + 1].cbox);
                });
                for (int t = 0; t < T; t++)
                    for (int k = 0; k < 2; k++) {
                        kidBounds[k].box.grow(kb[2 * (size_t)t + k].box);
                        kidBounds[k].cbox.grow(kb[2 * (size_t)t + k].cbox);
                    }
                haveKidBounds = true;
                cur = dst;
                mid = lo + L;
            } else {
                auto it = std::partition(prims + lo, prims + hi, pred);
                mid = (uint32_t)(it - prims);
            }
        } else {
            mid = lo;  // all centroids coincide
        }
        if (mid == lo || mid == hi) {  // degenerate: split by count (keeps depth bounded)
            mid = lo + count / 2;
            haveKidBounds = false;
        }
        // deep trees cost LDS stack: past a budget fall back to median splits, which bound the remaining depth
        if (depth >= 40 && bestAxis >= 0) {
            const int axis = bestAxis;
            mid = lo + count / 2;
            std::nth_element(cur + lo, cur + mid, cur + hi,
                             [&](const Prim &a, const Prim &b) { return a.c[axis] < b.c[axis]; });
            haveKidBounds = false;
        }
        const uint32_t left = (uint32_t)nodes.size();
        nodes.emplace_back();
        nodes.emplace_back();
        std::memset(&nodes[left], 0, 2 * sizeof(RtNode));
        nodes[self].a = left;
        nodes[self].b = 0;
        build(left, lo, mid, depth + 1, cur == tmp, haveKidBounds ? &kidBounds[0] : nullptr);
        build(left + 1, mid, hi, depth + 1, cur == tmp, haveKidBounds ? &kidBounds[1] : nullptr);
    }
};

// candidate order of the reference's per-mesh tree (splitMesh + ctor, MeshLoader.h:218-263; walk :265-284)
void refOrder(const float *v0, const float *v1, const float *v2, std::vector<uint32_t> &mesh, int a, int depth,
              uint32_t &next, uint32_t *rank) {
    if (mesh.size() <= 100 || depth > 48) {
        for (uint32_t t : mesh) rank[t] = next++;
        return;
    }
    float c = 0;
    for (uint32_t t : mesh) {
        c += v0[3 * (size_t)t + a];
        c += v1[3 * (size_t)t + a];
        c += v2[3 * (size_t)t + a];
    }
    c /= mesh.size() * 3;
    std::vector<uint32_t> left, right;
    for (uint32_t t : mesh) {
        const bool isLeft = v0[3 * (size_t)t + a] < c || v1[3 * (size_t)t + a] < c || v2[3 * (size_t)t + a] < c;
        (isLeft ? left : right).push_back(t);
    }
    std::vector<uint32_t>().swap(mesh);
    refOrder(v0, v1, v2, left, (a + 1) % 3, depth + 1, next, rank);
    refOrder(v0, v1, v2, right, (a + 1) % 3, depth + 1, next, rank);
}


// ---- collapse to 4 children per node + quantisation -------------------------------------------------------------
// Level by level: the 4-wide nodes of one level are independent of each other, so every level is two parallel passes over
// its nodes -- (1) choose the up to four children of each (open the inner child with the largest surface until there are
// four), count the inner ones, prefix-sum the counts: that IS the breadth-first numbering (the next level's nodes in parent
// order, children in slot order: the top levels, which every ray visits, become a prefix of the array and siblings stay
// close); (2) quantise the child boxes and write both node forms.  The tree does not depend on the thread count.
// (Round 2 emitted the nodes depth first on one thread and renumbered afterwards: 2.8 s of the 50 M-triangle commit.)
struct Collapser {
    const RtRawVec<RtNode> &bn;
    int threads = 1;
    RtRawVec<RtNode4> out;        // (fill() writes every byte of both forms)
    RtRawVec<RtNodeP> outP;       // same indices as `out`
    int maxDepth = 0, maxStack = 0;
    float bound = 0.0f;

    static float halfArea(const RtNode &n) {
        const float dx = n.mx[0] - n.mn[0], dy = n.mx[1] - n.mn[1], dz = n.mx[2] - n.mn[2];
        return dx * dy + dy * dz + dz * dx;
    }
    static uint32_t leafRef(const RtNode &n) { return 0x80000000u | ((n.b - 1u) << 28) | n.a; }  // slot; rebased later

    struct Item { uint32_t b; int pending; };  // binary inner node to emit; stack entries that can be live above it
    struct Kids { uint32_t k[4]; int nk, inner; };

    Kids choose(uint32_t b) const {
        Kids r;
        r.k[0] = bn[b].a; r.k[1] = bn[b].a + 1; r.k[2] = r.k[3] = 0;
        int nk = 2;
        while (nk < 4) {  // open the inner child with the largest surface
            int pick = -1;
            float best = -1.0f;
            for (int k = 0; k < nk; k++)
                if (bn[r.k[k]].b == 0 && halfArea(bn[r.k[k]]) > best) {
                    best = halfArea(bn[r.k[k]]);
                    pick = k;
                }
            if (pick < 0) break;
            const uint32_t open = r.k[pick];
            for (int k = nk; k > pick + 1; k--) r.k[k] = r.k[k - 1];
            r.k[pick] = bn[open].a;
            r.k[pick + 1] = bn[open].a + 1;
            nk++;
        }
        // Child slots in order of surface, smallest first.  Closest-hit walks sort the hit children by distance, so the
        // slot order means nothing to them; any-hit walks enter the first hit child and pop the others in slot order,
        // and the small boxes are where the geometry is dense: an occluded shadow ray finds its occluder sooner
        // (k_any 132 -> 128 ms on the 4K competition frame; largest first: 138).
        for (int i = 1; i < nk; i++)
            for (int j = i; j > 0 && halfArea(bn[r.k[j]]) < halfArea(bn[r.k[j - 1]]); j--) std::swap(r.k[j], r.k[j - 1]);
        r.nk = nk;
        r.inner = 0;
        for (int k = 0; k < nk; k++) r.inner += bn[r.k[k]].b == 0 ? 1 : 0;
        return r;
    }

    // writes node `self`; its inner children get the indices first_child, first_child + 1, ... in slot order
    void fill(uint32_t self, const Kids &kd, uint32_t first_child, float &bnd) {
        const int nk = kd.nk;
        const uint32_t *kids = kd.k;
        // frame = union of the children's (padded) boxes
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < nk; k++)
            for (int a = 0; a < 3; a++) {
                lo[a] = std::fmin(lo[a], bn[kids[k]].mn[a]);
                hi[a] = std::fmax(hi[a], bn[kids[k]].mx[a]);
            }
        double scale[3];
        for (int a = 0; a < 3; a++) {
            bnd = std::fmax(bnd, std::fmax(std::fabs(lo[a]), std::fabs(hi[a])));
            const double ext = (double)hi[a] - (double)lo[a];
            int e = -100;
            if (ext > 0) e = std::max(-100, (int)std::ceil(std::log2(ext / 255.0)));
            while (ext / std::ldexp(1.0, e) > 255.0) e++;
            scale[a] = std::ldexp(1.0, e);
        }
        uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
        uint32_t child[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        for (int k = 0; k < nk; k++) {
            const RtNode &c = bn[kids[k]];
            for (int a = 0; a < 3; a++) {
                double l = std::floor(((double)c.mn[a] - (double)lo[a]) / scale[a]);
                double h = std::ceil(((double)c.mx[a] - (double)lo[a]) / scale[a]);
                l = std::fmin(std::fmax(l, 0.0), 255.0);
                h = std::fmin(std::fmax(h, 0.0), 255.0);
                qlo[a] |= (uint32_t)l << (8 * k);
                qhi[a] |= (uint32_t)h << (8 * k);
            }
        }
        uint32_t next = first_child;
        for (int k = 0; k < nk; k++) {
            const RtNode &c = bn[kids[k]];
            child[k] = c.b ? leafRef(c) : next++;
        }
        RtNode4 &n = out[self];
        n.origin[0] = lo[0]; n.origin[1] = lo[1]; n.origin[2] = lo[2];
        n.scale_x = (float)scale[0]; n.scale_y = (float)scale[1]; n.scale_z = (float)scale[2];
        for (int k = 0; k < 4; k++) n.child[k] = child[k];
        RtNodeP &np = outP[self];
        std::memset(&np, 0, sizeof np);
        for (int k = 0; k < 4; k++) {
            np.child[k] = child[k];
            for (int a = 0; a < 3; a++) {  // empty slots keep an inverted box (never hit; the reference word says so too)
                np.lo[a][k] = k < nk ? bn[kids[k]].mn[a] : INFINITY;
                np.hi[a][k] = k < nk ? bn[kids[k]].mx[a] : -INFINITY;
            }
        }
        n.qlo[0] = qlo[0]; n.qlo[1] = qlo[1]; n.qlo[2] = qlo[2];
        n.qhi_x = qhi[0]; n.qhi_y = qhi[1]; n.qhi_z = qhi[2];
    }

    void run(uint32_t root) {
        out.reserve(bn.size() / 2 + 1);   // (every 4-wide node stands for at least one binary inner node)
        outP.reserve(bn.size() / 2 + 1);
        RtRawVec<Item> level{Item{root, 0}}, nextLevel;
        RtRawVec<Kids> kids;
        RtRawVec<uint32_t> firstChild;
        size_t base = 0;
        for (int depth = 1; !level.empty(); depth++) {
            const size_t n = level.size();
            maxDepth = depth;
            out.resize(base + n);
            outP.resize(base + n);
            kids.resize(n);
            firstChild.resize(n + 1);
            parallelRange(0, n, threads, [&](int, size_t a, size_t e) {
                for (size_t i = a; i < e; i++) kids[i] = choose(level[i].b);
            });
            uint32_t total = 0;
            for (size_t i = 0; i < n; i++) {  // (a serial scan: 12 M additions for the 50 M-triangle tree, all levels together)
                firstChild[i] = total;
                total += (uint32_t)kids[i].inner;
            }
            firstChild[n] = total;
            nextLevel.resize(total);
            const int nt = std::max(threads, 1);
            std::vector<float> tb((size_t)nt, 0.0f);
            std::vector<int> ts((size_t)nt, 0);
            parallelRange(0, n, threads, [&](int tid, size_t a, size_t e) {
                float bnd = 0.0f;
                int stk = 0;
                for (size_t i = a; i < e; i++) {
                    const Kids &kd = kids[i];
                    const int pend = level[i].pending + kd.nk - 1;
                    stk = std::max(stk, pend);
                    fill((uint32_t)(base + i), kd, (uint32_t)(base + n + firstChild[i]), bnd);
                    uint32_t at = firstChild[i];
                    for (int k = 0; k < kd.nk; k++)
                        if (bn[kd.k[k]].b == 0) nextLevel[at++] = Item{kd.k[k], pend};
                }
                tb[(size_t)tid] = bnd;
                ts[(size_t)tid] = stk;
            });
            for (int t = 0; t < nt; t++) {
                bound = std::fmax(bound, tb[(size_t)t]);
                maxStack = std::max(maxStack, ts[(size_t)t]);
            }
            base += n;
            level.swap(nextLevel);
        }
    }
};

// ---- pre-splitting of large triangles (early split clipping) -------------------------------------------------------
// A tree over whole triangles puts a large triangle into a leaf whose box is as large as the triangle's own, high up in
// the tree: every ray that crosses that box tests it.  In the competition scene 100 of 68 202 triangles drew half of all
// triangle tests (scripts/gpu_r3_tri_hist.py).  So before the SAH build the largest triangle boxes are cut in two at the
// middle of their longest axis, the triangle clipped against the cut and each half bounded again, largest first, until the
// boxes are down to a few times the average or a budget of extra references is spent.  The tree is then built over the
// REFERENCES (clipped box + triangle number): a triangle may sit in several leaves, its 64-byte record is duplicated per
// slot (rt_commit), and the exact triangle test is what it was -- the boxes only cull, and together the pieces' boxes
// cover the triangle.  A ray may meet the same triangle twice: the second test finds the same distance and the
// strict-minimum / tie rule leaves the first hit in place.
struct SplitRef {
    double v[10][3];  // the piece: a convex polygon (a triangle clipped by up to 6 planes has at most 9 corners)
    int nv;
    Box box;
    uint32_t tri;
};
// outward rounding of a double interval to floats
void growOut(Box &b, const double *p) {
    for (int k = 0; k < 3; k++) {
        float lo = (float)p[k], hi = (float)p[k];
        if ((double)lo > p[k]) lo = std::nextafterf(lo, -INFINITY);
        if ((double)hi < p[k]) hi = std::nextafterf(hi, INFINITY);
        b.mn[k] = std::fmin(b.mn[k], lo);
        b.mx[k] = std::fmax(b.mx[k], hi);
    }
}
// the part of polygon `in` with coordinate `axis` <= s (side 0) or >= s (side 1); corners on the plane belong to both
void clipPoly(const SplitRef &in, int axis, double s, int side, SplitRef &out) {
    out.nv = 0;
    out.tri = in.tri;
    out.box = Box();
    for (int i = 0; i < in.nv; i++) {
        const double *a = in.v[i], *b = in.v[(i + 1) % in.nv];
        const double da = side ? s - a[axis] : a[axis] - s, db = side ? s - b[axis] : b[axis] - s;  // <= 0: inside
        if (da <= 0) {
            if (out.nv >= 10) { out.nv = 0; return; }  // (cannot happen for a triangle inside a box: 9 corners at most; the piece then stays whole)
            std::memcpy(out.v[out.nv++], a, 3 * sizeof(double));
        }
        if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
            const double t = da / (da - db);
            double x[3];
            for (int k = 0; k < 3; k++) x[k] = a[k] + t * (b[k] - a[k]);
            x[axis] = s;
            if (out.nv >= 10) { out.nv = 0; return; }
            std::memcpy(out.v[out.nv++], x, 3 * sizeof(double));
        }
    }
    for (int i = 0; i < out.nv; i++) growOut(out.box, out.v[i]);
    // the piece lies inside its parent's box (rounding must not let it grow out of it)
    for (int k = 0; k < 3; k++) {
        out.box.mn[k] = std::fmax(out.box.mn[k], in.box.mn[k]);
        out.box.mx[k] = std::fmin(out.box.mx[k], in.box.mx[k]);
    }
}
// Cuts piece `cur` until its box is down to `floor_area` (or 16 cuts deep); fn(piece) for every final piece, in a fixed
// (depth-first, low side first) order.  A cut that only grazes the piece leaves it whole.
template <class F>
void splitPiece(const SplitRef &cur, double floor_area, int depth, F &fn) {
    const float area = cur.box.halfArea();
    if (!std::isfinite(area) || !((double)area > floor_area) || depth >= 16) return fn(cur);
    int axis = 0;
    float ext = -1.0f;
    for (int k = 0; k < 3; k++)
        if (cur.box.mx[k] - cur.box.mn[k] > ext) {
            ext = cur.box.mx[k] - cur.box.mn[k];
            axis = k;
        }
    if (!(ext > 0)) return fn(cur);
    const double s = 0.5 * ((double)cur.box.mn[axis] + (double)cur.box.mx[axis]);
    SplitRef lo, hi;
    clipPoly(cur, axis, s, 0, lo);
    clipPoly(cur, axis, s, 1, hi);
    if (lo.nv < 3 || hi.nv < 3) return fn(cur);
    splitPiece(lo, floor_area, depth + 1, fn);
    splitPiece(hi, floor_area, depth + 1, fn);
}
// Replaces the boxes of the large triangles of `prims` by the boxes of their pieces; returns the number of references
// added.  A piece is cut while its box's surface exceeds min_ratio x the average triangle box; if that would add more than
// budget_frac x n references the floor is raised (x 1.5 at a time) until it does not.  Every triangle is cut on its own: two
// parallel passes (count, then write: the first piece keeps the triangle's slot, the others follow the n triangles in
// triangle order), the same references for any thread count.
size_t presplit(RtRawVec<Prim> &prims, const float *v0, const float *v1, const float *v2, double budget_frac, double min_ratio, int threads) {
    const size_t n = prims.size();
    double sum = 0.0;
    size_t finite = 0;
    for (const Prim &p : prims) {
        const float a = p.box.halfArea();
        if (std::isfinite(a)) {
            sum += a;
            finite++;
        }
    }
    if (!finite || !(sum > 0)) return 0;
    double floor_area = min_ratio * sum / (double)finite;
    const size_t budget = (size_t)(budget_frac * (double)n) + 16;
    auto whole = [&](size_t t, SplitRef &r) {
        r.nv = 3;
        r.tri = (uint32_t)t;
        r.box = prims[t].box;
        const float *vs[3] = {v0 + 3 * t, v1 + 3 * t, v2 + 3 * t};
        for (int i = 0; i < 3; i++)
            for (int k = 0; k < 3; k++) r.v[i][k] = (double)vs[i][k];
    };
    std::vector<uint32_t> extra(n + 1, 0);  // pieces beyond the first, per triangle; then their first slot
    size_t added = 0;
    for (int attempt = 0; attempt < 64; attempt++) {
        parallelRange(0, n, threads, [&](int, size_t a, size_t e) {
            for (size_t t = a; t < e; t++) {
                const float ar = prims[t].box.halfArea();
                uint32_t pieces = 1;
                if (std::isfinite(ar) && (double)ar > floor_area) {
                    SplitRef r;
                    whole(t, r);
                    pieces = 0;
                    auto count = [&](const SplitRef &) { pieces++; };
                    splitPiece(r, floor_area, 0, count);
                }
                extra[t] = pieces - 1;
            }
        });
        added = 0;
        for (size_t t = 0; t < n; t++) added += extra[t];
        if (added <= budget) break;
        floor_area *= 1.5;
    }
    if (!added || added > budget) return 0;
    size_t at = n;
    for (size_t t = 0; t < n; t++) {  // exclusive scan: where a triangle's further pieces go
        const uint32_t c = extra[t];
        extra[t] = (uint32_t)at;
        at += c;
    }
    prims.resize(n + added);
    Prim *P = prims.data();
    parallelRange(0, n, threads, [&](int, size_t a, size_t e) {
        for (size_t t = a; t < e; t++) {
            const float ar = P[t].box.halfArea();
            if (!(std::isfinite(ar) && (double)ar > floor_area)) continue;
            SplitRef r;
            whole(t, r);
            uint32_t k = 0, next = extra[t];
            auto write = [&](const SplitRef &pc) {
                Prim &p = k == 0 ? P[t] : P[next++];
                k++;
                p.box = pc.box;
                for (int c = 0; c < 3; c++) p.c[c] = 0.5f * (pc.box.mn[c] + pc.box.mx[c]);
                p.idx = pc.tri;
            };
            splitPiece(r, floor_area, 0, write);
        }
    });
    return added;
}

}  // namespace

void rtBuildBvh(int64_t n, const float *v0, const float *v1, const float *v2, RtBvh &out) {
    out.nodes.clear();
    out.nodes_p.clear();
    out.order.clear();
    out.depth = 0;
    out.max_stack = 0;
    out.bound = 0.0f;
    out.binary_nodes = 0;
    out.root_ref = 0xFFFFFFFFu;
    if (n <= 0) return;
    const bool verbose = std::getenv("RT_VERBOSE") != nullptr;
    const double t0 = nowSeconds();
    const int threads = buildThreads();
    RtRawVec<Prim> prims((size_t)n);
    parallelRange(0, (size_t)n, threads, [&](int, size_t a, size_t e) {
        for (size_t t = a; t < e; t++) {
            Prim &pr = prims[t];
            pr.box = Box();
            pr.box.grow(v0 + 3 * t);
            pr.box.grow(v1 + 3 * t);
            pr.box.grow(v2 + 3 * t);
            for (int k = 0; k < 3; k++) pr.c[k] = 0.5f * (pr.box.mn[k] + pr.box.mx[k]);
            pr.idx = (uint32_t)t;
        }
    });
    // large triangles are cut into pieces first (above): scenes of up to 4 Mi triangles, at most twice the references
    // (RT_PRESPLIT=0 turns it off; RT_PRESPLIT_BUDGET / RT_PRESPLIT_MIN: sweep knobs)
    size_t split_added = 0;
    {
        double budget = 1.0, min_ratio = 0.5;  // (competition frame: 434 ms without, 402 at ratio 4, 397 at 1, 390 at 0.5, 394 at 0.25)
        int64_t max_tris = (int64_t)1 << 22;
        if (const char *e = std::getenv("RT_PRESPLIT_MAX_TRIS")) max_tris = std::atoll(e);
        bool on = n > RT_LEAF_MAX && n <= max_tris;  // (a scene that fits one leaf stays one leaf)
        if (const char *e = std::getenv("RT_PRESPLIT")) on = on && std::atoi(e) != 0;
        if (const char *e = std::getenv("RT_PRESPLIT_BUDGET")) budget = std::max(0.0, std::atof(e));
        if (const char *e = std::getenv("RT_PRESPLIT_MIN")) min_ratio = std::max(0.05, std::atof(e));
        if (on && budget > 0) split_added = presplit(prims, v0, v1, v2, budget, min_ratio, threads);
    }
    const int64_t nrefs = (int64_t)prims.size();
    out.references = nrefs;
    // (raw memory: a vector would construct 50 M records on one thread first; every record is written before it is read)
    struct Raw {
        Prim *p = nullptr;
        ~Raw() { std::free(p); }
    } scratch;
    if ((uint64_t)nrefs >= kStableFrom) scratch.p = static_cast<Prim *>(std::malloc((size_t)nrefs * sizeof(Prim)));
    Builder b;
    b.prims = prims.data();
    b.tmp = scratch.p;
    b.threads = threads;
    b.cutoff = (threads > 1 && nrefs >= (1 << 18)) ? (uint32_t)std::max<int64_t>(nrefs / (16 * (int64_t)threads), 4096) : 0u;
    b.nodes.reserve(2 * (size_t)nrefs + 2);
    b.nodes.resize(2);  // [0] unused, [1] root
    std::memset(b.nodes.data(), 0, 2 * sizeof(RtNode));
    b.build(1, 0, (uint32_t)nrefs, 0);
    const double t1 = nowSeconds();
    if (!b.tasks.empty()) {  // subtrees in parallel, each into a private node array ([0] = the subtree's root)
        std::vector<Builder> sub(b.tasks.size());
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++)
            pool.emplace_back([&] {
                for (size_t k; (k = next.fetch_add(1)) < b.tasks.size();) {
                    Builder &w = sub[k];
                    w.prims = b.prims;
                    w.tmp = b.tmp;  // (tasks own disjoint ranges of both arrays)
                    w.nodes.resize(1);
                    std::memset(w.nodes.data(), 0, sizeof(RtNode));
                    w.build(0, b.tasks[k].lo, b.tasks[k].hi, b.tasks[k].depth, b.tasks[k].in_tmp, b.tasks[k].have_bounds ? &b.tasks[k].bounds : nullptr);
                }
            });
        for (std::thread &th : pool) th.join();
        // append, in task order: local node j >= 1 of subtree k becomes global node base[k] + j - 1
        std::vector<size_t> base(sub.size() + 1);
        base[0] = b.nodes.size();
        for (size_t k = 0; k < sub.size(); k++) base[k + 1] = base[k] + sub[k].nodes.size() - 1;
        b.nodes.resize(base[sub.size()]);
        next = 0;
        pool.clear();
        for (int t = 0; t < threads; t++)
            pool.emplace_back([&] {
                for (size_t k; (k = next.fetch_add(1)) < sub.size();) {
                    const uint32_t shift = (uint32_t)base[k] - 1;
                    auto fix = [&](RtNode nd) {
                        if (nd.b == 0) nd.a += shift;
                        return nd;
                    };
                    b.nodes[b.tasks[k].self] = fix(sub[k].nodes[0]);
                    RtNode *dst = b.nodes.data() + base[k] - 1;
                    for (size_t j = 1; j < sub[k].nodes.size(); j++) dst[j] = fix(sub[k].nodes[j]);
                    RtRawVec<RtNode>().swap(sub[k].nodes);
                }
            });
        for (std::thread &th : pool) th.join();
        for (size_t k = 0; k < sub.size(); k++) b.maxDepth = std::max(b.maxDepth, sub[k].maxDepth);
    }
    const double t2 = nowSeconds();
    out.order.resize((size_t)nrefs);
    parallelRange(0, (size_t)nrefs, threads, [&](int, size_t a, size_t e) {
        for (size_t i = a; i < e; i++) out.order[i] = prims[i].idx;
    });
    RtRawVec<Prim>().swap(prims);
    out.binary_nodes = (int64_t)b.nodes.size();
    const RtNode &root = b.nodes[1];
    if (root.b) {  // the whole scene is one leaf
        out.root_ref = Collapser::leafRef(root);
        out.depth = 0;
        out.max_stack = 0;
        for (int a = 0; a < 3; a++) out.bound = std::fmax(out.bound, std::fmax(std::fabs(root.mn[a]), std::fabs(root.mx[a])));
    } else {
        Collapser c{b.nodes};
        c.threads = threads;
        c.run(1);
        out.nodes = std::move(c.out);
        out.nodes_p = std::move(c.outP);
        out.root_ref = 0;
        out.depth = c.maxDepth;
        out.max_stack = c.maxStack;
        out.bound = c.bound;
    }
    // triangle records follow the nodes in the record array: rebase the leaf references
    const uint32_t base = (uint32_t)out.nodes.size();
    parallelRange(0, out.nodes.size(), threads, [&](int, size_t a, size_t e) {
        for (size_t i = a; i < e; i++)
            for (int k = 0; k < 4; k++) {
                uint32_t &ch = out.nodes[i].child[k];
                if (ch != 0xFFFFFFFFu && (ch & 0x80000000u)) ch += base;
                out.nodes_p[i].child[k] = ch;
            }
    });
    if (out.root_ref & 0x80000000u) out.root_ref += base;
    if (verbose)
        std::fprintf(stderr, "[rt_bvh] %lld triangles (+ %zu references from pre-splitting), %d threads: top %.2f s, subtrees %.2f s (%zu tasks), collapse+quantise+renumber %.2f s\n",
                     (long long)n, split_added, threads, t1 - t0, t2 - t1, b.tasks.size(), nowSeconds() - t2);
}

int rtBuildThreads() { return buildThreads(); }
void rtParallelFor(size_t n, void (*fn)(size_t, size_t, void *), void *arg) {
    parallelRange(0, n, buildThreads(), [&](int, size_t a, size_t e) { fn(a, e, arg); });
}

void rtReferenceOrder(int64_t n, const float *v0, const float *v1, const float *v2, const int32_t *object_id,
                      uint32_t *rank) {
    // meshes are independent: the ranks of mesh m are first(m) .. first(m) + count(m) - 1
    std::vector<std::pair<int64_t, int64_t>> meshes;
    for (int64_t k = 0; k < n;) {
        int64_t e = k;
        while (e < n && object_id[e] == object_id[k]) e++;
        meshes.emplace_back(k, e);
        k = e;
    }
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t m; (m = next.fetch_add(1)) < meshes.size();) {
            std::vector<uint32_t> mesh((size_t)(meshes[m].second - meshes[m].first));
            std::iota(mesh.begin(), mesh.end(), (uint32_t)meshes[m].first);
            uint32_t r = (uint32_t)meshes[m].first;
            refOrder(v0, v1, v2, mesh, 0, 0, r, rank);
        }
    };
    const int threads = n >= (1 << 18) ? buildThreads() : 1;
    if (threads <= 1) return work();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(work);
    for (std::thread &th : pool) th.join();
}
