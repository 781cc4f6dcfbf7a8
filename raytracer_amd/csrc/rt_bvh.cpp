// rt_bvh.cpp -- host-side builder of the linear BVH the traversal kernels walk (rt_kernels.hip: traverseBvh).
//
// Replaces the reference's per-mesh pointer tree (bvh_node, MeshLoader.h:211-285: split at the mean vertex
// coordinate, axis cycling with depth, leaves of <= 100 triangles that are copied by value per visit) with ONE
// binned-SAH tree over every triangle of the scene, stored as 32-byte nodes in a flat array:
//   * node 0 is padding, the root is node 1, and the two children of an inner node are adjacent at an even index,
//     so a sibling pair is exactly one 64-byte line and one fetch per traversal step;
//   * leaves hold <= RT_LEAF_MAX triangles that are contiguous in the reordered triangle array.
// The reference's tree is not reproduced -- only its observable result is (closest / any hit), plus the ORDER in
// which it hands out candidates, because that order decides exact distance ties (MeshLoader.h:399-404):
// rtReferenceOrder() recomputes that order so the kernel can break ties identically.
#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

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

constexpr int kBins = 16;

struct Builder {
    const float *v0, *v1, *v2;
    std::vector<Box> triBox;
    std::vector<float> centroid;  // 3 per triangle
    std::vector<uint32_t> index;  // permutation being partitioned
    std::vector<RtNode> nodes;
    int maxDepth = 0;

    void setNodeBox(RtNode &n, const Box &b) {
        // pad: the slab test in the kernel uses fma(mn, 1/d, -o/d); its rounding scales with |coordinate|
        for (int k = 0; k < 3; k++) {
            float e = 2e-6f * std::fmax(std::fmax(std::fabs(b.mn[k]), std::fabs(b.mx[k])), 1e-3f);
            n.mn[k] = b.mn[k] - e;
            n.mx[k] = b.mx[k] + e;
        }
    }

    // builds the subtree of triangles index[lo, hi) into node `self`
    void build(uint32_t self, uint32_t lo, uint32_t hi, int depth) {
        maxDepth = std::max(maxDepth, depth);
        Box box, cbox;
        for (uint32_t i = lo; i < hi; i++) {
            box.grow(triBox[index[i]]);
            cbox.grow(&centroid[3 * (size_t)index[i]]);
        }
        setNodeBox(nodes[self], box);
        const uint32_t count = hi - lo;
        auto makeLeaf = [&]() {
            nodes[self].a = lo;
            nodes[self].b = count;
        };
        if (count <= 1) return makeLeaf();

        // binned SAH over the three axes
        float bestCost = INFINITY;
        int bestAxis = -1, bestSplit = -1;
        for (int axis = 0; axis < 3; axis++) {
            const float cmin = cbox.mn[axis], cext = cbox.mx[axis] - cbox.mn[axis];
            if (!(cext > 0)) continue;
            Box bins[kBins];
            uint32_t cnt[kBins] = {0};
            const float scale = kBins / cext;
            for (uint32_t i = lo; i < hi; i++) {
                const uint32_t t = index[i];
                int b = (int)((centroid[3 * (size_t)t + axis] - cmin) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                bins[b].grow(triBox[t]);
                cnt[b]++;
            }
            float rightArea[kBins];
            uint32_t rightCnt[kBins];
            Box acc;
            uint32_t c = 0;
            for (int b = kBins - 1; b > 0; b--) {
                acc.grow(bins[b]);
                c += cnt[b];
                rightArea[b] = acc.halfArea();
                rightCnt[b] = c;
            }
            acc = Box();
            c = 0;
            for (int b = 0; b < kBins - 1; b++) {
                acc.grow(bins[b]);
                c += cnt[b];
                if (c == 0 || rightCnt[b + 1] == 0) continue;
                const float cost = acc.halfArea() * c + rightArea[b + 1] * rightCnt[b + 1];
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
            if (bestAxis < 0 || bestCost + box.halfArea() >= leafCost) return makeLeaf();
        }
        uint32_t mid;
        if (bestAxis >= 0) {
            const float cmin = cbox.mn[bestAxis], scale = kBins / (cbox.mx[bestAxis] - cbox.mn[bestAxis]);
            auto it = std::partition(index.begin() + lo, index.begin() + hi, [&](uint32_t t) {
                int b = (int)((centroid[3 * (size_t)t + bestAxis] - cmin) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                return b <= bestSplit;
            });
            mid = (uint32_t)(it - index.begin());
        } else {
            mid = lo;  // all centroids coincide
        }
        if (mid == lo || mid == hi) {  // degenerate: split by count (keeps depth bounded)
            mid = lo + count / 2;
        }
        // deep trees cost LDS stack: past a budget fall back to median splits, which bound the remaining depth
        if (depth >= 40 && bestAxis >= 0) {
            const int axis = bestAxis;
            mid = lo + count / 2;
            std::nth_element(index.begin() + lo, index.begin() + mid, index.begin() + hi, [&](uint32_t a, uint32_t b) {
                return centroid[3 * (size_t)a + axis] < centroid[3 * (size_t)b + axis];
            });
        }
        const uint32_t left = (uint32_t)nodes.size();
        nodes.emplace_back();
        nodes.emplace_back();
        std::memset(&nodes[left], 0, 2 * sizeof(RtNode));
        nodes[self].a = left;
        nodes[self].b = 0;
        build(left, lo, mid, depth + 1);
        build(left + 1, mid, hi, depth + 1);
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

}  // namespace

void rtBuildBvh(int64_t n, const float *v0, const float *v1, const float *v2, RtBvh &out) {
    out.nodes.clear();
    out.order.clear();
    out.depth = 0;
    out.root_ref = 0xFFFFFFFFu;
    if (n <= 0) return;
    Builder b;
    b.v0 = v0; b.v1 = v1; b.v2 = v2;
    b.triBox.resize(n);
    b.centroid.resize(3 * (size_t)n);
    b.index.resize(n);
    std::iota(b.index.begin(), b.index.end(), 0u);
    for (int64_t t = 0; t < n; t++) {
        Box &bx = b.triBox[t];
        bx.grow(v0 + 3 * t);
        bx.grow(v1 + 3 * t);
        bx.grow(v2 + 3 * t);
        for (int k = 0; k < 3; k++) b.centroid[3 * t + k] = 0.5f * (bx.mn[k] + bx.mx[k]);
    }
    b.nodes.reserve(2 * (size_t)n + 2);
    b.nodes.resize(2);  // [0] padding so that sibling pairs start at even indices, [1] root
    std::memset(b.nodes.data(), 0, 2 * sizeof(RtNode));
    b.build(1, 0, (uint32_t)n, 0);
    out.nodes = std::move(b.nodes);
    out.order = std::move(b.index);
    out.depth = b.maxDepth;
    const RtNode &root = out.nodes[1];
    out.root_ref = root.b == 0 ? root.a : (0x80000000u | ((root.b - 1u) << 28) | root.a);
}

void rtReferenceOrder(int64_t n, const float *v0, const float *v1, const float *v2, const int32_t *object_id,
                      uint32_t *rank) {
    uint32_t next = 0;
    int64_t k = 0;
    while (k < n) {
        int64_t e = k;
        while (e < n && object_id[e] == object_id[k]) e++;
        std::vector<uint32_t> mesh((size_t)(e - k));
        std::iota(mesh.begin(), mesh.end(), (uint32_t)k);
        refOrder(v0, v1, v2, mesh, 0, 0, next, rank);
        k = e;
    }
}
