// rt_bvh.h -- host BVH builder interface (rt_bvh.cpp).
#pragma once
#include <cstdint>
#include <vector>
#include "rt_types.h"

struct RtBvh {
    std::vector<RtNode> nodes;    // [0] padding, [1] root, sibling pairs adjacent at even indices
    std::vector<uint32_t> order;  // slot -> input triangle index (triangles are re-ordered into leaf order)
    uint32_t root_ref = 0xFFFFFFFFu;  // traversal reference word of the root (rt_kernels.hip: nodeRef)
    int depth = 0;                // maximum leaf depth = LDS stack entries a traversal can need
};

void rtBuildBvh(int64_t n, const float *v0, const float *v1, const float *v2, RtBvh &out);

// rank[t] = position of triangle t in the candidate order of the reference's per-mesh trees
void rtReferenceOrder(int64_t n, const float *v0, const float *v1, const float *v2, const int32_t *object_id, uint32_t *rank);
