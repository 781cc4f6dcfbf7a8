// rt_bvh.h -- host BVH builder interface (rt_bvh.cpp).
#pragma once
#include <cstdint>
#include <memory>
#include <new>
#include <utility>
#include <vector>
#include "rt_types.h"

// Vectors whose resize() leaves the new elements alone: the arrays of a 50 M-triangle build are gigabytes, written once by the
// threads that fill them -- a vector's own value-initialisation is one thread touching every page first (it was a quarter of
// the build).  Only for element types that are written before they are read.
template <class T>
struct RtRawAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = RtRawAlloc<U>; };
    template <class U> void construct(U *) noexcept {}
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <class T> using RtRawVec = std::vector<T, RtRawAlloc<T>>;

struct RtBvh {
    RtRawVec<RtNode4> nodes;      // 4-wide quantised nodes, root = node 0 (rt_types.h: RtNode4)
    RtRawVec<RtNodeP> nodes_p;    // the same nodes with float child boxes (packet kernels)
    std::vector<uint32_t> order;  // slot -> input triangle index, in leaf order; a large triangle that was cut into pieces
                                  // before the build (rt_bvh.cpp: presplit) owns several slots
    int64_t references = 0;       // = order.size(): triangles + the pieces pre-splitting added
    uint32_t root_ref = 0xFFFFFFFFu;  // reference word of the root: node 0, or a leaf when the scene has <= 4 triangles
    int depth = 0;                // maximum depth of the 4-wide tree
    int max_stack = 0;            // stack entries a near-first walk can need: max over paths of sum(children - 1)
    float bound = 0.0f;           // max |coordinate| of any node box
    int64_t binary_nodes = 0;     // size of the binary SAH tree the 4-wide tree was collapsed from
};

void rtBuildBvh(int64_t n, const float *v0, const float *v1, const float *v2, RtBvh &out);

// rank[t] = position of triangle t in the candidate order of the reference's per-mesh trees
void rtReferenceOrder(int64_t n, const float *v0, const float *v1, const float *v2, const int32_t *object_id, uint32_t *rank);

// host threads of the build (RT_BUILD_THREADS, else OMP_NUM_THREADS, else the hardware's, at most 16), and
// fn(begin, end) over [0, n) cut into one contiguous piece per thread (rt_commit's record assembly uses both)
int rtBuildThreads();
void rtParallelFor(size_t n, void (*fn)(size_t, size_t, void *), void *arg);
