// rt_types.h -- records shared by the host side of the C ABI (rt_abi.cpp, rt_bvh.cpp) and the HIP kernels.
// Layouts are fixed-size, 16-byte friendly and documented in DESIGN.md ("data layout in HBM").
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>  // float2 / float4
#include "../../include/rt_hip.h"

#define RT_TILE 16              // image tile edge, reference main.cpp:565
#define RT_TILE_PIXELS 256
#define RT_MAX_LIGHTS 16         // k_shade keeps a slab cursor per light in lanes 0..15 of a register (+ lane 16 for the ray slab)
#define RT_MAX_SPHERES 64
#define RT_MAX_PLANES 64
#ifndef RT_LDS_STACK
#define RT_LDS_STACK 16          // traversal stack rows per lane kept in LDS; deeper entries (rare) spill to global memory
#endif
#define RT_EPSILON 0.001f       // reference main.cpp:21
#define RT_ACC_SCALE 4294967296.0   // fixed-point accumulators: 2^32 steps per unit of radiance
#define RT_ACC_LIMIT 16777216.0f    // contributions are clamped to +-2^24 before conversion

// Binary build node (host only, rt_bvh.cpp): inner a = index of the left child (right = a+1), b = 0; leaf a = first
// triangle slot, b = triangle count (1..RT_LEAF_MAX).
struct RtNode {
    float mn[3];
    uint32_t a;
    float mx[3];
    uint32_t b;
};
#ifndef RT_LEAF_MAX
#define RT_LEAF_MAX 4
#endif

// 4-wide BVH node as the traversal kernels read it: 64 bytes = one cache line = four 16-byte loads per lane.
// Child boxes are quantised to 8 bits per plane inside the node's frame: plane = origin + q * scale (scale a power
// of two, so the product is exact); qlo is rounded down and qhi up, so a decoded box always contains the child's.
//   q0 = (origin.xyz, scale.x)   q1 = child[0..3]   q2 = (qlo_x, qlo_y, qlo_z, qhi_x)   q3 = (qhi_y, qhi_z, scale.y, scale.z)
// where every q?? word packs the four children's bytes (child c in byte c).
// child[c]: 0xFFFFFFFF = empty slot; inner node -> index of the child node; leaf -> RT_NODE_LEAF | (count-1) << 28 |
// record index of its first triangle.  Nodes and triangle records live in ONE array of 64-byte records (nodes first,
// then the triangles in leaf order), so a traversal step is the same four loads whether the lane is at a node or
// at a triangle.
struct RtNode4 {
    float origin[3];
    float scale_x;
    uint32_t child[4];
    uint32_t qlo[3];
    uint32_t qhi_x;
    uint32_t qhi_y, qhi_z;
    float scale_y, scale_z;
};
// The same 4-wide tree with full-precision child boxes, 128 bytes per node, same node indices and child references:
// what the packet kernels read with scalar loads (a whole wave walks one node at a time).
//   words 0-3 lo.x of children 0..3, 4-7 lo.y, 8-11 lo.z, 12-15 hi.x, 16-19 hi.y, 20-23 hi.z, 24-27 child refs, 28-31 0
struct RtNodeP {
    float lo[3][4];
    float hi[3][4];
    uint32_t child[4];
    uint32_t _pad[4];
};
#define RT_NODE_LEAF 0x80000000u
#define RT_NODE_DONE 0xFFFFFFFFu
#define RT_NODE_INDEX_MASK 0x0FFFFFFFu

// Triangle record, 64 bytes, BVH (slot) order:
// q0 = (A.xyz, n.x)  q1 = (B.xyz, n.y)  q2 = (C.xyz, n.z); n = normalize(cross(B-A, C-A)) (MeshLoader.h:58)
// q3 = (object, rank, prim, 0): what the exact-tie rule needs (see RtTriShade)
struct RtTriGeom {
    float q0[4], q1[4], q2[4];
    uint32_t object, rank, prim, general;  // general: 0 or RT_ID_GENERAL (ORed into the hit id)
};

// Per-slot shading attributes, 80 bytes: what Triangle::intersect needs beyond geometry (MeshLoader.h:99-121)
struct RtTriShade {
    float nA[3], nB[3], nC[3];
    float tA[2], tB[2], tC[2];
    uint32_t flags;   // bit0 vertexNormals, bit1 vertexTextures
    int32_t object;   // scene object index
    uint32_t prim;    // global triangle index (scene order) -- the primitive id reported to the caller
    uint32_t rank;    // position in the reference's candidate order: resolves exact distance ties like MeshLoader.h:399-404
    int32_t material; // = object_material[object] (saves k_shade a dependent load)
};

struct RtSphere {       // a transformed unit primitive: Sphere (Objects.h:12-104) or Cone (Objects.h:188-278); 240 bytes
    float M[16], Minv[16], N[16];
    int32_t object;
    int32_t material;
    float box_mn[3], box_mx[3];  // padded world-space bounds of the transformed primitive (rt_commit): a ray that
                                 // misses them cannot hit it, so the exact test is skipped
    int32_t kind;                // RT_KIND_SPHERE / RT_KIND_CONE
    int32_t _pad[3];
};
#define RT_KIND_SPHERE 0
#define RT_KIND_CONE 1

struct RtPlane {        // Objects.h:109-183
    float point[3];
    int32_t object;
    float normal[3];
    int32_t material;
};

struct RtLight {
    float pos[3];
    float _p0;
    float color[3];
    float _p1;
};

// Everything a kernel needs to see the scene; passed by value (kernarg -> SGPRs).
struct RtSceneDev {
    const RtNode4 *nodes;        // record array: n_nodes nodes, then n_tris triangle records
    const RtTriGeom *tri_geom;   // = nodes + n_nodes
    const RtNodeP *nodes_p;      // packet form of the nodes (float boxes)
    const RtTriShade *tri_shade;
    const RtSphere *spheres;
    const RtPlane *planes;
    const rt_material *materials;
    const int32_t *object_material;
    const RtLight *lights;
    const uint8_t *bmp[4];
    int32_t bmp_w[4], bmp_h[4], bmp_c[4];
    float ambient[3];
    float bound;                 // max |coordinate| of any node box (slab-test error bound, rt_kernels.hip)
    int32_t n_nodes, n_tris, n_spheres, n_planes, n_lights, n_objects, n_materials;
    int32_t any_ordered;         // shadow rays descend near-first like closest-hit rays (chosen at rt_commit)
    int32_t occluder_cache;      // k_any starts a ray on the triangle that last occluded a ray of its wave (rt_commit)
    uint64_t plane_general;      // bit k: plane k's material needs the general k_shade
    int32_t has_general;         // the scene holds primitives that need the general k_shade
    int32_t _pad0;
};

// Camera + image constants of the tile loop, precomputed on the host exactly as main.cpp:561-563 does.
struct RtFrameDev {
    float origin[3];
    float dz;
    float rot[16];
    float s, X, Y;      // pixel size and image-plane corner
    int32_t width, height, spp, tiles_x;
    int32_t tile_rank, tile_world;
    int32_t use_table;  // 1 = reference jitter table (spp == 4)
    uint32_t seed;
};

// hit ids: triangle slot, or analytic primitive
#define RT_ID_MISS 0xFFFFFFFFu
#define RT_ID_SPHERE 0x80000000u  // | sphere index
#define RT_ID_PLANE 0xC0000000u   // | plane index
#define RT_ID_KIND_MASK 0xC0000000u
// A hit whose shading needs the GENERAL k_shade (rt_kernels.hip): a sphere / cone (double-precision uv, Objects.h:80-81),
// or a triangle / plane whose material samples a BMP map or is anisotropic.  Spheres are recognised by their kind;
// triangles and planes carry this bit in their id (from the triangle record's 4th word / RtSceneDev::plane_general).
#define RT_ID_GENERAL 0x10000000u
#define RT_ID_TRI_MASK 0x0FFFFFFFu            // triangle slot of a triangle id
#define RT_GENERAL_HOLE 0xFFFFFFFFu  // unused slot of a wave's slab of the general list (RtQueues::general)

// ray flags packed above the sample index (30 bits of sample index: a chunk holds up to 2^30 - 1 primary samples)
#define RT_SAMPLE_MASK 0x3FFFFFFFu
#define RT_FLAG_HAS_REFR_SIBLING 0x40000000u  // reflection ray whose refraction sibling sits in the next slot
#define RT_SHADOW_HOLE 0xFFFFFFFFu            // w of a shadow record that holds no ray (unused slot of a k_shade slab)
#ifndef RT_SLAB_MAX
#define RT_SLAB_MAX 1024u                    // slots a k_shade wave reserves at a time, per light / for the next level
#endif
#ifndef RT_SHADE_BLOCKS_PER_CU
#define RT_SHADE_BLOCKS_PER_CU 4
#endif
#ifndef RT_SHADE_SEGMENT
#define RT_SHADE_SEGMENT 1024u                // consecutive queue entries a k_shade wave shades before it jumps
#endif
#define RT_BLOCK 256                         // threads per block of every kernel
#define RT_FLAG_INVALID 0x80000000u           // slot of a pixel outside the image or unused slab slot: traced as a miss, never shaded

// Wavefront queues of one chunk (device pointers).  Rays: A = (origin.xyz, weight), B = (direction.xyz, sample|flags),
// R = Fresnel reflectance to apply if the sibling hits.  Shadow rays: A = (shaded point, sample), L = light index,
// C = (pending rgb, 0); origin, direction and light distance are recomputed from the point (rtShadowRay).
struct RtQueues {
    float4 *rayA[2];
    float4 *rayB[2];
    float *rayR[2];
    float2 *hits;
    float4 *shA, *shC;
    uint8_t *shL;              // light index of a shadow record (one byte per slot)
    long long *accum;          // [chunk samples][3] fixed point
    uint32_t *counters;        // see RtCounter
    uint32_t *spill;           // traversal stack entries beyond the LDS part: [entry][grid thread]
    uint32_t *resid;           // rays the packet kernels hand to the per-lane kernels (queue indices)
    uint32_t *general;         // queue indices of the hits the fast k_shade leaves to the general one
    uint2 *split[2];           // work items (shadow ray, subtree) of any-hit walks split for being long; ping-pong by round
    uint32_t ray_cap, shadow_cap, split_cap;
    // Level-0 shadow results without atomics (null = off): a sample has ONE level-0 hit, hence at most one level-0 shadow
    // ray per light, so k_any can STORE the radiance of an unoccluded one at [light][sample] (col0, flag in occ0) and
    // k_fold0 adds them to the accumulator afterwards, one lane per sample.  Same fixed-point sums, same bits; pays where
    // most shadow rays are level-0 rays that reach their light (the L2 retires only ~75 G atomics per second).
    uint8_t *occ0;             // [light][l0_stride] 1 = col0 holds a contribution; cleared per chunk
    float4 *col0;              // [light][l0_stride]
    uint32_t l0_stride;        // samples of the chunk
};

// Device counters.  Every hot counter owns a 128-byte line: returning atomics on one line serialise at the memory
// side (~10 ns each), so queue heads that are bumped from every CU must not share a line.
#define RT_CNT_STRIDE 32
#define RT_CNT_RAYS(d) ((d) * RT_CNT_STRIDE)                 // rays queued for level d (d = 0..15)
#define RT_CNT_SHADOW(d) ((16 + (d)) * RT_CNT_STRIDE)        // shadow rays of level d
#define RT_CNT_CULLED (32 * RT_CNT_STRIDE)
#define RT_CNT_SHADED (33 * RT_CNT_STRIDE)
#define RT_CNT_OVERFLOW (34 * RT_CNT_STRIDE)
#define RT_CNT_INVALID (35 * RT_CNT_STRIDE)                  // level-0 slots outside the image (ragged edge tiles)
#define RT_CNT_NODES (36 * RT_CNT_STRIDE)                    // u64 nodes at +0, u64 triangle tests at +2, u32 longest walk at +4 (closest rays)
#define RT_CNT_ANY_NODES (37 * RT_CNT_STRIDE)                // same for shadow rays
#define RT_CNT_PROF_CLOSEST (38 * RT_CNT_STRIDE)             // 10 x u64 wave-time breakdown of k_closest (stats build)
#define RT_CNT_PROF_ANY (39 * RT_CNT_STRIDE)                 // same for k_any
#define RT_CNT_WORK_CLOSEST(d) ((40 + (d)) * RT_CNT_STRIDE)  // next unclaimed entry of level d's ray queue (dynamic fetch)
#define RT_CNT_WORK_ANY(d) ((56 + (d)) * RT_CNT_STRIDE)      // same for level d's shadow queue
#define RT_CNT_PROF_SHADE (72 * RT_CNT_STRIDE)               // 6 x u64 wave-time breakdown of k_shade (stats build)
#define RT_CNT_PROF_PACKET_CLOSEST (73 * RT_CNT_STRIDE)      // 8 x u64 packet statistics of k_packet<closest> (stats build)
#define RT_CNT_PROF_PACKET_ANY (74 * RT_CNT_STRIDE)          // same for shadow packets
#define RT_CNT_SHADOW_HOLES (75 * RT_CNT_STRIDE)             // unused slab slots of the shadow queues (not rays)
#define RT_CNT_GENERAL(d) (75 * RT_CNT_STRIDE + 12 + (d))     // hits of level d left to the general k_shade (d = 0..15)
#define RT_CNT_ZERO (75 * RT_CNT_STRIDE + 8)                 // shadow rays not emitted because their pending radiance is exactly zero
#define RT_CNT_UNOCCLUDED (75 * RT_CNT_STRIDE + 4)           // u64: shadow rays that reached their light (stats build)
#define RT_CNT_RESID_CLOSEST(d) ((76 + (d)) * RT_CNT_STRIDE) // rays of level d handed to the per-lane closest-hit kernel
#define RT_CNT_RESID_ANY(d) ((92 + (d)) * RT_CNT_STRIDE)     // same for shadow rays
#define RT_CNT_PWORK_CLOSEST(d) ((108 + (d)) * RT_CNT_STRIDE) // next unclaimed packet batch of level d (packet kernels)
#define RT_CNT_PWORK_ANY(d) ((124 + (d)) * RT_CNT_STRIDE)
// cursors of the parts a level's queue is cut into for the dynamic fetch (one 128-byte line each)
#ifndef RT_WORK_PARTS
#define RT_WORK_PARTS 8
#endif
#define RT_CNT_WORKP_CLOSEST(d, p) (RT_CNT_WORKP_BASE + ((d) * RT_WORK_PARTS + (p)) * RT_CNT_STRIDE)
#define RT_CNT_WORKP_ANY(d, p) (RT_CNT_WORKP_BASE + ((16 + (d)) * RT_WORK_PARTS + (p)) * RT_CNT_STRIDE)
// split rounds of the shadow kernels (rt_kernels.hip, "long any-hit walks"): per level and round the number of work
// items queued for it (+0) and the next unclaimed one (+1)
#define RT_SPLIT_ROUNDS 6
#define RT_CNT_SPLIT(level, round) (140 * RT_CNT_STRIDE + ((level) * (RT_SPLIT_ROUNDS + 1) + (round)) * 2)
#define RT_CNT_WORKP_BASE ((140 + 8) * RT_CNT_STRIDE)  // (16 levels x 7 rounds x 2 words of RT_CNT_SPLIT fit 7 strides)
#define RT_CNT_WORDS (RT_CNT_WORKP_BASE + 32 * RT_WORK_PARTS * RT_CNT_STRIDE)
#define RT_SPLIT_NULL 0xFFFFFFFFu     // ray index of an item that holds no work (reserved slots of a wave that could not export)
#define RT_JOIN_OCCLUDED 0x80000000u  // join word of a split shadow ray (shC.w as u32): outstanding items | this flag

