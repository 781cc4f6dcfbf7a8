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
    float reach;                 // max |coordinate| of any node box or unit-primitive box: how far the finite geometry reaches
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

// A ray's tag word (rayB.w): the sample's slot in the accumulator ring in the low bits, the ray's level (bounce count)
// above them, two flags on top.  How the 30 bits are split is decided per frame (RtQueues::sample_mask / level_shift):
// depth 3 needs 2 level bits, which leaves 2^28 ring samples.
#define RT_SAMPLE_MASK 0x3FFFFFFFu            // sample + level bits
#define RT_FLAG_HAS_REFR_SIBLING 0x40000000u  // reflection ray whose refraction sibling sits in the next slot
#define RT_SHADOW_HOLE 0xFFFFFFFFu            // w of a shadow record that holds no ray (unused slot of a k_shade slab)
#define RT_SHADOW_L0 0x40000000u              // w of a shadow record: the ray left a level-0 hit (RtQueues::occ0)
#ifndef RT_SLAB_MAX
#define RT_SLAB_MAX 2048u                    // slots a k_shade wave reserves at a time, per light / for the next level
#endif                                       // (4K frame, end of round 3: 512: +9.5 ms, 1024: 0, 2048: -2.2, 4096: +5 -- k_shade pays for the holes)
#ifndef RT_SHADE_BLOCKS_PER_CU
#define RT_SHADE_BLOCKS_PER_CU 4
#endif
#ifndef RT_SHADE_SEGMENT
#define RT_SHADE_SEGMENT 2048u                // consecutive queue entries a k_shade wave shades before it jumps
#endif                                        // (512: +9 ms, 1024: 0, 2048: -2.5; 2048 with slabs of 2048: -5.4, 4096 / 2048: the same; longer
                                              // runs of neighbouring hits in the queues are what the traversal launch gains from)
#ifndef RT_TRACE_MIN_BLOCKS
#define RT_TRACE_MIN_BLOCKS 7                // blocks per CU the traversal kernels are compiled and launched for: 7 = 72 VGPRs.  (8 = 64 VGPRs with
#endif                                       // 25 spilled VGPRs and 122 spilled SGPRs in the fused kernel: 4K frame 375.9 ms; 7: 374.3; 6 = 80 VGPRs: 387.3)
#define RT_TRACE_LDS_ROWS 2                  // LDS rows of 256 words behind a traversal block's stack rows: ages / batch order, sort bins
#define RT_TAIL_WAVES 8192                   // waves of a launch that RT_TAILPROF records
#define RT_BLOCK 256                         // threads per block of every kernel
#define RT_FLAG_INVALID 0x80000000u           // slot of a pixel outside the image or unused slab slot: traced as a miss, never shaded

// Wavefront queues (device pointers).  The frame is rendered as a PIPELINE of chunks (rt_abi.cpp: runFrame): iteration i
// traces and shades, in ONE pass per kernel, level 0 of chunk i, level 1 of chunk i-1, ... level D of chunk i-D.  The ray
// queues ping-pong by iteration (iteration i reads buffer i & 1, its k_shade appends the next levels' rays to the other
// one, behind the slots [0, prim_n) that the next chunk's primary rays occupy); every per-queue counter exists once per
// parity.  Sample accumulators form a ring of D + 1 chunk-sized slots.
// Rays: A = (origin.xyz, weight), B = (direction.xyz, tag: ring sample | level | flags), R = Fresnel reflectance to
// apply if the sibling hits.  Shadow rays: A = (shaded point, ring sample | RT_SHADOW_L0), L = light index,
// C = (pending rgb, 0); origin, direction and light distance are recomputed from the point (rtShadowRay).
struct RtQueues {
    float4 *rayA[2];
    float4 *rayB[2];
    float *rayR[2];
    float2 *hits;
    float4 *shA, *shC;
    long long *accum;          // [ring samples][3] fixed point
    uint32_t *counters;        // see RtCounter
    uint32_t *spill;           // traversal stack entries beyond the LDS part: [entry][grid thread]
    uint32_t *general;         // queue indices of the hits the fast k_shade leaves to the general one
    uint2 *split[2];           // work items (shadow ray, subtree) of any-hit walks split for being long; ping-pong by round
    uint32_t ray_cap, shadow_cap, split_cap, general_cap;
    // k_shade reserves the shadow queue a slab (2^sh_slab_shift slots) at a time and fills a slab with rays towards ONE light:
    // the light of record k is shSlabLight[k >> sh_slab_shift], written once per slab (a byte per record was one more
    // scattered store per shadow ray in k_shade and 2.8 GB each way per 4K frame)
    uint8_t *shSlabLight;
    uint32_t sh_slab_shift;
    // RT_TAILPROF (diagnosis): per launch and wave {start, left phase 0, left phase 1, end} in 100 MHz ticks, at
    // tailprof[(launch_id * RT_TAIL_WAVES + wave) * 4]; null = off
    unsigned long long *tailprof;
    uint32_t launch_id;
    // this iteration (set by the host per launch): entries [0, prim_n) of the current ray queue are the primary rays of the
    // chunk that enters the pipeline -- level 0, entry k belongs to ring sample prim_sample0 + k
    uint32_t prim_n, prim_sample0;
    // A frame's primary rays all start at the camera with weight 1: with prim_implied their (origin, weight) words are
    // neither written nor read -- primA stands for rayA[..][k], k < prim_n (16 of a primary ray's 36 bytes, three times:
    // k_gen_primary's store, the packets' load, k_shade's load).  0 for rays handed in by the host (rt_trace_batch).
    float4 primA;
    uint32_t prim_implied;
    uint32_t prim_far;         // the primary rays start far outside the scene (or anywhere: a host's batch): no box pre-test of the unit primitives
    uint32_t tile_stride;      // local tiles of a chunk: tile0 + k * tile_stride (tile0 = the chunk's number, stride = chunks of the frame)
    uint32_t l0_sample0;       // ring sample of the first level-0 shadow ray's chunk (the fused k_trace walks the shadow rays of
                               // iteration j next to the primary rays of chunk j + 1, which prim_n / prim_sample0 then describe)
    uint32_t sample_mask;      // tag & sample_mask = ring sample
    uint32_t level_shift, level_mask;  // (tag >> level_shift) & level_mask = level of the ray
    // Level-0 shadow results without atomics (null = off): a sample has ONE level-0 hit, hence at most one level-0 shadow
    // ray per light, so k_any can STORE the radiance of an unoccluded one at [light][sample - prim_sample0] (col0, flag
    // in occ0) and k_fold0 adds them to the accumulator afterwards, one lane per sample.  Same fixed-point sums, same bits;
    // pays where most shadow rays are level-0 rays that reach their light (the L2 retires only ~75 G atomics per second).
    uint8_t *occ0;             // [light][l0_stride] 1 = col0 holds a contribution; cleared per iteration
    float4 *col0;              // [light][l0_stride]
    uint32_t l0_stride;        // samples of a chunk
};

// Device counters.  Every hot counter owns a 128-byte line: returning atomics on one line serialise at the memory
// side (~10 ns each), so queue heads that are bumped from every CU must not share a line.  `p` is the parity of the
// iteration (0 / 1) the queue belongs to.
#define RT_CNT_STRIDE 32
#define RT_CNT_RAYS(p) ((p) * RT_CNT_STRIDE)                 // entries of ray queue p (primary slots + appended bounce rays + slab holes)
#define RT_CNT_SHADOW(p) ((2 + (p)) * RT_CNT_STRIDE)         // entries of the shadow queue of iteration parity p
#define RT_CNT_CULLED (4 * RT_CNT_STRIDE)
#define RT_CNT_SHADED (5 * RT_CNT_STRIDE)
#define RT_CNT_OVERFLOW (6 * RT_CNT_STRIDE)
#define RT_CNT_INVALID (7 * RT_CNT_STRIDE)                   // ray-queue slots that hold no ray: pixels outside the image (ragged edge tiles), unused slab slots
#define RT_CNT_NODES (8 * RT_CNT_STRIDE)                     // u64 nodes at +0, u64 triangle tests at +2, u32 longest walk at +4 (closest rays)
#define RT_CNT_ANY_NODES (9 * RT_CNT_STRIDE)                 // same for shadow rays
#define RT_CNT_PROF_CLOSEST (10 * RT_CNT_STRIDE)             // 10 x u64 wave-time breakdown of k_closest (stats build)
#define RT_CNT_PROF_ANY (11 * RT_CNT_STRIDE)                 // same for k_any
#define RT_CNT_PROF_SHADE (12 * RT_CNT_STRIDE)               // 6 x u64 wave-time breakdown of k_shade (stats build)
#define RT_CNT_PROF_PACKET (13 * RT_CNT_STRIDE)              // 8 x u64 packet statistics of k_packet (stats build)
#define RT_CNT_SHADOW_HOLES (14 * RT_CNT_STRIDE)             // unused slab slots of the shadow queues (not rays)
#define RT_CNT_ZERO (14 * RT_CNT_STRIDE + 8)                 // shadow rays not emitted because their pending radiance is exactly zero
#define RT_CNT_UNOCCLUDED (14 * RT_CNT_STRIDE + 4)           // u64: shadow rays that reached their light (stats build)
#define RT_CNT_SHADOW_L0 (14 * RT_CNT_STRIDE + 12)           // shadow rays emitted by level-0 hits (decides RtQueues::occ0)
#define RT_CNT_GENERAL(p) ((15 + (p)) * RT_CNT_STRIDE)       // hits of iteration parity p left to the general k_shade
// Frame totals (u64 each, one line): k_iter_ctl moves the per-iteration 32-bit counters here after every iteration, so
// that no 32-bit counter has to hold more than one iteration's worth
#define RT_CNT_TOTALS (23 * RT_CNT_STRIDE)
#define RT_TOT_RAYS 0          // ray-queue entries (incl. slots that hold no ray: RT_TOT_INVALID)
#define RT_TOT_SHADOW 1        // shadow-queue entries (incl. RT_TOT_SHADOW_HOLES)
#define RT_TOT_CULLED 2
#define RT_TOT_SHADED 3
#define RT_TOT_OVERFLOW 4
#define RT_TOT_INVALID 5
#define RT_TOT_SHADOW_HOLES 6
#define RT_TOT_ZERO 7
#define RT_TOT_SPLIT 8         // work items of split any-hit walks
#define RT_TOT_MAX_RAYS 9      // fullest ray queue / shadow queue of the frame, in entries
#define RT_TOT_MAX_SHADOW 10
#define RT_TOT_SHADOW_L0 11    // shadow rays emitted by level-0 hits
#define RT_TOT_WORDS 12
// cursors of the parts a queue is cut into for the dynamic fetch (one 128-byte line each)
#ifndef RT_WORK_PARTS
#define RT_WORK_PARTS 8
#endif
#define RT_CNT_WORKP_BASE (24 * RT_CNT_STRIDE)
#define RT_CNT_WORKP_CLOSEST(p, k) (RT_CNT_WORKP_BASE + ((p) * RT_WORK_PARTS + (k)) * RT_CNT_STRIDE)
#define RT_CNT_WORKP_ANY(p, k) (RT_CNT_WORKP_BASE + ((2 + (p)) * RT_WORK_PARTS + (k)) * RT_CNT_STRIDE)
#define RT_CNT_WORKP_PACKET(p, k) (RT_CNT_WORKP_BASE + ((4 + (p)) * RT_WORK_PARTS + (k)) * RT_CNT_STRIDE)  // primary rays of queue p (packets)
#define RT_CNT_WORKP_SHADE(p, k) (RT_CNT_WORKP_BASE + ((6 + (p)) * RT_WORK_PARTS + (k)) * RT_CNT_STRIDE)   // segments of ray queue p (fast k_shade)
#define RT_CNT_WORKP_SHADEG(p, k) (RT_CNT_WORKP_BASE + ((8 + (p)) * RT_WORK_PARTS + (k)) * RT_CNT_STRIDE)  // segments of the general list (general k_shade)
#define RT_CNT_WORKP_KINDS 10
// split rounds of the shadow kernels (rt_kernels.hip, "long any-hit walks"): per parity and round the number of work
// items queued for it (+0) and the next unclaimed one (+1)
#define RT_SPLIT_ROUNDS 6
#define RT_CNT_SPLIT_BASE (RT_CNT_WORKP_BASE + RT_CNT_WORKP_KINDS * RT_WORK_PARTS * RT_CNT_STRIDE)
#define RT_CNT_SPLIT(p, round) (RT_CNT_SPLIT_BASE + ((p) * (RT_SPLIT_ROUNDS + 1) + (round)) * 2)
#define RT_CNT_WORDS (RT_CNT_SPLIT_BASE + RT_CNT_STRIDE)  // (2 parities x 7 rounds x 2 words fit one stride)
#define RT_SPLIT_NULL 0xFFFFFFFFu     // ray index of an item that holds no work (reserved slots of a wave that could not export)
#define RT_JOIN_OCCLUDED 0x80000000u  // join word of a split shadow ray (shC.w as u32): outstanding items | this flag
