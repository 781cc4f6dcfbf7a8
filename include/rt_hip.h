/* rt_hip.h -- C ABI of the MI355X-native trace loop (libraytracer_hip.so).
 *
 * This is the drop-in boundary for the per-pixel trace loop of cosmcif/raytracer (reference main.cpp:607-650
 * and everything it calls: trace_ray :221, closest :59, is_shadowed :41, PhongModel :81, toneMapping :240,
 * Sphere/Plane::intersect Objects.h:38,138, Triangle/BoundingBox/MeshLoader::intersect MeshLoader.h:70,151,392,
 * the texture functions Textures.h:12-188, Image::setPixel Image.h:91-95).
 *
 * The reference has no FFI of its own: its seam is source-level -- three mutable globals
 * `lights`, `ambient_light`, `objects` (main.cpp:36-38) read by trace_ray().  Each entry point below names the
 * reference construct it stands in for.  Plain C types only; every pointer is a HOST pointer unless the name
 * starts with d_ (device pointer, e.g. torch.Tensor.data_ptr()).  Input buffers are copied during the call; the
 * library owns all device memory until rt_destroy().  Every function returns 0 on success, non-zero on
 * failure; rt_last_error() then describes the failure (thread-local).  A context is bound to one GPU and is not
 * re-entrant.  There is no CPU fallback: without a usable HIP device rt_create() fails.
 */
#ifndef RT_HIP_H
#define RT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_context rt_context;

/* Material function pointers of the reference (Material.h:26,30,40,41) become IDs. */
enum rt_texture_id {
    RT_TEX_NONE = 0,
    RT_TEX_PERLIN_NORMAL = 1,      /* Textures.h:22  */
    RT_TEX_PERLIN_WATER = 2,       /* Textures.h:39  */
    RT_TEX_OPAL = 3,               /* Textures.h:55  */
    RT_TEX_PERLIN_TERRAIN = 4,     /* Textures.h:76  */
    RT_TEX_PERLIN_ICE_TERRAIN = 5, /* Textures.h:87  */
    RT_TEX_SNOW_TERRAIN = 6,       /* Textures.h:102 */
    RT_TEX_QWILFISH = 7,           /* Textures.h:123 */
    RT_TEX_BMP_COLOR = 8,          /* Textures.h:173 colorAt            (BMP slot 0) */
    RT_TEX_BMP_NORMAL = 9,         /* Textures.h:178 normalAt           (BMP slot 1) */
    RT_TEX_BMP_ROUGHNESS = 10,     /* Textures.h:186 roughnessAt        (BMP slot 2) */
    RT_TEX_BMP_OCCLUSION = 11,     /* Textures.h:182 ambientOcclusionAt (BMP slot 3) */
    RT_TEX_COUNT = 12
};

/* struct Material, reference Material.h:14-42 (same fields, same defaults expected from the caller). 96 bytes. */
typedef struct rt_material {
    float ambient[3];
    float diffuse[3];
    float specular[3];
    float reflection;
    float refraction;
    float sigma;
    float shininess;
    int32_t texture;         /* rt_texture_id, 0 = nullptr */
    int32_t has_normal_map;  /* bool hasNormalMap */
    int32_t normal_map;      /* rt_texture_id */
    int32_t is_anisotropic;
    float alpha_x;
    float alpha_y;
    int32_t has_img_texture;
    int32_t roughness;       /* rt_texture_id */
    int32_t occlusion;       /* rt_texture_id */
    int32_t _pad[2];
} rt_material;

/* Camera of the tile loop, reference main.cpp:561-563,570-578,629-641.  rot is column-major (glm layout) and is
 * always applied (identity for sampleScene); dz is +1 (sampleScene, main.cpp:634) or -1 (competitionScene, :637). */
typedef struct rt_camera {
    float origin[3];
    float dz;
    float rot[16];
    float fov_deg;
    float _pad[3];
} rt_camera;

enum rt_jitter_mode {
    RT_JITTER_AUTO = 0,  /* reference table when spp == 4, counter-based hash otherwise */
    RT_JITTER_TABLE = 1, /* reference jitterMatrix, main.cpp:601-606; requires spp == 4 */
    RT_JITTER_HASH = 2   /* rt_hash32-based jitter in [-0.75, 0.75) px, see DESIGN.md */
};

typedef struct rt_render_params {
    int32_t width;
    int32_t height;
    int32_t spp;            /* samples per pixel, reference: 4 (main.cpp:625) */
    int32_t max_depth;      /* bounce depth, reference: 3 (main.cpp:643) */
    uint32_t seed;          /* hash-jitter seed */
    int32_t jitter_mode;    /* rt_jitter_mode */
    int32_t tile_rank;      /* this context renders tiles t with t % tile_world == tile_rank (16x16 tiles, main.cpp:565) */
    int32_t tile_world;     /* 1 = whole image */
    int64_t chunk_samples;  /* primary samples per wavefront chunk; 0 = default */
    int32_t collect_stats;  /* bit0: count BVH node fetches / triangle tests and the longest walk in-kernel;
                               bit1: time every kernel class with HIP events on the launch stream (ms_gen ...);
                               bit2: wave-time profile of the kernels (prof_closest / prof_any / prof_shade; much slower) */
    int32_t flags;          /* rt_render_flags; 0 = a frame that others like it will follow */
} rt_render_params;

/* rt_render_params.flags */
#define RT_RENDER_SINGLE_FRAME 1 /* one image and no more (the reference's main()): the library then neither grows its
                                    queues inside the frame nor prepares them for a repeat of it -- allocating tens of
                                    GB costs seconds, which only pays when further frames follow */

typedef struct rt_stats {
    uint64_t primary_samples;
    uint64_t rays_closest;    /* primary + reflection + refraction rays traced (Ray ctor at main.cpp:156,183,642) */
    uint64_t rays_shadow;     /* shadow rays traced (Ray ctor at main.cpp:49; back-face culled ones are not rays; see shadow_zero) */
    uint64_t shadow_culled;   /* lights rejected by main.cpp:44-46 */
    uint64_t hits_shaded;     /* PhongModel invocations */
    uint64_t nodes_visited;   /* BVH nodes fetched by closest-hit rays  (collect_stats & 1 only) */
    uint64_t tris_tested;     /* triangle tests of closest-hit rays     (collect_stats & 1 only) */
    uint64_t queue_overflow;  /* must be 0 */
    double ms_total;          /* GPU time of the whole call (HIP events) */
    double ms_gen, ms_closest, ms_shade, ms_any, ms_resolve; /* per kernel class, summed over launches */
    uint64_t launches_closest, launches_any;
    uint64_t bvh_nodes, bvh_depth, n_triangles;
    uint64_t nodes_visited_any; /* BVH nodes fetched by shadow rays (collect_stats & 1 only) */
    uint64_t tris_tested_any;   /* triangle tests of shadow rays    (collect_stats & 1 only) */
    uint64_t bvh_max_stack;     /* traversal stack entries a ray can need in this BVH */
    /* wave-time breakdown of the traversal kernels, summed over waves; the counts ([2] [3] [5] [6] [8] [9]) come with
     * collect_stats & 1, the shader clocks ([0] [1] [4] [7]) with collect_stats & 4:
     * [0] refill (fetch + ray set-up + result writes), [1] node phases, [2] their count, [3] lanes active in them,
     * [4] triangle phases, [5] count, [6] lanes, [7] wave lifetime, [8] refills, [9] rays fetched */
    uint64_t prof_closest[10];
    uint64_t prof_any[10];
    /* k_shade: [0] loads + hit reconstruction, [1] material, light culling, textures, ambient, bounce set-up,
     * [2] queue reservation, [3] shadow-ray emission, [4] next-level emission, [5] wave iterations */
    uint64_t prof_shade[6];
    /* packet kernels: [0] packets, [1] node steps, [2] lanes interested in a child, summed over node steps,
     * [3] triangle steps, [4] live lanes summed over them, [5] packets over the step budget, [6] octant
     * groups walked, [7] rays handed to the per-lane kernels */
    uint64_t prof_packet_closest[8];
    uint64_t prof_packet_any[8];
    /* collect_stats & 1 only: the longest walk of a single ray, in steps (a launch cannot end before its longest ray) */
    uint64_t max_walk_closest, max_walk_any;
    double ms_packet;           /* packet kernel of the closest-hit queues (collect_stats & 2); ms_closest is the per-lane kernel alone */
    uint64_t shadow_unoccluded; /* shadow rays that reached their light, i.e. accumulator updates (collect_stats & 1 only) */
    uint64_t split_items;       /* work items that long shadow walks were split into (0 in most frames) */
    uint64_t shadow_zero;       /* shadow rays the reference casts (Ray ctor at main.cpp:49) that are NOT traced here because their
                                   pending radiance is exactly zero (e.g. main.cpp:172 with refraction 1): rays_shadow + shadow_zero
                                   is the reference's shadow-ray count */
    uint64_t chunks;            /* chunks of whole tiles the frame was rendered in (including repeated ones) */
    uint64_t chunk_retries;     /* chunks of a frame whose queues, sized from the fan-out seen so far, overflowed: the frame was
                                   rendered again with worst-case queues (results are those of the repeat; 0 in steady state) */
    double ms_trace;            /* the fused traversal launches (k_trace: shadow rays of an iteration + bounce rays and primary
                                   packets of the next; collect_stats & 2).  ms_any / ms_closest / ms_packet are then the split
                                   rounds / 0 / 0; RT_FUSE=0 launches the phases separately and fills those instead */
} rt_stats;

/* A closest-hit record as the reference's `Hit` (Core.h:34-57) plus primitive identity. 96 bytes. */
typedef struct rt_hit {
    int32_t hit;
    int32_t object;      /* index into the scene's object list, -1 on miss */
    int32_t triangle;    /* triangle index inside its mesh, -1 for sphere / plane / miss */
    uint32_t prim;       /* global primitive id (rt_render's out_primid encoding) */
    float distance;
    float intersection[3];
    float normal[3];
    float normal_shading[3];
    float uv[2];
    float tangent[3];
    float bitangent[3];
    float _pad[2];
} rt_hit;

#define RT_PRIM_MISS 0xFFFFFFFFu
#define RT_PRIM_ANALYTIC 0x80000000u /* | object index, for spheres and planes; triangles use their global index */

const char *rt_last_error(void);

/* Device bring-up. Fails (no fallback) when device_id is not a usable HIP device. */
int rt_create(int device_id, rt_context **out);
int rt_destroy(rt_context *ctx);

/* -- scene upload: stands in for filling the globals `objects`, `lights`, `ambient_light` (main.cpp:36-38) -- */

/* Material table. Objects refer to it through rt_set_object_materials(). */
int rt_set_materials(rt_context *ctx, int32_t n, const rt_material *materials);

/* object_material[o] = material index of scene object o (mesh, sphere or plane), in `objects` order.
 * Object order decides exact distance ties (first object wins, main.cpp:67). */
int rt_set_object_materials(rt_context *ctx, int32_t n_objects, const int32_t *object_material);

/* All mesh triangles of the scene, grouped by object in `objects` order and in file order inside a mesh, i.e.
 * what MeshLoader (MeshLoader.h:300-390) leaves in `triangles`, translation already baked in.
 * v0,v1,v2,nA,nB,nC: n*3 floats; tA,tB,tC: n*2 floats; flags bit0 = vertexNormals, bit1 = vertexTextures
 * (Triangle ctor, MeshLoader.h:62-68); object_id[n] non-decreasing. */
int rt_set_triangles(rt_context *ctx, int64_t n, const float *v0, const float *v1, const float *v2,
                     const float *nA, const float *nB, const float *nC, const float *tA, const float *tB,
                     const float *tC, const uint32_t *flags, const int32_t *object_id);

/* Spheres (Objects.h:12-104): unit sphere under a transformation; the three matrices are what
 * Object::setTransformation (Object.h:53-57) stores, column-major, n*16 floats each. */
int rt_set_spheres(rt_context *ctx, int32_t n, const float *M, const float *Minv, const float *Nmat,
                   const int32_t *object_id);

/* Cones (Objects.h:188-278): the unit cone (apex at the origin, opening towards +y, closed by the disc y = 1) under
 * a transformation; same three matrices as a sphere.  Spheres + cones: at most 64.  Optional (default: none). */
int rt_set_cones(rt_context *ctx, int32_t n, const float *M, const float *Minv, const float *Nmat,
                 const int32_t *object_id);

/* Planes (Objects.h:109-183): point[n*3], normal[n*3]. */
int rt_set_planes(rt_context *ctx, int32_t n, const float *point, const float *normal, const int32_t *object_id);

/* Point lights (main.cpp:26-34) and ambient_light (main.cpp:37). */
int rt_set_lights(rt_context *ctx, int32_t n, const float *position, const float *color, const float *ambient);

/* One of the four BMP maps of Textures.h:168-171, already decoded as bmpmini.hpp:65-88 leaves it:
 * top-down rows, BGR byte order, `channels` bytes per pixel.  slot: 0 color, 1 normal, 2 roughness, 3 occlusion. */
int rt_set_bmp(rt_context *ctx, int32_t slot, int32_t width, int32_t height, int32_t channels,
               const uint8_t *bytes);

/* Flatten check, BVH build (linear array of 64-byte 4-wide nodes), upload.  Must precede rendering; may be repeated. */
int rt_commit(rt_context *ctx);

/* Where rt_commit spent its time (the reference lumps scene loading into its one timer, main.cpp:541): host BVH build,
 * the reference's per-mesh candidate order (exact-tie rule), assembly of the 64-byte records, upload. */
typedef struct rt_commit_stats {
    uint64_t n_triangles, bvh_nodes;
    double ms_bvh_build, ms_reference_order, ms_records, ms_upload;
} rt_commit_stats;
int rt_get_commit_stats(rt_context *ctx, rt_commit_stats *out);

/* -- the render loop: stands in for main.cpp:607-650 (+ toneMapping :240, Image::setPixel Image.h:91-95) -- */

/* Whole-image render into host buffers (any may be NULL): out_rgb8[h*w*3] row-major as Image::data but bytes,
 * out_rgbf[h*w*3] = mean of the clamped samples before tone mapping (pixelColor/spp, main.cpp:646),
 * out_primid[h*w] = primitive hit by sample 0 of the pixel.  Requires tile_world == 1. */
int rt_render(rt_context *ctx, const rt_camera *cam, const rt_render_params *p, uint8_t *out_rgb8,
              float *out_rgbf, uint32_t *out_primid, rt_stats *stats);

/* Number of 16x16 tiles of a w*h image that belong to `rank` of `world`, and the total. */
int rt_tile_counts(int32_t width, int32_t height, int32_t rank, int32_t world, int64_t *n_local, int64_t *n_total);

/* Multi-GPU form: renders only this rank's interleaved tiles into DEVICE buffers in tile-major order
 * (local tile k = global tile k*world+rank; 256 pixels per tile, pixel (x,y) of the tile at y*16+x):
 * d_rgb8[n_local*256*3] u8, d_rgbf[n_local*256*3] f32, d_primid[n_local*256] u32 (any may be NULL).
 * Work is enqueued on `stream` (a hipStream_t; 0 = the context's own non-blocking stream, NOT the HIP default
 * stream) and is complete when the call returns (the call synchronises that stream). */
int rt_render_tiles_device(rt_context *ctx, const rt_camera *cam, const rt_render_params *p, void *d_rgb8,
                           void *d_rgbf, void *d_primid, void *stream, rt_stats *stats);

/* rt_render_tiles_device for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE (its loop allocates freely on the host).
 * Same inputs and outputs as rt_render_tiles_device, bit-identical results; the wavefront queues, counters and sample
 * accumulators live in a 64 MiB `static __device__` array of the library (one per GPU, held until the process ends)
 * instead of a workspace sized to the frame, so the frame is rendered in as many chunks of whole tiles as that array
 * dictates (slower than rt_render_tiles_device: short launches).  The call makes no hipMalloc / hipFree / event or
 * stream creation, does not synchronise and returns no statistics: work is enqueued on `stream` (0 = the context's
 * stream) and the outputs are complete when the stream reaches the end of it.  Limits: one 16x16 tile at the given
 * spp / max_depth / light count must fit the array (e.g. up to ~100 spp at depth 3 with 4 lights; depth <= 6 at
 * 4 spp), otherwise the call fails with a message.  Every call on a GPU shares that GPU's array: calls that overlap
 * on one GPU (two streams, two contexts, two threads) race -- serialise them.  Scene-sized buffers (BVH, traversal
 * spill area) were allocated by rt_commit. */
int rt_render_tiles_static(rt_context *ctx, const rt_camera *cam, const rt_render_params *p, void *d_rgb8,
                           void *d_rgbf, void *d_primid, void *stream);

/* Assembles the image on the gathering rank: d_tiles_* hold the ranks' tile-major buffers back to back in rank
 * order (exactly what an RCCL gather of rt_render_tiles_device outputs yields, ranks padded to `tiles_per_rank`
 * tiles each); outputs are row-major device images.  No counterpart in the reference (single process). */
int rt_assemble_tiles_device(rt_context *ctx, int32_t width, int32_t height, int32_t world,
                             int64_t tiles_per_rank, const void *d_tiles_rgb8, const void *d_tiles_rgbf,
                             const void *d_tiles_primid, void *d_rgb8, void *d_rgbf, void *d_primid, void *stream);

/* -- one frame over N GPUs of this process: stands where `#pragma omp parallel for` over the 16x16 tiles stood
 * (main.cpp:607) -- tiles interleaved over the GPUs (tile t -> rank t % N), scene + BVH replicated, one host thread
 * per GPU, no collective during rendering, ONE exchange at frame end: RCCL point-to-point over xGMI (ncclGroupStart;
 * ncclRecv x (N-1) on rank 0 / ncclSend on the peers; ncclGroupEnd) into a preallocated [rank][tile][256] buffer,
 * then the assembly kernel.  RCCL is dlopen'ed by rt_multi_create (single-GPU users never load it).
 * Usage: rt_multi_create; rt_set_* on rt_multi_context(m, 0) ONLY; rt_multi_commit (BVH built once, uploaded to every
 * GPU); rt_multi_render per frame.  The same device may be listed more than once (rehearsal of N ranks on one card):
 * those ranks exchange by device-to-device copies, since RCCL needs one GPU per rank. */
typedef struct rt_multi rt_multi;
#define RT_MULTI_REHEARSE_RCCL 1u /* with ONE device: push rank 0's tiles through ncclSend/ncclRecv to itself */
int rt_multi_create(const int32_t *device_ids, int32_t n, uint32_t flags, rt_multi **out);
int rt_multi_destroy(rt_multi *m);
int32_t rt_multi_size(const rt_multi *m);
const char *rt_multi_transport(const rt_multi *m); /* "rccl", "copy" or "none" (one rank, nothing to exchange) */
rt_context *rt_multi_context(rt_multi *m, int32_t rank);
int rt_multi_commit(rt_multi *m);
/* out_* as rt_render (host buffers, any may be NULL); stats: NULL or rt_multi_size(m) entries, one per rank.
 * p->tile_rank / tile_world are ignored (set per rank). */
int rt_multi_render(rt_multi *m, const rt_camera *cam, const rt_render_params *p, uint8_t *out_rgb8, float *out_rgbf,
                    uint32_t *out_primid, rt_stats *stats);
/* Where the last rt_multi_render spent its wall time (host clock, ms), so that a scaling loss can be assigned to the tile
 * partition (ms_share_max vs ms_share_min: GPU time of the slowest / fastest rank's share), to what a share costs over
 * its ideal 1/N (compare ms_share_max with a one-GPU frame), or to the exchange / assembly / read-back.  No counterpart in
 * the reference (its only timer wraps the whole program, main.cpp:541,651-653). */
typedef struct rt_multi_timings {
    double ms_render_wall; /* all ranks' shares, started together, until the last one has finished */
    double ms_share_max, ms_share_min;
    double ms_exchange;    /* the RCCL send/recv group (or the copies) until rank 0 holds every rank's tiles */
    double ms_assemble;    /* de-interleave on rank 0 */
    double ms_readback;    /* images to the host buffers */
} rt_multi_timings;
int rt_multi_last_timings(const rt_multi *m, rt_multi_timings *out);

/* -- trace_ray-style API (main.cpp:221-233) for batches of arbitrary rays; used by the parity tests -- */

/* rays[n*6] = origin xyz, direction xyz (direction is used as given, as trace_ray does).
 * out_color[n*3] = trace_ray(ray, depth) (clamped to [0,1]); out_hits[n] = closest(ray) (either may be NULL);
 * out_shadow[n*4] = is_shadowed() per light for the closest hit as PhongModel calls it (1 = shadowed). */
int rt_trace_batch(rt_context *ctx, int64_t n, const float *rays, int32_t depth, float *out_color,
                   rt_hit *out_hits, uint8_t *out_shadow, rt_stats *stats);

/* Known-answer hooks for the texture functions: out[n*3] = texture_id(uv[n*2]) (scalar maps replicate .r);
 * rt_eval_perlin: out[n] = glm::perlin(vec3 p[n*3]) (glm/gtc/noise.inl:65-132). */
int rt_eval_texture(rt_context *ctx, int32_t texture_id, int64_t n, const float *uv, float *out);
int rt_eval_perlin(rt_context *ctx, int64_t n, const float *p, float *out);

/* Read back the committed BVH (for the oracle-side counter checks): nodes[n_nodes*16] 32-bit words as stored
 * (64-byte 4-wide quantised nodes, layout in DESIGN.md), tri_order[*n_triangles] = global triangle index held by
 * BVH slot i.  *n_triangles is the number of SLOTS, >= the number of triangles: large triangles are cut into pieces
 * before the build and own one slot (one copy of their record) per piece.  Pass NULL to query sizes only. */
int rt_get_bvh(rt_context *ctx, int64_t *n_nodes, float *nodes, int64_t *n_triangles, uint32_t *tri_order);
/* ... and what a walk over it needs besides: the root's reference (node 0, or a leaf reference for a tree of <= 4 triangles),
 * the slab test's error-bound scale (max |coordinate| of any node box), the deepest traversal stack and how many of its
 * entries live in LDS.  tests/test_gpu_counters.py walks the tree on the CPU with them (oracle: ro_walk_counts) and
 * compares node steps / triangle tests with rt_trace_batch's rt_stats.nodes_visited / tris_tested (the
 * in-kernel counters that the roofline's flops rest on).  No counterpart in the reference (its tree is per mesh,
 * MeshLoader.h:211-285). */
int rt_get_bvh_info(rt_context *ctx, uint32_t *root_ref, float *bound, int32_t *max_stack, int32_t *lds_stack_entries);

/* Host-only self check of the BVH builder (no GPU, no context; used by the CPU tests): builds the tree of the given
 * triangles with `threads` host threads (0 = default), returns the node count, a hash of nodes + triangle order (equal
 * for every thread count) and the number of containment violations (a triangle outside a box above it, a quantised
 * box smaller than its float box, a triangle not reached exactly once) -- which must be 0.  No reference counterpart. */
int rt_bvh_check(int64_t n, const float *v0, const float *v1, const float *v2, int32_t threads, int64_t *n_nodes,
                 uint64_t *hash, int64_t *violations);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
