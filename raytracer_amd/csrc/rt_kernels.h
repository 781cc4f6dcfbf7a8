// rt_kernels.h -- host-callable launchers of the HIP kernels in rt_kernels.hip (internal to libraytracer_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include "rt_types.h"

struct RtLaunchCfg {
    hipStream_t stream = nullptr;
    int num_cus = 256;             // from hipDeviceProp_t at rt_create
    int stack_entries = 16;        // LDS traversal stack rows per lane (chosen at rt_commit); deeper entries spill to global memory
    bool packets = true;           // primary rays go through the packet kernel first (needs a tree whose worst-case stack fits 64
                                   // entries).  Only primary rays are coherent enough to pay (profiles/r01_packet_levels.md)
    int analytic_bytes = 0;        // LDS behind the stack holding the staged spheres + planes
    int split_budget = 1024;       // wave iterations (a power of two) after which a shadow ray's walk is split into work items (0 = never)
    int trace_blocks_per_cu = 4;   // resident 256-thread blocks per CU for the traversal kernels (LDS-limited)
    int shade_blocks = 0;          // k_shade grid in blocks; 0 = num_cus * RT_SHADE_BLOCKS_PER_CU (rt_render_tiles_static
                                   // uses a small grid: every wave owns (lights + 1) output slabs, i.e. queue slack)
    uint32_t shade_slab = 0;       // slots per slab; 0 = chosen from the queue's size (rtShadeSlab)
};

// One ITERATION of the pipeline works on ray queue `par` (0 / 1): entries [0, q.prim_n) are the primary rays of the chunk
// that enters, behind them the bounce rays (levels 1..D) the previous iteration's k_shade appended.
void rtLaunchIterCtl(const RtLaunchCfg &cfg, const RtQueues &q, int prev, uint32_t prim_next, bool set_cur, uint32_t prim_cur);
void rtLaunchGenPrimary(const RtLaunchCfg &cfg, const RtFrameDev &f, const RtQueues &q, int par, uint32_t tile0, uint32_t ntiles);
// stats: 0 plain, 1 counters, 2 + wave-time profile
// the fused traversal launch: shadow queue par_any, bounce rays [first, n) of ray queue par_closest and its primary rays
// [0, q.prim_n) as packets (a parity < 0 skips the phase); then the same phases as launches of their own
void rtLaunchTrace(const RtLaunchCfg &cfg, const RtSceneDev &sc, uint32_t root_ref, const RtQueues &q, int par_any, int par_closest,
                   uint32_t first, bool packets, int stats);
void rtLaunchPacket(const RtLaunchCfg &cfg, const RtSceneDev &sc, uint32_t root_ref, const RtQueues &q, int par, int stats);
void rtLaunchClosest(const RtLaunchCfg &cfg, const RtSceneDev &sc, uint32_t root_ref, const RtQueues &q, int par, uint32_t first, int stats);
void rtLaunchAny(const RtLaunchCfg &cfg, const RtSceneDev &sc, uint32_t root_ref, const RtQueues &q, int par, int stats);
void rtLaunchSplitRounds(const RtLaunchCfg &cfg, const RtSceneDev &sc, uint32_t root_ref, const RtQueues &q, int par, int stats);
void rtLaunchShade(const RtLaunchCfg &cfg, const RtSceneDev &sc, const RtQueues &q, int par, int max_depth, int spp,
                   uint32_t tile0, uint32_t *primid_tiles, bool prof, uint64_t max_inputs, int which = 0);  // which: 0 both instances, 1 fast, 2 general
// waves of a k_shade launch: each owns up to (lights + 1) partly filled output slabs at the end (queue slack, rt_abi.cpp)
inline int rtShadeBlocks(const RtLaunchCfg &cfg) { return cfg.shade_blocks > 0 ? cfg.shade_blocks : cfg.num_cus * RT_SHADE_BLOCKS_PER_CU; }
inline int rtShadeWaves(const RtLaunchCfg &cfg) { return rtShadeBlocks(cfg) * (RT_BLOCK / 64); }
// slots per slab for a queue of up to `max_inputs` entries: at most an eighth of what a wave can receive, so that the
// unused tails stay a small part of the queues
inline uint32_t rtShadeSlab(const RtLaunchCfg &cfg, uint64_t max_inputs) {
    if (cfg.shade_slab) return cfg.shade_slab;
    const uint64_t waves = (uint64_t)rtShadeWaves(cfg);
    uint32_t slab = 64;
    while (slab < RT_SLAB_MAX && (uint64_t)slab * 16 * waves <= max_inputs) slab <<= 1;
    return slab;
}
void rtLaunchFold0(const RtLaunchCfg &cfg, const RtQueues &q, int nlights);
void rtLaunchResolve(const RtLaunchCfg &cfg, const RtFrameDev &f, const RtQueues &q, uint32_t sample0, uint32_t tile0, uint32_t ntiles,
                     uint8_t *rgb8_tiles, float *rgbf_tiles);
void rtLaunchAssemble(const RtLaunchCfg &cfg, int width, int height, int world, long long tiles_per_rank, const void *t8,
                      const void *tf, const void *tp, void *o8, void *of, void *op);
void rtLaunchHitInfo(const RtLaunchCfg &cfg, const RtSceneDev &sc, uint32_t root_ref, const RtQueues &q, uint32_t n,
                     rt_hit *out_hits, uint8_t *out_shadow);
void rtLaunchRaysFromHost(const RtLaunchCfg &cfg, const float *d_rays, uint32_t n, const RtQueues &q);
void rtLaunchBatchColors(const RtLaunchCfg &cfg, const RtQueues &q, uint32_t n, float *d_out);
void rtLaunchEvalTexture(const RtLaunchCfg &cfg, const RtSceneDev &sc, int id, long long n, const float *d_uv, float *d_out);
void rtLaunchEvalPerlin(const RtLaunchCfg &cfg, long long n, const float *d_p, float *d_out);

// The library's static scratch (rt_kernels.hip): a file-scope __device__ array that rt_render_tiles_static carves its
// wavefront queues from, so that the call allocates nothing.  rtScratchResolve looks up the array's address on the
// CURRENT device (every GPU holds its own copy at its own address) -- called once per device by rt_create.
#define RT_STATIC_SCRATCH_BYTES (64ull << 20)
int rtScratchResolve(void **ptr, size_t *bytes, const char **err);
