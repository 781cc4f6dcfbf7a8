// Micro-benchmark: SIMD cycles per wave64 instruction for the VALU opcodes the traversal step is made of (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_ops scripts/micro/valu_ops.hip && /tmp/valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define KERNEL(NAME, ASM, CLOB)                                                                          \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float a, float b) {              \
        float x[8];                                                                                     \
        for (int k = 0; k < 8; k++) x[k] = threadIdx.x + k;                                             \
        for (int i = 0; i < iters; i++) {                                                               \
            _Pragma("unroll") for (int u = 0; u < 8; u++) {                                             \
                _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile(ASM : "+v"(x[k]) : "v"(a), "v"(b) : CLOB); \
            }                                                                                           \
        }                                                                                               \
        float s = 0;                                                                                    \
        for (int k = 0; k < 8; k++) s += x[k];                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                        \
    }
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2", "memory")
KERNEL(k_mul, "v_mul_f32 %0, %0, %1", "memory")
KERNEL(k_add, "v_add_f32 %0, %0, %1", "memory")
KERNEL(k_addu, "v_add_u32 %0, %0, %1", "memory")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 1, %1", "memory")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", "memory")
KERNEL(k_cvtub, "v_cvt_f32_ubyte0 %0, %0", "memory")
KERNEL(k_min3, "v_min3_f32 %0, %0, %1, %2", "memory")
KERNEL(k_max, "v_max_f32 %0, %0, %1", "memory")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1", "vcc")
#define CLOB2 "s20", "s21"
KERNEL(k_cmpe64, "v_cmp_lt_f32 s[20:21], %0, %1", CLOB2)
KERNEL(k_mov, "v_mov_b32 %0, %1", "memory")
KERNEL(k_and, "v_and_b32 %0, %0, %1", "memory")
KERNEL(k_rcp, "v_rcp_f32 %0, %0", "memory")

typedef void (*kern)(float *, int, float, float);
int main() {
    float *d; hipMalloc(&d, 256 * 256 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    struct { const char *name; kern k; } ks[] = {{"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_add_u32", k_addu},
        {"v_lshl_add_u32", k_lshladd}, {"v_cndmask_b32", k_cndmask}, {"v_cvt_f32_ubyte0", k_cvtub}, {"v_min3_f32", k_min3},
        {"v_max_f32", k_max}, {"v_cmp_lt_f32 vcc", k_cmp}, {"v_cmp_lt_f32 sgpr", k_cmpe64}, {"v_mov_b32", k_mov}, {"v_and_b32", k_and}, {"v_rcp_f32", k_rcp}};
    for (auto &e : ks) {
        const int grid = 256 * 8;  // 8 waves per SIMD
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double inst = (double)grid * 4 * iters * 64;  // wave-instructions
        printf("%-20s %.3f ms  wave-instr per SIMD per us %.1f  (cycles/instr at 2.4 GHz: %.2f)\n", e.name, ms, inst / 1024 / (ms * 1e3),
               2400.0 / (inst / 1024 / (ms * 1e3)));
    }
    return 0;
}
