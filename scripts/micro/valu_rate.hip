// Micro-benchmark: issue rate of scalar-f32 vs packed-f32 VALU on gfx950 (decides whether SLP-packed math pays).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256) void k_fma(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ __launch_bounds__(256) void k_pkfma(float *out, int iters, float a, float b) {
    f2 A = {a, a}, B = {b, b};
    f2 x0 = {(float)threadIdx.x, 1}, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            x0 = __builtin_elementwise_fma(x0, A, B); x1 = __builtin_elementwise_fma(x1, A, B); x2 = __builtin_elementwise_fma(x2, A, B); x3 = __builtin_elementwise_fma(x3, A, B);
            x4 = __builtin_elementwise_fma(x4, A, B); x5 = __builtin_elementwise_fma(x5, A, B); x6 = __builtin_elementwise_fma(x6, A, B); x7 = __builtin_elementwise_fma(x7, A, B);
        }
    }
    f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
int main() {
    float *d; hipMalloc(&d, 256 * 256 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int blocks_per_cu : {1, 2, 4, 8}) {
        for (int pk = 0; pk < 2; pk++) {
            const int grid = 256 * blocks_per_cu;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (pk) hipLaunchKernelGGL(k_pkfma, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k_fma<1>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double inst = (double)grid * 4 * iters * 64;  // wave-instructions
            const double fma = inst * 64 * (pk ? 2 : 1);
            printf("waves/SIMD %d %s: %.3f ms, %.1f TFLOP/s, wave-instr per SIMD per us %.1f (cycles/instr at 2.4GHz: %.2f)\n", blocks_per_cu,
                   pk ? "v_pk_fma_f32" : "v_fma_f32", ms, 2 * fma / ms / 1e9, inst / 1024 / (ms * 1e3), 2400.0 / (inst / 1024 / (ms * 1e3)));
        }
    }
    return 0;
}
