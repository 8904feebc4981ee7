// Measured ceiling of the fp32 matrix pipe: every wave issues independent v_mfma_f32_32x32x2_f32 back to back
// (4 accumulators, no memory traffic).  Build + run: tools/bench_mfma_peak.py
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) k_mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    // blocks per CU: 8 -> 8 waves / SIMD (peak); 1 -> one wave per SIMD; 2 -> two (the conv kernels' occupancy)
    const int per_cu = argc > 1 ? atoi(argv[1]) : 8;
    const int blocks = 256 * per_cu, iters = 20000 * 8 / per_cu;
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 /*waves*/ * iters * 32.0 /*mfma per iter*/ * (2.0 * 32 * 32 * 2);
        printf("mfma_f32_32x32x2 loop, %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", per_cu, ms, flops / ms / 1e9);
    }
    return 0;
}
