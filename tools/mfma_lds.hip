// Does feeding the matrix pipe from LDS cost throughput?  Same MFMA stream as tools/mfma_peak.hip (4 accumulators,
// v_mfma_f32_32x32x2_f32) but the operands come from ds_read_b128 at the conv kernels' 80-byte lane stride, double
// buffered exactly like k_conv3x3_mfma's fragment loop (4 reads feed 16 MFMAs).  One 512-thread workgroup per CU
// (2 waves / SIMD), no barriers, no global traffic in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: operands from registers only, 1: ds_read_b128 double-buffered, 3: same with random operand bits
__global__ void __launch_bounds__(512) k_loop(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 18 * 1024; i += blockDim.x) {
        if (MODE == 3) {            // activations / weights with full-entropy mantissas, as in the real convolutions
            unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            lds[i] = ((float)(h & 0xffffff) / 16777216.0f - 0.5f) * 2.0f;
        } else {
            lds[i] = 1e-3f * (float)(i & 255);
        }
    }
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float* pa = lds + ((2 * wv) * 34 + li) * 20 + 4 * lh;
    const float* pb = lds + 12288 + li * 20 + 4 * lh;
    float4 A0[2], B0[2], A1[2], B1[2];
    auto ld = [&](int j, float4 (&A)[2], float4 (&B)[2]) {
        const int o = (j % 9) * 20 + (j & 1) * 8;
        A[0] = *(const float4*)(pa + o);
        A[1] = *(const float4*)(pa + 34 * 20 + o);
        B[0] = *(const float4*)(pb + o);
        B[1] = *(const float4*)(pb + 32 * 20 + o);
    };
    auto mma = [&](const float4 (&A)[2], const float4 (&B)[2]) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                acc[2 * m + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].x, B[n].x, acc[2 * m + n], 0, 0, 0);
                acc[2 * m + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].y, B[n].y, acc[2 * m + n], 0, 0, 0);
                acc[2 * m + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].z, B[n].z, acc[2 * m + n], 0, 0, 0);
                acc[2 * m + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].w, B[n].w, acc[2 * m + n], 0, 0, 0);
            }
    };
    ld(0, A0, B0);
    if (MODE == 0) ld(1, A1, B1);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int j = 0; j < 18; j += 2) {
            if (MODE == 1 || MODE == 3) ld(j + 1, A1, B1);
            mma(A0, B0);
            if (MODE == 1 || MODE == 3) ld(j + 2, A0, B0);
            mma(A1, B1);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + tid] = s + A0[0].x + B1[1].w;
}

template <int MODE>
static void run(const char* name, float* out, int threads = 512, int blocks = 256, int lds_kb = 80, int iters = 600) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_loop<MODE>, dim3(blocks), dim3(threads), lds_kb * 1024, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * (threads / 64) * iters * 18 * 16 * (2.0 * 32 * 32 * 2);
        if (rep) printf("%-44s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
    }
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 2560 * 512);
    run<0>("operands resident in registers", out);
    run<1>("ds_read_b128 double-buffered (conv loop)", out);
    run<3>("same, random operand bits (power)", out, 512, 256, 80, 6000);
    run<1>("same, regular operand bits, long run", out, 512, 256, 80, 6000);
    // the 8-row conv tiles: two independent 256-thread workgroups per CU (73 KB of LDS each)
    run<1>("same, 2 x 256-thread workgroups per CU", out, 256, 512, 73);
    run<0>("registers only, 2 x 256-thread workgroups", out, 256, 512, 73);
    // workgroup churn: the 64->64 conv's grid - 1280 workgroups of 1152 MFMAs per wave (4 chunks), 2 per CU
    run<1>("1280 short workgroups (4 x 288 MFMAs), 2 / CU", out, 256, 1280, 73, 4);
    run<1>("2560 short workgroups (4 x 288 MFMAs), 2 / CU", out, 256, 2560, 73, 4);
    run<1>("640 x 512-thread workgroups (8 x 288), 1 / CU", out, 512, 1280, 95, 8);
    return 0;
}
