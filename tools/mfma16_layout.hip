// Layout probe (GPU box): v_mfma_f32_16x16x32_bf16 operand maps and ds_read_b64_tr_b16, against a host product.
// hipcc --offload-arch=gfx950 -O2 tools/mfma16_layout.hip -o /tmp/mfma16 && /tmp/mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: A and B both gathered by hand (A[row l&15][k=8(l>>4)+j], B[k=8(l>>4)+j][col l&15])
// mode 1: B through two transposed LDS reads from a [k][n] image with row stride ST elements
__global__ void k(const __bf16* A, const __bf16* B, float* C, int mode, int ST) {
    __shared__ __attribute__((aligned(16))) __bf16 sB[32 * 256];
    const int l = threadIdx.x;
    for (int i = l; i < 32 * 16; i += 64) sB[(i / 16) * ST + (i % 16)] = B[i];     // B is [k][n] row-major 32x16
    __syncthreads();
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j];
    if (mode == 0) {
        for (int j = 0; j < 8; ++j) b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)];
    } else {
        const int kg = l >> 4, tq = (l & 15) >> 2, tp = l & 3;
        const __bf16* p = sB + (8 * kg + tq) * ST + 4 * tp;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * ST));
        bf16x4 l4, h4;
        __builtin_memcpy(&l4, &lo, 8);
        __builtin_memcpy(&h4, &hi, 8);
        for (int e = 0; e < 4; ++e) { b[e] = l4[e]; b[4 + e] = h4[e]; }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}

int main() {
    __bf16 hA[16 * 32], hB[32 * 16];
    float ref[256];
    srand(1);
    for (int i = 0; i < 512; ++i) { hA[i] = (__bf16)(float)((rand() % 17) - 8); hB[i] = (__bf16)(float)((rand() % 13) - 6); }
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            float s = 0;
            for (int kk = 0; kk < 32; ++kk) s += (float)hA[m * 32 + kk] * (float)hB[kk * 16 + n];
            ref[m * 16 + n] = s;
        }
    __bf16 *dA, *dB;
    float* dC;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, 1024);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    int sts[3] = {16, 160, 96};
    for (int mode = 0; mode < 2; ++mode)
        for (int si = 0; si < (mode ? 3 : 1); ++si) {
            float hC[256];
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, mode, sts[si]);
            hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int i = 0; i < 256; ++i) bad += hC[i] != ref[i];
            printf("mode %d stride %d: %d of 256 wrong\n", mode, sts[si], bad);
        }
    return 0;
}
