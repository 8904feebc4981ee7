// dynk.hip — depth matrix -> per-sample dynamic 3x3 kernels, and the backward chain.
// Reference: SEAN.A_i_j, the expand/permute/matmul "broadcast" and mlp_gamma_s / mlp_beta_s
// (normalization.py:27-29,80-85).  conv3x3(sum_k m_k * st'_k) == sum_k m_k (*) (W . st'_k)  (SURVEY.md §8a 6c),
// so the 256-channel style map is never built: D[b,s,tap,k,c] = sum_l W_s[c,l,tap] * st'[b,k,l].
// Sizes are tiny (D is 46 KB per sample); these are plain one-thread-per-output kernels.
#include "dasr_common.h"

__global__ void k_dynk_stp(const float* __restrict__ st, const float* __restrict__ A_w, const float* __restrict__ A_b,
                           float* __restrict__ stp, int K, int L, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int l = (int)(i % L), k = (int)((i / L) % K);
        size_t b = i / ((size_t)L * K);
        float acc = A_b[k];
        for (int j = 0; j < K; ++j) acc = fmaf(A_w[k * K + j], st[(b * K + j) * L + l], acc);
        stp[i] = acc;
    }
}
// D layout [B][2][9][K][C]
__global__ void k_dynk_D(const float* __restrict__ stp, const float* __restrict__ Wg, const float* __restrict__ Wb,
                         float* __restrict__ D, int K, int L, int C, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(i % C), k = (int)((i / C) % K), tap = (int)((i / ((size_t)C * K)) % 9);
        int s = (int)((i / ((size_t)C * K * 9)) % 2);
        size_t b = i / ((size_t)C * K * 18);
        const float* Wp = (s ? Wb : Wg) + (size_t)c * L * 9 + tap;
        const float* sp = stp + (b * K + k) * L;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(Wp[(size_t)l * 9], sp[l], acc);
        D[i] = acc;
    }
}
extern "C" int dasr_dynk_fwd(const float* st, const float* A_w, const float* A_b, const float* Wg, const float* Wb,
                             float* stp, float* D, int B, int K, int L, int C, void* stream) {
    DASR_CHECK_PTR(st); DASR_CHECK_PTR(A_w); DASR_CHECK_PTR(A_b); DASR_CHECK_PTR(Wg); DASR_CHECK_PTR(Wb);
    DASR_CHECK_PTR(stp); DASR_CHECK_PTR(D);
    DASR_CHECK_SHAPE(B > 0 && K > 0 && L > 0 && C > 0);
    size_t n1 = (size_t)B * K * L, n2 = (size_t)B * 18 * K * C;
    DASR_LAUNCH(k_dynk_stp, dim3(dasr_ew_grid(n1)), dim3(256), 0, stream, st, A_w, A_b, stp, K, L, n1);
    DASR_LAUNCH(k_dynk_D, dim3(dasr_ew_grid(n2)), dim3(256), 0, stream, stp, Wg, Wb, D, K, L, C, n2);
    DASR_RETURN_LAUNCH_STATUS();
}

// dW_s[c,l,tap] = sum_{b,k} dD[b,s,tap,k,c] * stp[b,k,l]
__global__ void k_dynk_dW(const float* __restrict__ dD, const float* __restrict__ stp, float* __restrict__ dWg,
                          float* __restrict__ dWb, int B, int K, int L, int C, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int tap = (int)(i % 9), l = (int)((i / 9) % L), c = (int)((i / ((size_t)9 * L)) % C);
        int s = (int)(i / ((size_t)9 * L * C));
        float acc = 0.f;
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < K; ++k)
                acc = fmaf(dD[((((size_t)b * 2 + s) * 9 + tap) * K + k) * C + c], stp[((size_t)b * K + k) * L + l], acc);
        (s ? dWb : dWg)[((size_t)c * L + l) * 9 + tap] = acc;
    }
}
// dstp[b,k,l] = sum_{s,tap,c} dD[b,s,tap,k,c] * W_s[c,l,tap]; blockIdx.y = (s,tap) slice, partial sums are
// added with float atomics into the zeroed dstp (18 adds per element)
__global__ void k_dynk_dstp(const float* __restrict__ dD, const float* __restrict__ Wg, const float* __restrict__ Wb,
                            float* __restrict__ dstp, int K, int L, int C, size_t n) {
    const int st = blockIdx.y, s = st / 9, tap = st % 9;
    const float* Wp = s ? Wb : Wg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int l = (int)(i % L), k = (int)((i / L) % K);
        size_t b = i / ((size_t)L * K);
        const float* dp = dD + ((b * 18 + st) * K + k) * C;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc = fmaf(dp[c], Wp[((size_t)c * L + l) * 9 + tap], acc);
        atomicAdd(&dstp[i], acc);
    }
}
// dA_w[k,j] = sum_{b,l} dstp[b,k,l]*st[b,j,l]; dA_b[k] = sum_{b,l} dstp[b,k,l]; one workgroup per output
__global__ void __launch_bounds__(256) k_dynk_dA(const float* __restrict__ dstp, const float* __restrict__ st,
                                                 float* __restrict__ dA_w, float* __restrict__ dA_b, int B, int K,
                                                 int L) {
    __shared__ float red[4];
    const int e = blockIdx.x;
    const bool is_bias = e >= K * K;
    const int k = is_bias ? e - K * K : e / K, j = is_bias ? 0 : e % K;
    float acc = 0.f;
    for (int i = threadIdx.x; i < B * L; i += 256) {
        int b = i / L, l = i % L;
        float d = dstp[((size_t)b * K + k) * L + l];
        acc += is_bias ? d : d * st[((size_t)b * K + j) * L + l];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = red[0] + red[1] + red[2] + red[3];
        if (is_bias) dA_b[k] = r; else dA_w[e] = r;
    }
}
// dst[b,j,l] += sum_k A_w[k,j] * dstp[b,k,l]
__global__ void k_dynk_dst(const float* __restrict__ dstp, const float* __restrict__ A_w, float* __restrict__ dst,
                           int K, int L, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int l = (int)(i % L), j = (int)((i / L) % K);
        size_t b = i / ((size_t)L * K);
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(A_w[k * K + j], dstp[(b * K + k) * L + l], acc);
        dst[i] += acc;
    }
}
extern "C" int dasr_dynk_bwd(const float* dD, const float* st, const float* stp, const float* A_w, const float* Wg,
                             const float* Wb, float* dWg, float* dWb, float* dA_w, float* dA_b, float* dst,
                             float* dstp, int B, int K, int L, int C, void* stream) {
    DASR_CHECK_PTR(dD); DASR_CHECK_PTR(st); DASR_CHECK_PTR(stp); DASR_CHECK_PTR(A_w); DASR_CHECK_PTR(Wg);
    DASR_CHECK_PTR(Wb); DASR_CHECK_PTR(dWg); DASR_CHECK_PTR(dWb); DASR_CHECK_PTR(dA_w); DASR_CHECK_PTR(dA_b);
    DASR_CHECK_PTR(dst); DASR_CHECK_PTR(dstp);
    DASR_CHECK_SHAPE(B > 0 && K > 0 && L > 0 && C > 0);
    size_t nW = (size_t)2 * C * L * 9, nS = (size_t)B * K * L;
    DASR_LAUNCH(k_dynk_dW, dim3(dasr_ew_grid(nW)), dim3(256), 0, stream, dD, stp, dWg, dWb, B, K, L, C, nW);
    hipError_t e = hipMemsetAsync(dstp, 0, sizeof(float) * nS, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    DASR_LAUNCH(k_dynk_dstp, dim3(dasr_ew_grid(nS), 18), dim3(256), 0, stream, dD, Wg, Wb, dstp, K, L, C, nS);
    DASR_LAUNCH(k_dynk_dA, dim3(K * K + K), dim3(256), 0, stream, dstp, st, dA_w, dA_b, B, K, L);
    DASR_LAUNCH(k_dynk_dst, dim3(dasr_ew_grid(nS)), dim3(256), 0, stream, dstp, A_w, dst, K, L, nS);
    DASR_RETURN_LAUNCH_STATUS();
}
