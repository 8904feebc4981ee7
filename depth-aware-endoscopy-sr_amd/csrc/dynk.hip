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
// The three contractions below are small GEMMs (M,N ~ 10^2..10^3, K = L, B*K or 18*C) over L2-resident operands.
// One wave = one 32x32 output tile on v_mfma_f32_32x32x2_f32, operands gathered straight from global memory
// (lane (i,h) supplies A[m0+i][2s+h] and B[2s+h][n0+i]); no LDS, no barriers.
//
// D[b,st,k,c] = sum_l stp[b,k,l] * W_s[c,l,tap]      M = (b,k), N = (s,c,tap), K = l
// The N index runs tap-fastest: W is OIHW ([c][l][3][3]), so the 32 lanes of a B-operand load read runs of nine
// consecutive floats (one (c, l) kernel) instead of 32 words 9 KB apart (one per channel: 32 cache lines per load).
__global__ void __launch_bounds__(256) k_dynk_D_mfma(const float* __restrict__ stp, const float* __restrict__ Wg,
                                                     const float* __restrict__ Wb, float* __restrict__ D, int B, int K,
                                                     int L, int C) {
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int M = B * K, N = 18 * C;
    const int tiles_n = (N + 31) / 32;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ((M + 31) / 32) * tiles_n) return;
    const int m0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
    const int m = m0 + li, n = n0 + li;
    const bool mv = m < M, nv = n < N;
    const int gb = nv ? n / (9 * C) : 0, rem = nv ? n % (9 * C) : 0;
    const int c = rem / 9, st = gb * 9 + rem % 9;
    const float* ap = stp + (size_t)(mv ? m : 0) * L;
    const float* bp = (st >= 9 ? Wb : Wg) + (size_t)c * L * 9 + (st % 9);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // sixteen K steps per trip: 32 operand loads in flight before the first MFMA (one load pair per MFMA was a chain of
    // L/2 = 128 L2 round trips: 77 us for 6 MFLOP)
    for (int l0 = lh; l0 < L; l0 += 32) {
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int l = l0 + 2 * u;
            av[u] = (mv && l < L) ? ap[l] : 0.f;
            bv[u] = (nv && l < L) ? bp[(size_t)l * 9] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
    if (!nv) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mm = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (mm >= M) continue;
        const int b = mm / K, k = mm % K;
        D[(((size_t)b * 18 + st) * K + k) * C + c] = acc[r];
    }
}
extern "C" int dasr_dynk_fwd(const float* st, const float* A_w, const float* A_b, const float* Wg, const float* Wb,
                             float* stp, float* D, int B, int K, int L, int C, void* stream) {
    DASR_CHECK_PTR(st); DASR_CHECK_PTR(A_w); DASR_CHECK_PTR(A_b); DASR_CHECK_PTR(Wg); DASR_CHECK_PTR(Wb);
    DASR_CHECK_PTR(stp); DASR_CHECK_PTR(D);
    DASR_CHECK_SHAPE(B > 0 && K > 0 && L > 0 && C > 0);
    size_t n1 = (size_t)B * K * L, n2 = (size_t)B * 18 * K * C;
    DASR_LAUNCH(k_dynk_stp, dim3(dasr_ew_grid(n1)), dim3(256), 0, stream, st, A_w, A_b, stp, K, L, n1);
    (void)n2;
    if ((L % 2) != 0) return DASR_E_UNSUPPORTED;
    {
        int tiles = ((B * K + 31) / 32) * ((18 * C + 31) / 32);
        DASR_LAUNCH(k_dynk_D_mfma, dim3((tiles + 3) / 4), dim3(256), 0, stream, stp, Wg, Wb, D, B, K, L, C);
    }
    DASR_RETURN_LAUNCH_STATUS();
}

// dW_s[c,l,tap] = sum_{(b,k)} dD[b,st,k,c] * stp[b,k,l]      M = (s,c,tap) tap-fastest, N = l, K = (b,k)
// (tap-fastest rows: the 16 accumulator rows of a lane are stored into two 36-byte kernels instead of 16 places 9 KB apart)
__global__ void __launch_bounds__(256) k_dynk_dW_mfma(const float* __restrict__ dD, const float* __restrict__ stp,
                                                      float* __restrict__ dWg, float* __restrict__ dWb, int B, int K,
                                                      int L, int C, float* __restrict__ zero, size_t nzero) {
    // clears the accumulator the NEXT kernel (k_dynk_dstp) adds into: spares a memset dispatch per SEAN
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nzero; i += (size_t)gridDim.x * 256) zero[i] = 0.f;
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int M = 18 * C, N = L, KK = B * K;
    const int tiles_n = (N + 31) / 32;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ((M + 31) / 32) * tiles_n) return;
    const int m0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
    const int m = m0 + li, n = n0 + li;
    const bool mv = m < M, nv = n < N;
    const int gb = mv ? m / (9 * C) : 0, rem = mv ? m % (9 * C) : 0;
    const int c = rem / 9, st = gb * 9 + rem % 9;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kk0 = lh; kk0 < KK; kk0 += 32) {          // sixteen K steps per trip, loads first (see k_dynk_D_mfma)
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int kk = kk0 + 2 * u;
            av[u] = bv[u] = 0.f;
            if (kk < KK) {
                const int b = kk / K, k = kk % K;
                if (mv) av[u] = dD[(((size_t)b * 18 + st) * K + k) * C + c];
                if (nv) bv[u] = stp[(size_t)kk * L + n];
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
    if (!nv) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mm = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (mm >= M) continue;
        const int gb2 = mm / (9 * C), rem2 = mm % (9 * C);
        (gb2 ? dWb : dWg)[((size_t)(rem2 / 9) * L + n) * 9 + rem2 % 9] = acc[r];
    }
}
// dstp[b,k,l] = sum_{s,tap,c} dD[b,s,tap,k,c] * W_s[c,l,tap]      M = (b,k), N = l, K = (s,tap,c) = 18*C
// One workgroup = one 32x32 output tile, six waves each contracting three (s,tap) slices (3*C of the 18*C terms) on
// v_mfma_f32_32x32x2_f32; the six partial tiles meet in LDS and are summed in a fixed order (no atomics, no memset:
// the one-thread-per-output version with an 18-way float-atomic fan-in took 58 us alone and 527 us inside the step).
#define DSTP_WAVES 6
__global__ void __launch_bounds__(64 * DSTP_WAVES) k_dynk_dstp_mfma(const float* __restrict__ dD, const float* __restrict__ Wg,
                                                                    const float* __restrict__ Wb, float* __restrict__ dstp,
                                                                    int B, int K, int L, int C) {
    __shared__ float part[DSTP_WAVES][32][33];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const int M = B * K, N = L;
    const int tiles_n = (N + 31) / 32;
    const int m0 = (blockIdx.x / tiles_n) * 32, n0 = (blockIdx.x % tiles_n) * 32;
    const int m = m0 + li, n = n0 + li;
    const bool mv = m < M, nv = n < N;
    const int b = mv ? m / K : 0, k = mv ? m % K : 0;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        // this wave's K range: its three consecutive (s, tap) slices x C channels, walked TAP-fastest (kk = 3 c + q): W is
        // OIHW, the three taps of a (c, l) kernel are 12 consecutive bytes and the six waves of the workgroup cover its 36 -
        // a channel-fastest walk touched a new cache line (9 KB further) on every step
        constexpr int QW = 18 / DSTP_WAVES;
        const int st0 = wv * QW;                                                     // QW divides 9: one s per wave
        const float* ap = dD + (((size_t)b * 18 + st0) * K + k) * C;                 // + q * K * C + c
        const float* bp = (st0 >= 9 ? Wb : Wg) + (size_t)(nv ? n : 0) * 9 + st0 % 9; // + c * L * 9 + q
        const int KKW = QW * C;
        for (int kk0 = lh; kk0 < KKW; kk0 += 32) { // sixteen K steps per trip, all 32 loads issued before the first MFMA
            float av[16], bv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int kk = kk0 + 2 * u, c = kk / QW, q = kk - c * QW;
                av[u] = (mv && kk < KKW) ? ap[(size_t)q * K * C + c] : 0.f;
                bv[u] = (nv && kk < KKW) ? bp[(size_t)c * L * 9 + q] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wv][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 32; e += 64 * DSTP_WAVES) {
        const int row = e >> 5, col = e & 31;
        float sum = part[0][row][col];
#pragma unroll
        for (int w = 1; w < DSTP_WAVES; ++w) sum += part[w][row][col];
        if (m0 + row < M && n0 + col < N) dstp[(size_t)(m0 + row) * L + n0 + col] = sum;
    }
}
// One launch for the two small tails: blocks [0, K*K+K): dA_w[k,j] = sum_{b,l} dstp[b,k,l]*st[b,j,l] and
// dA_b[k] = sum_{b,l} dstp[b,k,l] (one workgroup per output); the other blocks: dst[b,j,l] += sum_k A_w[k,j]*dstp[b,k,l]
__global__ void __launch_bounds__(256) k_dynk_dA_dst(const float* __restrict__ dstp, const float* __restrict__ st,
                                                     const float* __restrict__ A_w, float* __restrict__ dA_w,
                                                     float* __restrict__ dA_b, float* __restrict__ dst, int B, int K,
                                                     int L, size_t n) {
    __shared__ float red[4];
    const int nA = K * K + K;
    if ((int)blockIdx.x < nA) {
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
        return;
    }
    const size_t nblk = gridDim.x - nA;
    for (size_t i = (size_t)(blockIdx.x - nA) * 256 + threadIdx.x; i < n; i += nblk * 256) {
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
    (void)nW;
    {
        int tiles = ((18 * C + 31) / 32) * ((L + 31) / 32);
        DASR_LAUNCH(k_dynk_dW_mfma, dim3((tiles + 3) / 4), dim3(256), 0, stream, dD, stp, dWg, dWb, B, K, L, C,
                    (float*)nullptr, (size_t)0);
    }
    {
        int tiles = ((B * K + 31) / 32) * ((L + 31) / 32);
        DASR_LAUNCH(k_dynk_dstp_mfma, dim3(tiles), dim3(64 * DSTP_WAVES), 0, stream, dD, Wg, Wb, dstp, B, K, L, C);
    }
    DASR_LAUNCH(k_dynk_dA_dst, dim3(K * K + K + dasr_ew_grid(nS)), dim3(256), 0, stream, (const float*)dstp, st, A_w, dA_w,
                dA_b, dst, B, K, L, nS);
    DASR_RETURN_LAUNCH_STATUS();
}
