// conv_mfma.hip — 3x3 / stride 1 / pad 1 convolutions of the trunk as im2col-free implicit GEMMs on the
// fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD), NHWC x HWIO.
//
//   forward : out[p, co]  = sum_{tap, ci} in[p + off(tap), ci] * W[tap][ci][co]       M = pixels, N = co, K = 9*Cin
//   dgrad   : the same kernel run on dconv with the taps flipped and ci/co swapped      (WMODE = 1)
//   wgrad   : dW[tap][ci][co] = sum_p in[p + off(tap), ci] * dconv[p, co]               M = ci,  N = co, K = pixels
//
// Forward / dgrad kernel.  Workgroup = 256 threads = 4 waves, output tile = 8 rows x 32 columns of pixels x
// (32*NT) output channels.  Wave w owns tile rows 2w and 2w+1 (two 32-pixel M-tiles) x NT 32-channel N-tiles:
// 2*NT accumulators of 16 VGPRs.  K loop: input channels in chunks of 16; per chunk the 10 x 34 input halo
// tile (zero outside the image = the convolution's zero padding) and the 9 x (32*NT) x 16 weight slice are
// staged in LDS with a pixel / channel stride of 20 floats, which makes every ds_read_b128 of a wave
// conflict-free (MI355X_MICROARCH.md §LDS: 16 lanes x 16 B at stride 80 B hit 16 distinct 4-bank slots).
// One ds_read_b128 feeds four K=2 MFMA steps: lane (i, h) holds channels 8q+4h .. 8q+4h+3 of pixel i (A) or
// of output channel i (B), so MFMA step j contracts channel 8q+j on the lower half-wave and 8q+4+j on the
// upper one — any K order is valid as long as A and B agree.
// Epilogue: D fragment has the output channel on the lane (col = lane&31) and 16 pixels in registers, so each
// store instruction writes two 128-byte runs; bias, residual, activation and the PixelShuffle index map are
// applied there.
#include "dasr_common.h"
#include "bf16.h"
#include "conv_kernels.h"

#define CM_TW 32
#define CM_CK 16
#define CM_CKP 20
#define CM_HALO_W (CM_TW + 2)

// Phase timing of the forward kernel (tools/conv_phase_timing.py builds a private copy with -DDASR_CONV_TIMING;
// never defined in the shipped library).
#ifdef DASR_CONV_TIMING
__device__ unsigned long long g_cm_phase[8];
#define CM_T_DECL unsigned long long cm_tprev = __builtin_readcyclecounter(), cm_tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define CM_T_MARK(i) do { unsigned long long cm_now = __builtin_readcyclecounter(); cm_tacc[i] += cm_now - cm_tprev; cm_tprev = cm_now; } while (0)
#define CM_T_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int cm_i = 0; cm_i < 8; ++cm_i) atomicAdd(&g_cm_phase[cm_i], cm_tacc[cm_i]); } while (0)
extern "C" int dasr_debug_conv_phase_read(unsigned long long* host8, int reset) {
    hipError_t e = hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_cm_phase), sizeof(unsigned long long) * 8);
    if (e == hipSuccess && reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_cm_phase), z, sizeof(z));
    }
    return (int)e;
}
#else
#define CM_T_DECL
#define CM_T_MARK(i)
#define CM_T_FLUSH()
#endif

struct ConvMfmaArgs {
    const float* x;         // [B,H,W,Cin]
    const float* w;         // packed kernel of the FORWARD conv: HWIO then per-tap transpose (dasr_weight_pack_fwd)
    const float* bias;      // [Cout] or null
    const float* residual;  // [B,H,W,Cout] or null
    float* y;
    int B, H, W, Cin, Cout;
    int act, ps_r, accumulate;
    // dgrad only: multiply by act'(mask_src[same index]) (mask_src = the forward conv's INPUT, which is the
    // activated output of the producing layer) and store through the INVERSE PixelShuffle(unps_r) map, i.e.
    // write the gradient w.r.t. the producing conv's raw output directly (no separate epilogue-backward pass)
    const float* mask_src;
    int mask_act, unps_r;
    // forward only, optional: per-tile instance-norm partials of the (pre-activation) output, [B][tiles][Cout][2] =
    // (mean, M2) of the tile's pixels per channel; every tile must take the fast epilogue (W % 32 == 0, act / residual /
    // PixelShuffle off) - conv_mfma_fwd_stats checks that
    float* stats_part;
};

// CM_TH = tile rows = 2 x waves per workgroup (8 rows / 256 threads, or 16 rows / 512 threads: the 16-row tile
// stages each weight slice once for twice the pixels - weights are ~60 % of the staged bytes - and still keeps
// two waves per SIMD with a single workgroup per CU).
template <int NT, int WMODE, int CM_TH, bool UNMASK>
__global__ void __launch_bounds__(32 * CM_TH) k_conv3x3_mfma(ConvMfmaArgs a) {
    DASR_DYN_SMEM(smem);
    constexpr int CM_HALO_H = CM_TH + 2;
    constexpr int NTHR = 32 * CM_TH;
    float* sIn = (float*)smem;                                // [HALO_H*HALO_W][CKP]
    float* sW = sIn + CM_HALO_H * CM_HALO_W * CM_CKP;         // [9][32*NT][CKP]
    constexpr int NTILE = 32 * NT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware block order (1-D grid): consecutive workgroup ids are dealt round-robin to the 8 XCDs, so the
    // N-slices of one pixel tile get ids 8 apart - same XCD, back to back - and the second one finds the input
    // tile in that XCD's L2 instead of fetching it from HBM again.
    const int tiles_x = (a.W + CM_TW - 1) / CM_TW, tiles_y = (a.H + CM_TH - 1) / CM_TH;
    const int nsl = a.Cout / NTILE, G = tiles_x * tiles_y * a.B;
    const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
    const int gt = (kq / nsl) * 8 + xcd;
    if (gt >= G) return;
    const int tile = gt % (tiles_x * tiles_y);
    const int x0 = (tile % tiles_x) * CM_TW, y0 = (tile / tiles_x) * CM_TH;
    const int b = gt / (tiles_x * tiles_y), n0 = (kq % nsl) * NTILE;

    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // Software pipeline over the 16-channel chunks: the global loads of chunk c+1 are issued into registers before
    // the MFMA work of chunk c starts and are written to LDS after it (split "issue early / write late" staging),
    // so their latency hides under this workgroup's own matrix work instead of relying on the co-resident one.
    constexpr int NIN = (CM_HALO_H * CM_HALO_W * 4 + NTHR - 1) / NTHR;
    constexpr int NWT = (9 * NTILE * 4 + NTHR - 1) / NTHR;
    float4 pin[NIN], pwt[NWT];
    const float* wsrc = WMODE == 0 ? a.w + (size_t)9 * a.Cin * a.Cout : a.w;
    // Staging addresses are computed ONCE per workgroup, as in the bf16 kernel (conv_bf16_mfma.hip): a byte offset per
    // 16-byte piece from the sample's / the kernel slice's base, DASR_OOB for halo pixels outside the image (the buffer
    // load's range check returns the zero padding), so that the per-chunk prefetch is "add the chunk offset, load".  The
    // pixel index of a thread's first piece is decoded with one division and advanced by a compile-time (rows, columns)
    // step.  (Per chunk and piece it was: two divisions, a 64-bit address, four compares and an exec-masked load - ~450
    // VALU instructions per thread and chunk next to 72 MFMAs per wave.)
    constexpr int NINP = CM_HALO_H * CM_HALO_W * 4, NWTP = 9 * NTILE * 4;
    const BufRsrc rx = dasr_make_rsrc(a.x + (size_t)b * a.H * a.W * a.Cin, (size_t)a.H * a.W * a.Cin * sizeof(float));
    const BufRsrc rw = dasr_make_rsrc(wsrc, (size_t)9 * a.Cin * a.Cout * sizeof(float));
    unsigned offx[NIN], offw[NWT];
    {
        constexpr int DPIX = NTHR / 4, DROW = DPIX / CM_HALO_W, DCOL = DPIX % CM_HALO_W;
        const int pixb = a.Cin * (int)sizeof(float), rowb = a.W * pixb;
        int prow = (tid >> 2) / CM_HALO_W, pcol = (tid >> 2) % CM_HALO_W;
        int off = (y0 - 1 + prow) * rowb + (x0 - 1 + pcol) * pixb + 16 * (tid & 3);
#pragma unroll
        for (int u = 0; u < NIN; ++u) {
            const bool ok = ((u + 1) * NTHR <= NINP || tid + NTHR * u < NINP) && (unsigned)(y0 - 1 + prow) < (unsigned)a.H &&
                            (unsigned)(x0 - 1 + pcol) < (unsigned)a.W;
            offx[u] = ok ? (unsigned)off : DASR_OOB;
            prow += DROW;
            pcol += DCOL;
            off += DROW * rowb + DCOL * pixb;
            const bool wrap = pcol >= CM_HALO_W;
            prow += wrap ? 1 : 0;
            pcol -= wrap ? CM_HALO_W : 0;
            off += wrap ? rowb - CM_HALO_W * pixb : 0;
        }
        // weight slice as [tap][n][k]: both modes read a [tap][n][k]-ordered source with k contiguous
        //   forward: second half of the packed kernel, [tap][co][ci]
        //   dgrad  : first half (HWIO of the forward conv = [tap][n = ci_f][k = co_f]) with the taps flipped
        const int tapb = a.Cout * a.Cin * (int)sizeof(float);
#pragma unroll
        for (int u = 0; u < NWT; ++u) {
            const int idx = tid + NTHR * u;
            const int q4 = idx & 3, nl = (idx >> 2) % NTILE, tap = idx / (NTILE * 4);
            const int tsrc = WMODE == 0 ? tap : 8 - tap;
            offw[u] = ((u + 1) * NTHR <= NWTP || idx < NWTP)
                          ? (unsigned)(DASR_MUL24(tsrc, tapb) + ((n0 + nl) * a.Cin + 4 * q4) * (int)sizeof(float))
                          : DASR_OOB;
        }
    }
    auto ld16 = [&](BufRsrc r, unsigned off) {
        const u32x4_t v = dasr_buffer_load16(r, off);
        float4 f;
        __builtin_memcpy(&f, &v, 16);
        return f;
    };
    auto prefetch = [&](int c0) {
        const unsigned cb = (unsigned)c0 * (unsigned)sizeof(float);
#pragma unroll
        for (int u = 0; u < NIN; ++u) pin[u] = ld16(rx, offx[u] + cb);
#pragma unroll
        for (int u = 0; u < NWT; ++u) pwt[u] = ld16(rw, offw[u] + cb);
    };
    // LDS images: piece idx goes to (idx / 4) * CM_CKP + 4 * (idx % 4) of its tile - a per-thread base plus a compile-time
    // step per u (the weight tile's [tap][n] rows are consecutive, so it is one flat array of 16-byte pieces as well)
    float* const lin = sIn + (tid >> 2) * CM_CKP + 4 * (tid & 3);
    float* const lwt = sW + (tid >> 2) * CM_CKP + 4 * (tid & 3);
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < NIN; ++u)
            if ((u + 1) * NTHR <= NINP || tid + NTHR * u < NINP) *(float4*)(lin + u * (NTHR / 4) * CM_CKP) = pin[u];
#pragma unroll
        for (int u = 0; u < NWT; ++u)
            if ((u + 1) * NTHR <= NWTP || tid + NTHR * u < NWTP) *(float4*)(lwt + u * (NTHR / 4) * CM_CKP) = pwt[u];
    };
    CM_T_DECL;
    prefetch(0);
    CM_T_MARK(0);
    for (int c0 = 0; c0 < a.Cin; c0 += CM_CK) {
#ifdef DASR_CONV_NOSTAGE       // experiment only (tools/conv_phase_timing.py): the MFMA loop on whatever is in LDS
        if (c0 == 0)
#endif
        {
        __syncthreads();                       // every wave is done reading the previous chunk
        CM_T_MARK(1);
        commit();
        CM_T_MARK(2);
        __syncthreads();
        CM_T_MARK(3);
        if (c0 + CM_CK < a.Cin) prefetch(c0 + CM_CK);
        CM_T_MARK(4);
        }
        // ---- 9 taps x 16 channels = 18 fragment steps of 4 K=2 MFMAs per accumulator.  The LDS reads of step j+1
        // are issued before the MFMAs of step j (two statically named fragment sets), so the matrix pipe never
        // waits for a ds_read at the start of a step.
        auto ldfrag = [&](int j, float4 (&A)[2], float4 (&Bf)[NT]) {
            const int tap = j >> 1, q = j & 1;
            const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
            for (int m = 0; m < 2; ++m)
                A[m] = *(const float4*)(sIn + ((2 * wv + m + dy) * CM_HALO_W + li + dx) * CM_CKP + 8 * q + 4 * lh);
#pragma unroll
            for (int n = 0; n < NT; ++n)
                Bf[n] = *(const float4*)(sW + (tap * NTILE + 32 * n + li) * CM_CKP + 8 * q + 4 * lh);
        };
        auto mma = [&](const float4 (&A)[2], const float4 (&Bf)[NT]) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].x, Bf[n].x, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].y, Bf[n].y, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].z, Bf[n].z, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].w, Bf[n].w, acc[m][n], 0, 0, 0);
                }
        };
        float4 A0[2], B0[NT], A1[2], B1[NT];
        ldfrag(0, A0, B0);
        // 16-row tiles: fully unrolled, every LDS address becomes "base register + immediate" (tap offsets are
        // compile-time), no integer multiplies between the MFMAs (128->128: 775 -> 735 us).  8-row tiles and NT = 1 lose when
        // fully unrolled (242 -> 254 us) but gain from unrolling one kernel row (3 taps: dx immediate, dy by increment).
#pragma unroll(CM_TH == 16 && NT == 2 ? 9 : 3)
        for (int j = 0; j < 18; j += 2) {
            ldfrag(j + 1, A1, B1);
            mma(A0, B0);
            if (j + 2 < 18) ldfrag(j + 2, A0, B0);
            mma(A1, B1);
        }
        CM_T_MARK(5);
    }
    // ---- epilogue
#ifdef DASR_CM_NOEPI     // timing experiment only: what the epilogue costs (garbage results)
    {
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[m][n][r];
        if (s == 12345.678f) a.y[(size_t)blockIdx.x * 16 + wv] = s;
        return;
    }
#endif
    const int rr = a.ps_r * a.ps_r;
    // Fast path (full-width tiles, the usual case): every address is "scalar row base + scalar pixel offset + one
    // per-lane 32-bit byte offset", the optional residual / accumulate operands are fetched for all 16 pixels of a
    // fragment before the first store, and the activation is branch-free.  The generic path below spends ~40 VALU
    // instructions and several scalar branches per ELEMENT on 64-bit index arithmetic (measured: 15 % of the
    // 128->128 kernel, 32 % of the 64->64 one).
    if (!UNMASK && x0 + CM_TW <= a.W && !(a.ps_r > 1 && a.residual != nullptr)) {
        const int wvu = DASR_UNIFORM(wv);
        const int ps = a.ps_r, Cq = a.Cout / rr;
        const unsigned pstride = (unsigned)(ps > 1 ? ps * Cq : a.Cout) * 4u;      // bytes between x and x+1 for one lane
        const bool is_relu = a.act == DASR_ACT_RELU;
        const float slope = a.act == DASR_ACT_LRELU02 ? 0.2f : 1.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int gy = y0 + 2 * wvu + m;
            if (gy >= a.H) continue;
            // scalar element index of (b, gy, x0) in the output (through the PixelShuffle map when ps > 1)
            const size_t obase = ps > 1 ? (((size_t)b * a.H * ps + (size_t)gy * ps) * ((size_t)a.W * ps) + (size_t)x0 * ps) * Cq
                                        : (((size_t)b * a.H + gy) * a.W + x0) * a.Cout;
            char* yrow = (char*)(a.y + obase);
            const char* rrow = a.residual ? (const char*)(a.residual + obase) : nullptr;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int co = n0 + 32 * n + li;
                const float bv = a.bias ? a.bias[co] : 0.f;
                unsigned loff;                                                    // this lane's byte offset in the row
                if (ps > 1) {
                    const int c = co / rr, i = (co / ps) % ps, j = co % ps;
                    loff = ((unsigned)(i * a.W * ps + j) * (unsigned)Cq + (unsigned)c) * 4u + 4u * lh * pstride;
                } else {
                    loff = (unsigned)co * 4u + 4u * lh * pstride;
                }
                float rv[16], av[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = av[r] = 0.f;
                if (rrow) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        rv[r] = *(const float*)((rrow + (size_t)(((r & 3) + 8 * (r >> 2)) * pstride)) + loff);
                }
                if (a.accumulate) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        av[r] = *(const float*)((yrow + (size_t)(((r & 3) + 8 * (r >> 2)) * pstride)) + loff);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[m][n][r] + bv;
                    if (rrow) v += rv[r];
                    const float neg = is_relu ? 0.f : v * slope;
                    v = v > 0.f ? v : neg;
                    if (a.accumulate) v += av[r];
                    *(float*)((yrow + (size_t)(((r & 3) + 8 * (r >> 2)) * pstride)) + loff) = v;
                }
            }
        }
        if (a.stats_part) {
            // InstanceNorm statistics of this tile, fused (nn.InstanceNorm2d after the DGB convs, sftmd_arch.py:813): a
            // lane holds 32 pixels (2 rows x 16) of channel co; two-pass (mean, M2) in registers, then Chan merges:
            // with the other half-wave (shuffle), across the waves (LDS), and later across tiles (k_instnorm_merge_tiles).
            float* sStat = (float*)smem;                           // [waves][NTILE][2]; the MFMA operands are dead by now
            const int rows = (y0 + 2 * wvu < a.H ? 1 : 0) + (y0 + 2 * wvu + 1 < a.H ? 1 : 0);
            float mean_l[NT], m2_l[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float bv = a.bias ? a.bias[n0 + 32 * n + li] : 0.f;
                float sum = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    if (y0 + 2 * wvu + m < a.H)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sum += acc[m][n][r] + bv;
                const float mu = rows ? sum / (16.f * (float)rows) : 0.f;
                float q = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    if (y0 + 2 * wvu + m < a.H)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float d = acc[m][n][r] + bv - mu;
                            q = fmaf(d, d, q);
                        }
                // the other half-wave holds the same channel, the same number of (other) pixels: equal-count Chan merge
                const float mu2 = __shfl_xor(mu, 32, 64), q2 = __shfl_xor(q, 32, 64);
                const float dlt = mu2 - mu;
                mean_l[n] = 0.5f * (mu + mu2);
                m2_l[n] = q + q2 + dlt * dlt * (8.f * (float)rows);      // n1*n2/(n1+n2) = 16*rows/2
            }
            __syncthreads();                                       // every wave is done with the last chunk's LDS reads
            if (lh == 0) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    sStat[(wv * NTILE + 32 * n + li) * 2 + 0] = mean_l[n];
                    sStat[(wv * NTILE + 32 * n + li) * 2 + 1] = m2_l[n];
                }
            }
            __syncthreads();
            if (tid < NTILE) {
                float cnt = 0.f, mu = 0.f, m2 = 0.f;
                for (int w = 0; w < NTHR / 64; ++w) {
                    const int rw = (y0 + 2 * w < a.H ? 1 : 0) + (y0 + 2 * w + 1 < a.H ? 1 : 0);
                    if (rw == 0) continue;
                    const float nb = 32.f * (float)rw, mb = sStat[(w * NTILE + tid) * 2], m2b = sStat[(w * NTILE + tid) * 2 + 1];
                    const float tot = cnt + nb, delta = mb - mu;
                    mu += delta * (nb / tot);
                    m2 += m2b + delta * delta * (cnt * nb / tot);
                    cnt = tot;
                }
                float* dst = a.stats_part + (((size_t)b * (tiles_x * tiles_y) + tile) * a.Cout + n0 + tid) * 2;
                dst[0] = mu;
                dst[1] = m2;
            }
        }
        CM_T_MARK(6);
        CM_T_FLUSH();
        return;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int gy = y0 + 2 * wv + m;
        if (gy >= a.H) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = n0 + 32 * n + li;
            const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gx = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (gx >= a.W) continue;
                const size_t pidx = (((size_t)b * a.H + gy) * a.W + gx) * a.Cout + co;
                float v = acc[m][n][r] + bv;
                if (a.residual) v += a.residual[pidx];
                v = dasr_act(v, a.act);
                size_t o = pidx;
                if (UNMASK) {            // dgrad fused with the producer's activation backward (+ inverse PixelShuffle)
                    v *= dasr_act_grad_from_out(a.mask_src[pidx], a.mask_act);
                    if (a.unps_r > 1) {
                        const int ur = a.unps_r;
                        o = ((((size_t)b * (a.H / ur) + gy / ur) * (a.W / ur) + gx / ur) * a.Cout + co) * (ur * ur) +
                            (gy % ur) * ur + (gx % ur);
                    }
                }
                if (a.ps_r > 1) {
                    int c = co / rr, i = (co / a.ps_r) % a.ps_r, j = co % a.ps_r;
                    o = (((size_t)b * a.H * a.ps_r + (size_t)gy * a.ps_r + i) * ((size_t)a.W * a.ps_r) +
                         (size_t)gx * a.ps_r + j) * (a.Cout / rr) + c;
                }
                if (a.accumulate) v += a.y[o];
                a.y[o] = v;
            }
        }
    }
    CM_T_MARK(6);
    CM_T_FLUSH();
}

static size_t conv_mfma_lds(int NT, int TH) {
    return sizeof(float) * (size_t)((TH + 2) * CM_HALO_W * CM_CKP + 9 * 32 * NT * CM_CKP);
}

// (a sample's input is addressed through a buffer descriptor with 32-bit byte offsets: below 2 GiB)
bool conv_mfma_supported(const ConvGeom& g) {
    return g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && !g.transposed && (g.Cin % CM_CK) == 0 &&
           (g.Cout % 32) == 0 && g.H == g.Ho && g.W == g.Wo && (size_t)g.H * g.W * g.Cin * sizeof(float) < ((size_t)1 << 31) &&
           (size_t)g.Cin * g.Cout * sizeof(float) < ((size_t)1 << 23);      // one tap of the kernel: a 24-bit multiply
}

static int cm_tile_rows(int B, int H, int W, int Cout) {
    const bool nt2 = (Cout % 64) == 0;
    const long wg16 = (long)((W + CM_TW - 1) / CM_TW) * ((H + 15) / 16) * B * (Cout / (nt2 ? 64 : 32));
    return ((H % 16) == 0 && (wg16 % 256 == 0 || wg16 >= 2048)) ? 16 : 8;
}

template <int WMODE>
static int launch_conv_mfma(ConvMfmaArgs& a, void* stream) {
    // 16-row tiles (one 512-thread workgroup per CU) stage every weight slice once for twice the pixels; 8-row
    // tiles (two 256-thread workgroups per CU) overlap one workgroup's barriers / prologue / epilogue with the
    // other's matrix work and halve the tail.  Measured (B=16): 16 rows win when the 16-row grid fills whole rounds
    // of the 256 CUs or is many rounds long, 8 rows win otherwise (64->64 at 128x160: 264 -> 238 us).
    const bool nt2 = (a.Cout % 64) == 0;
    const int TH = cm_tile_rows(a.B, a.H, a.W, a.Cout);
    int tiles = ((a.W + CM_TW - 1) / CM_TW) * ((a.H + TH - 1) / TH);
    const int G8 = (tiles * a.B + 7) / 8 * 8;                 // pixel tiles, padded to whole rounds over the 8 XCDs
    const dim3 grid(G8 * (a.Cout / (nt2 ? 64 : 32)));
    const dim3 block(32 * TH);
    const size_t lds = conv_mfma_lds(nt2 ? 2 : 1, TH);
    const bool um = a.mask_src != nullptr;
#define CM_GO(NTv, THv, UMv) DASR_LAUNCH((k_conv3x3_mfma<NTv, WMODE, THv, UMv>), grid, block, lds, stream, a)
    if (!um) {
        if (nt2) { if (TH == 16) CM_GO(2, 16, false); else CM_GO(2, 8, false); }
        else     { if (TH == 16) CM_GO(1, 16, false); else CM_GO(1, 8, false); }
    } else {
        if (nt2) { if (TH == 16) CM_GO(2, 16, true); else CM_GO(2, 8, true); }
        else     { if (TH == 16) CM_GO(1, 16, true); else CM_GO(1, 8, true); }
    }
#undef CM_GO
    DASR_RETURN_LAUNCH_STATUS();
}

int conv_mfma_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, const float* residual,
                  float* y, int act, int ps_r, void* stream) {
    ConvMfmaArgs a{x, w, bias, residual, y, g.B, g.H, g.W, g.Cin, g.Cout, act, ps_r, 0, nullptr, 0, 1, nullptr};
    return launch_conv_mfma<0>(a, stream);
}

// ---- forward + InstanceNorm statistics of the output (the DGB convs) -------------------------------------------------
static int conv_mfma_tile_rows(const ConvGeom& g) { return cm_tile_rows(g.B, g.H, g.W, g.Cout); }
bool conv_mfma_fwd_stats_supported(const ConvGeom& g) { return conv_mfma_supported(g) && (g.W % CM_TW) == 0; }
size_t conv_mfma_fwd_stats_workspace(const ConvGeom& g) {
    const int th = conv_mfma_tile_rows(g);
    return sizeof(float) * 2 * (size_t)g.B * (g.W / CM_TW) * ((g.H + th - 1) / th) * g.Cout;
}
// mean / var per (b, c) from the per-tile (mean, M2) records: the same Chan recurrence as k_instnorm_merge, tile by tile
__global__ void __launch_bounds__(256) k_instnorm_merge_tiles(const float* __restrict__ part, float* __restrict__ mean,
                                                              float* __restrict__ var, int H, int W, int C, int th,
                                                              int tiles_x, int tiles_y, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = i % C, b = i / C, nt = tiles_x * tiles_y;
    float cnt = 0.f, mu = 0.f, m2 = 0.f;
    for (int k0 = 0; k0 < nt; k0 += 16) {
        float2 rec[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + u < nt ? k0 + u : nt - 1;
            rec[u] = *(const float2*)(part + (((size_t)b * nt + k) * C + c) * 2);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + u;
            if (k >= nt) break;
            const int ty = k / tiles_x;
            const int rows = (ty + 1) * th <= H ? th : H - ty * th;
            const float nb = (float)(rows * CM_TW);
            const float tot = cnt + nb, delta = rec[u].x - mu;
            mu += delta * (nb / tot);
            m2 += rec[u].y + delta * delta * (cnt * nb / tot);
            cnt = tot;
        }
    }
    mean[i] = mu;
    var[i] = m2 / (float)(H * W);
}
int conv_mfma_fwd_stats(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, float* mean,
                        float* var, void* workspace, void* stream) {
    ConvMfmaArgs a{x, w, bias, nullptr, y, g.B, g.H, g.W, g.Cin, g.Cout, DASR_ACT_NONE, 1, 0, nullptr, 0, 1,
                   (float*)workspace};
    int rc = launch_conv_mfma<0>(a, stream);
    if (rc) return rc;
    const int th = conv_mfma_tile_rows(g), n = g.B * g.Cout;
    DASR_LAUNCH(k_instnorm_merge_tiles, dim3(dasr_cdiv((size_t)n, 256)), dim3(256), 0, stream, (const float*)workspace, mean,
                var, g.H, g.W, g.Cout, th, g.W / CM_TW, (g.H + th - 1) / th, n);
    DASR_RETURN_LAUNCH_STATUS();
}

// dx[p, ci] (+)= sum_{tap, co} dconv[p - off(tap), co] * W[tap][ci][co]: a 3x3 conv of dconv (channels Cout) to Cin
int conv_mfma_dgrad(const ConvGeom& g, const float* dconv, const float* w, float* dx, int accumulate,
                    const float* mask_src, int mask_act, int unps_r, void* stream) {
    ConvMfmaArgs a{dconv, w, nullptr, nullptr, dx, g.B, g.H, g.W, g.Cout, g.Cin, DASR_ACT_NONE, 1, accumulate,
                   mask_src, mask_act, unps_r < 1 ? 1 : unps_r, nullptr};
    return launch_conv_mfma<1>(a, stream);
}
bool conv_mfma_dgrad_supported(const ConvGeom& g) {
    return g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && !g.transposed && (g.Cout % CM_CK) == 0 &&
           (g.Cin % 32) == 0 && g.H == g.Ho && g.W == g.Wo && (size_t)g.H * g.W * g.Cout * sizeof(float) < ((size_t)1 << 31) &&
           (size_t)g.Cin * g.Cout * sizeof(float) < ((size_t)1 << 23);
}

// ------------------------------------------------------------------------------------------ wgrad
// Workgroup = 4 waves, owns a (32*MT) x (32*NTW) block of (ci, co) for all 9 taps and walks a strip of
// 2 x 32 pixel tiles (51 KB of LDS: three workgroups per CU) with the accumulators resident; K = pixels, two per MFMA step (h = lane>>5 picks the
// pixel of the pair).  A fragment: lane i reads channel ci0+i of pixel (p + off(tap)) -> 32 consecutive
// floats per half-wave (conflict-free ds_read_b32); B fragment likewise from the dconv tile.  Each strip
// writes its partial dW as a slab; k_wgrad_reduce sums the slabs in a fixed order (bitwise reproducible).
#define WG_TW 32
// tile rows per (MT, NTW): fewer channels per workgroup -> less LDS per pixel -> taller tiles, so that the fixed
// per-tile cost (two barriers, staging latency) is amortised over the same amount of MFMA work
__host__ __device__ constexpr int wg_th(int MT, int NTW) { return MT * NTW == 4 ? 2 : (MT * NTW == 2 ? 4 : 8); }

struct WgradArgs {
    const float* x;      // [B,H,W,Cin]
    const float* dy;     // [B,H,W,Cout]
    float* slabs;        // [P][9][Cin][Cout]
    float* bslabs;       // [P][Cout] or null: column sums of dy (bias gradient partials) written by the ci-group-0 workgroups
    int B, H, W, Cin, Cout;
    int P, ntiles;
};

template <int MT, int NTW>
__global__ void __launch_bounds__(256, 2) k_conv3x3_wgrad_mfma(WgradArgs a) {
    DASR_DYN_SMEM(smem);
    constexpr int CIG = 32 * MT, COG = 32 * NTW;
    constexpr int WG_TH = wg_th(MT, NTW);
    constexpr int G = 4 / (MT * NTW);           // wave groups sharing one (mt, nt) pair, splitting the taps
    constexpr int NACC = (9 + G - 1) / G;
    float* sX = (float*)smem;                                   // [(TH+2)*(TW+2)][CIG]
    float* sD = sX + (WG_TH + 2) * (WG_TW + 2) * CIG;           // [TH*TW][COG]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int pair = wv % (MT * NTW), grp = wv / (MT * NTW);
    const int mt = pair / NTW, nt = pair % NTW;
    const int cgroups = a.Cout / COG;
    const int ci0 = (blockIdx.x / cgroups) * CIG, co0 = (blockIdx.x % cgroups) * COG;
    const int tiles_x = (a.W + WG_TW - 1) / WG_TW, tiles_y = (a.H + WG_TH - 1) / WG_TH;

    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bool do_bias = a.bslabs != nullptr && ci0 == 0 && tid < COG;
    float bsum = 0.f;

    for (int tile = blockIdx.y; tile < a.ntiles; tile += a.P) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int x0 = tx * WG_TW, y0 = ty * WG_TH;
        // Stage the tile: ALL global loads of a thread are issued before the first LDS write (a "load; wait; write"
        // loop is a chain of ~13 dependent L2/HBM round trips per tile - measured 10 us of a 25 us tile).  They are
        // issued before the barrier, so their latency also covers the wait for the slowest wave of the last tile.
        constexpr int NX = (WG_TH + 2) * (WG_TW + 2) * (CIG / 4), NXI = (NX + 255) / 256;
        constexpr int ND = WG_TH * WG_TW * (COG / 4), NDI = (ND + 255) / 256;
        // the 64x64 block already holds 144 accumulator registers: its x loads go in two batches (256-VGPR budget)
        constexpr int XA = MT * NTW == 4 ? (NDI >= 8 ? 1 : 8 - NDI) : NXI;
        float4 vx[XA > NXI - XA ? XA : NXI - XA], vd[NDI];
        auto ldx = [&](int u) {
            const int idx = tid + 256 * u;
            const int c4 = idx % (CIG / 4), pix = idx / (CIG / 4);
            const int gy = y0 + pix / (WG_TW + 2) - 1, gx = x0 + pix % (WG_TW + 2) - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < NX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                v = *(const float4*)(a.x + (((size_t)b * a.H + gy) * a.W + gx) * a.Cin + ci0 + 4 * c4);
            return v;
        };
        auto stx = [&](int u, float4 v) {
            const int idx = tid + 256 * u;
            if (idx < NX) *(float4*)(sX + (idx / (CIG / 4)) * CIG + 4 * (idx % (CIG / 4))) = v;
        };
#pragma unroll
        for (int u = 0; u < XA; ++u) vx[u] = ldx(u);
#pragma unroll
        for (int u = 0; u < NDI; ++u) {
            const int idx = tid + 256 * u;
            const int c4 = idx % (COG / 4), pix = idx / (COG / 4);
            const int gy = y0 + pix / WG_TW, gx = x0 + pix % WG_TW;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < ND && gy < a.H && gx < a.W)
                v = *(const float4*)(a.dy + (((size_t)b * a.H + gy) * a.W + gx) * a.Cout + co0 + 4 * c4);
            vd[u] = v;
        }
        __syncthreads();                        // every wave is done with the previous tile
#pragma unroll
        for (int u = 0; u < XA; ++u) stx(u, vx[u]);
#pragma unroll
        for (int u = 0; u < NDI; ++u) {
            const int idx = tid + 256 * u;
            if (idx < ND) *(float4*)(sD + (idx / (COG / 4)) * COG + 4 * (idx % (COG / 4))) = vd[u];
        }
        if (XA < NXI) {
#pragma unroll
            for (int u = XA; u < NXI; ++u) vx[u - XA] = ldx(u);
#pragma unroll
            for (int u = XA; u < NXI; ++u) stx(u, vx[u - XA]);
        }
        __syncthreads();
        if (do_bias) {                          // bias gradient: column sums of the staged dy tile
#pragma unroll 8
            for (int px = 0; px < WG_TH * WG_TW; ++px) bsum += sD[px * COG + tid];
        }
        // this wave's taps: tap(a) = a*G + grp (a < NACC); a slot past the ninth tap multiplies by zero
        int aoff[NACC];
        float amask[NACC];
#pragma unroll
        for (int t = 0; t < NACC; ++t) {
            const int tap = t * G + grp;
            const int tt = tap < 9 ? tap : 0;
            aoff[t] = ((tt / 3) * (WG_TW + 2) + tt % 3) * CIG + 32 * mt + li;
            amask[t] = tap < 9 ? 1.f : 0.f;
        }
        auto ldk = [&](int s, float& bv, float (&av)[NACC]) {
            const int py = s / (WG_TW / 2), px = 2 * (s % (WG_TW / 2)) + lh;
            bv = sD[(py * WG_TW + px) * COG + 32 * nt + li];
            const float* xs = sX + (py * (WG_TW + 2) + px) * CIG;
#pragma unroll
            for (int t = 0; t < NACC; ++t) av[t] = xs[aoff[t]];
        };
        auto mmak = [&](float bv, float (&av)[NACC]) {
            if (G > 1) {
#pragma unroll
                for (int t = 0; t < NACC; ++t) av[t] *= amask[t];
            }
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv, acc[t], 0, 0, 0);
        };
        // two statically named fragment sets: the LDS reads of pixel pair s+1 are in flight during the MFMAs of pair s
        float bv0, bv1, av0[NACC], av1[NACC];
        constexpr int NS = WG_TH * WG_TW / 2;
        ldk(0, bv0, av0);
        // fully unrolled for the square blocks: LDS addresses become immediates, no index math between the MFMAs
        // (measured: 64x64 block 864 -> 776 us at 128->128, 32x32 block 1330 -> 1155 us at 512x640; the 64x32 / 32x64
        // blocks lose 25-60 % when unrolled and keep the rolled loop)
#pragma unroll(MT == NTW ? NS / 2 : 1)
        for (int s = 0; s < NS; s += 2) {
            ldk(s + 1, bv1, av1);
            mmak(bv0, av0);
            if (s + 2 < NS) ldk(s + 2, bv0, av0);
            mmak(bv1, av1);
        }
    }
    if (do_bias) a.bslabs[(size_t)blockIdx.y * a.Cout + co0 + tid] = bsum;
    float* slab = a.slabs + (size_t)blockIdx.y * 9 * a.Cin * a.Cout;
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
        const int tap = t * G + grp;
        if (tap >= 9) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int ci = ci0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
            slab[((size_t)tap * a.Cin + ci) * a.Cout + co0 + 32 * nt + li] = acc[t][r];
        }
    }
}

// dw = sum of P slabs in a fixed order (bitwise reproducible, no atomics, nothing to clear).  A workgroup owns 256 / Q float4
// columns; its Q thread groups sum the slabs q, q + Q, ... with eight loads in flight per thread (the slabs were just written:
// they come from L2 / MALL, so the latency chain, not the bytes, is what a launch waits for) and meet in LDS.
template <int Q>
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float* __restrict__ slabs, float* __restrict__ dw, size_t n4,
                                                      int P, int gx, const float* __restrict__ bslabs,
                                                      float* __restrict__ dbias, int Cout) {
    if ((int)blockIdx.x >= gx) {                 // tail blocks: bias gradient = fixed-order sum of the P partials
        // 32 channels per block, 8 slab parts per channel, eight loads in flight per thread (a one-load-per-trip
        // loop over P = 512 partials is a 512-deep latency chain)
        __shared__ float bpart[8][32];
        const int cl = threadIdx.x & 31, q = threadIdx.x >> 5;
        const int c = (blockIdx.x - gx) * 32 + cl;
        float acc = 0.f;
        if (c < Cout) {
            for (int p0 = q; p0 < P; p0 += 64) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p0 + 8 * u < P ? bslabs[(size_t)(p0 + 8 * u) * Cout + c] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
        }
        bpart[q][cl] = acc;
        __syncthreads();
        if (q == 0 && c < Cout) {
            float r = bpart[0][cl];
#pragma unroll
            for (int t = 1; t < 8; ++t) r += bpart[t][cl];
            dbias[c] = r;
        }
        return;
    }
    constexpr int COLS = 256 / Q;
    __shared__ float4 part[256];
    const int col = threadIdx.x % COLS, q = threadIdx.x / COLS;
    const size_t i = (size_t)blockIdx.x * COLS + col;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        const float4* src = (const float4*)slabs + i;
        for (int p0 = q; p0 < P; p0 += 8 * Q) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = p0 + u * Q < P ? src[(size_t)(p0 + u * Q) * n4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (q == 0 && i < n4) {
        float4 r = part[col];
#pragma unroll
        for (int t = 1; t < Q; ++t) {
            const float4 o = part[t * COLS + col];
            r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
        }
        ((float4*)dw)[i] = r;
    }
}

// bslabs / dbias: optional bias tail
int wgrad_reduce_launch(const float* slabs, float* dw, size_t n, int P, void* stream, const float* bslabs, float* dbias,
                        int Cout) {
    // n is a multiple of 4 for every supported shape (Cout % 32 == 0)
    const size_t n4 = n / 4;
    int Q = 4;                                   // thread groups per column: enough workgroups to cover the chip
    while (Q < 32 && dasr_cdiv(n4, (size_t)(256 / Q)) < 512 && 2 * Q <= P) Q *= 2;
    const unsigned gx = dasr_cdiv(n4, (size_t)(256 / Q));
    const unsigned tail = dbias ? dasr_cdiv((size_t)Cout, 32) : 0;
    const dim3 grid(gx + tail);
    if (Q == 4)       DASR_LAUNCH((k_wgrad_reduce<4>), grid, dim3(256), 0, stream, slabs, dw, n4, P, (int)gx, bslabs, dbias, Cout);
    else if (Q == 8)  DASR_LAUNCH((k_wgrad_reduce<8>), grid, dim3(256), 0, stream, slabs, dw, n4, P, (int)gx, bslabs, dbias, Cout);
    else if (Q == 16) DASR_LAUNCH((k_wgrad_reduce<16>), grid, dim3(256), 0, stream, slabs, dw, n4, P, (int)gx, bslabs, dbias, Cout);
    else              DASR_LAUNCH((k_wgrad_reduce<32>), grid, dim3(256), 0, stream, slabs, dw, n4, P, (int)gx, bslabs, dbias, Cout);
    DASR_RETURN_LAUNCH_STATUS();
}

bool conv_mfma_wgrad_supported(const ConvGeom& g) {
    return g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && !g.transposed && (g.Cin % 32) == 0 &&
           (g.Cout % 32) == 0 && g.H == g.Ho && g.W == g.Wo;
}
static void wgrad_plan(const ConvGeom& g, int& MT, int& NTW, int& groups, int& ntiles, int& P) {
    MT = (g.Cin % 64) == 0 ? 2 : 1;
    NTW = (g.Cout % 64) == 0 ? 2 : 1;
    // 32 input channels x a multiple of 128 output channels (the upscale convs): a 32x128 block gives every wave all
    // nine taps (no zero-padded tap slots; the 32x64 block wastes one slot in ten)
    if (MT == 1 && (g.Cout % 128) == 0) NTW = 4;
    groups = (g.Cin / (32 * MT)) * (g.Cout / (32 * NTW));
    const int th = wg_th(MT, NTW);
    ntiles = g.B * ((g.H + th - 1) / th) * ((g.W + WG_TW - 1) / WG_TW);
    P = 512 / groups;
    if (P < 1) P = 1;
    if (P > ntiles) P = ntiles;
}
size_t conv_mfma_wgrad_workspace(const ConvGeom& g) {
    int MT, NTW, groups, ntiles, P;
    wgrad_plan(g, MT, NTW, groups, ntiles, P);
    return sizeof(float) * ((size_t)P * 9 * g.Cin * g.Cout + (size_t)P * g.Cout);   // weight slabs + bias partials
}
int conv_mfma_wgrad(const ConvGeom& g, const float* x, const float* dconv, float* dw, float* dbias, void* workspace,
                    void* stream) {
    int MT, NTW, groups, ntiles, P;
    wgrad_plan(g, MT, NTW, groups, ntiles, P);
    const size_t nW = (size_t)9 * g.Cin * g.Cout;
    float* slabs = (float*)workspace;
    float* bslabs = dbias ? slabs + (size_t)P * nW : nullptr;
    WgradArgs a{x, dconv, slabs, bslabs, g.B, g.H, g.W, g.Cin, g.Cout, P, ntiles};
    const int th = wg_th(MT, NTW);
    size_t lds = sizeof(float) * (size_t)((th + 2) * (WG_TW + 2) * 32 * MT + th * WG_TW * 32 * NTW);
    dim3 grid(groups, P);
    if (MT == 2 && NTW == 2) {
        DASR_LAUNCH((k_conv3x3_wgrad_mfma<2, 2>), grid, dim3(256), lds, stream, a);
    } else if (MT == 2 && NTW == 1) {
        DASR_LAUNCH((k_conv3x3_wgrad_mfma<2, 1>), grid, dim3(256), lds, stream, a);
    } else if (MT == 1 && NTW == 4) {
        DASR_LAUNCH((k_conv3x3_wgrad_mfma<1, 4>), grid, dim3(256), lds, stream, a);
    } else if (MT == 1 && NTW == 2) {
        DASR_LAUNCH((k_conv3x3_wgrad_mfma<1, 2>), grid, dim3(256), lds, stream, a);
    } else {
        DASR_LAUNCH((k_conv3x3_wgrad_mfma<1, 1>), grid, dim3(256), lds, stream, a);
    }
    return wgrad_reduce_launch(slabs, dw, nW, P, stream, bslabs, dbias, g.Cout);
}
