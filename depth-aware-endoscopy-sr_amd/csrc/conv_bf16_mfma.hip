// conv_bf16_mfma.hip — the trunk's 3x3 / stride 1 / pad 1 convolutions with bf16 activations and kernels on the bf16
// matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate: 16x the rate of the exact-fp32 MFMA of conv_mfma.hip), for
// BASELINE.json configs[2..3].  NHWC bf16 x packed bf16 kernels (dasr_weight_pack_fwd_bf16: HWIO then per-tap transpose).
//
//   forward : out[p, co]  = sum_{tap, ci} in[p + off(tap), ci] * W[tap][ci][co]       M = pixels, N = co, K = 9*Cin
//   dgrad   : the same kernel run on dconv with the taps flipped and ci/co swapped      (WMODE = 1)
//   wgrad   : dW[tap][ci][co] = sum_p in[p + off(tap), ci] * dconv[p, co]               M = ci,  N = co, K = pixels
//
// Forward / dgrad.  Workgroup = 4 waves, output tile = 8 rows x 32 columns of pixels x 32*NT output channels; wave w owns
// rows 2w, 2w+1.  K loop over 32-channel chunks: the 10 x 34 halo tile and the 9 x (32*NT) x 32 kernel slice are staged in
// LDS with an 80-byte pixel / channel stride (5 x 16 B: the 16 lanes of every ds_read_b128 lane group hit 16 distinct
// 4-bank slots).  One ds_read_b128 IS one MFMA operand: lane (i, h) holds channels 16q+8h .. +7 of pixel i (A) or of
// output channel i (B).  Software pipeline as in conv_mfma.hip: chunk c+1 is fetched into registers before the MFMAs of
// chunk c and written to LDS after them.
// Epilogue.  The D fragment has the output channel on the lane and 16 pixels in registers; stored as it stands that is
// 2 bytes per lane per store.  Instead every wave passes its rows through its own slice of LDS ([pixel][co] fp32, the
// operand buffers are dead by then) and reads them back with 8 consecutive channels of one pixel per lane: residual /
// accumulate operands are fetched and the result stored 16 bytes per lane, 8 lanes (NT=2) per 128-byte pixel.  The
// PixelShuffle(2) store gathers its 8 channels at stride 4 from the same LDS image.
//
// Weight gradient.  K = pixels, 16 per MFMA: the operand a lane needs is 8 CONSECUTIVE PIXELS of one channel, i.e. the
// transpose of NHWC.  The tiles are staged as they come ([pixel][channel], coalesced) and read with ds_read_b64_tr_b16,
// gfx950's transposing LDS read (cdna_hip_programming.md T10): two of them per operand.  Pixel strides of 64 or 192
// bytes make those reads conflict-free.  Accumulators stay resident over a strip of tiles; slabs + the fixed-order
// reduction of conv_mfma.hip (k_wgrad_reduce) finish the sum in fp32.
#include "bf16.h"
#include "conv_kernels.h"

#define CB_TW 32
#define CB_HALO_W (CB_TW + 2)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct ConvBf16Args {
    const bf16_t* x;         // [B,H,W,Cin]
    const bf16_t* w;         // packed bf16 kernel of the FORWARD conv: [2][9][..][..] (HWIO, then [tap][co][ci])
    const float* bias;       // [Cout] or null (fp32)
    const bf16_t* residual;  // [B,H,W,Cout] or null
    bf16_t* y;
    int B, H, W, Cin, Cout;
    int act, ps_r, accumulate;
};

// NW = waves per workgroup: 4 (8-row tiles, two workgroups per CU) or 8 (16-row tiles, one 512-thread workgroup per CU:
// the kernel slice - 60 % of the staged bytes - is written to LDS once for twice the pixels).  NT = 32-channel N tiles
// per workgroup: with NT = 4 (a 128-channel output in one workgroup) a wave issues 6 operand reads per 8 MFMAs instead
// of 4 per 4, and the input tile is fetched once instead of once per N slice.  LDS is the contended resource of this
// kernel (operand reads + staging writes were 82 % of the CU's LDS cycles at NW = 4 / NT = 2: 868 TF on 128 -> 128).
// CB_MTW = tile rows per wave: 2, or 4 (32-row tiles with NW = 8: 6 operand reads per 8 MFMAs, the kernel slice staged
// once per 1024 pixels - LDS cycles per MFMA drop from 82 % to 56 % of the pipe time; 128 accumulator registers).
// TD ("transposed D"): the MFMA operands swapped - D^T = W^T . X^T (both fragments have the same register layout) - puts the
// PIXEL on the lane and 4 consecutive output channels in each group of 4 registers, so bias / residual / activation / store
// can go straight from registers, 8 bytes per lane, no LDS.  Measured and NOT used: every store instruction then touches 32
// different cache lines (a lane per pixel), and the kernel was 15-30 % SLOWER than with the LDS-transposing epilogue
// (128 -> 128: 1014 vs 880 us; 64 -> 64: 366 vs 287 us).  For scale: with the epilogue compiled out entirely (garbage results)
// the same kernels take 647 / 162 us - the main loop runs at 1.2-1.3 PF, the epilogue is 26 % / 44 % of the kernel.
template <int NT, int WMODE, int NW, int CB_MTW, int CK, bool TD>
__global__ void __launch_bounds__(64 * NW, 2) k_conv3x3_bf16(ConvBf16Args a) {
    DASR_DYN_SMEM(smem);
    constexpr int NTHR = 64 * NW, CB_TH = CB_MTW * NW, CB_HALO_H = CB_TH + 2;
    constexpr int CKP = CK + 8, PPX = CK / 8;        // LDS stride (odd multiple of 16 B), 16-byte pieces per pixel / channel row
    constexpr int NTILE = 32 * NT;
    bf16_t* sIn = (bf16_t*)smem;                                  // [HALO_H*HALO_W][CKP]
    bf16_t* sW = sIn + CB_HALO_H * CB_HALO_W * CKP;            // [9][NTILE][CKP]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware block order: the N-slices of one pixel tile get workgroup ids 8 apart (same XCD, back to back)
    const int tiles_x = (a.W + CB_TW - 1) / CB_TW, tiles_y = (a.H + CB_TH - 1) / CB_TH;
    const int nsl = a.Cout / NTILE, G = tiles_x * tiles_y * a.B;
    const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
    const int gt = (kq / nsl) * 8 + xcd;
    if (gt >= G) return;
    const int tile = gt % (tiles_x * tiles_y);
    const int x0 = (tile % tiles_x) * CB_TW, y0 = (tile / tiles_x) * CB_TH;
    const int b = gt / (tiles_x * tiles_y), n0 = (kq % nsl) * NTILE;

    // The accumulators start at the bias (a lane's 16 registers of an accumulator belong to ONE output channel - four
    // consecutive ones per register quad in the transposed layout), not at zero: 64 v_add_f32 less per wave in an epilogue
    // that costs as many VALU cycles as the MFMAs of a 64-channel layer.
    f32x16 acc[CB_MTW][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float bq[16];
        if (TD) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 bv = a.bias ? *(const float4*)(a.bias + n0 + 32 * n + 8 * g4 + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);
                bq[4 * g4] = bv.x; bq[4 * g4 + 1] = bv.y; bq[4 * g4 + 2] = bv.z; bq[4 * g4 + 3] = bv.w;
            }
        } else {
            const float bv = a.bias ? a.bias[n0 + 32 * n + li] : 0.f;      // ONE load: a lane's 16 registers are one channel
#pragma unroll
            for (int r = 0; r < 16; ++r) bq[r] = bv;
        }
#pragma unroll
        for (int m = 0; m < CB_MTW; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = bq[r];
    }

    constexpr int NINP = CB_HALO_H * CB_HALO_W * PPX;             // 16-byte pieces of the halo tile
    constexpr int NWTP = 9 * NTILE * PPX;
    constexpr int NIN = (NINP + NTHR - 1) / NTHR, NWT = (NWTP + NTHR - 1) / NTHR;
    u32x4 pin[NIN], pwt[NWT];
    // Piece addressing is computed ONCE per workgroup: a byte offset per piece from the sample's / the kernel slice's base
    // (DASR_OOB for halo pixels outside the image: the buffer load returns the zero padding), so that the per-chunk loop
    // below is one add and one buffer load per piece.
    const bf16_t* wsrc = WMODE == 0 ? a.w + (size_t)9 * a.Cin * a.Cout : a.w;
    const BufRsrc rx = dasr_make_rsrc(a.x + (size_t)b * a.H * a.W * a.Cin, (size_t)a.H * a.W * a.Cin * sizeof(bf16_t));
    const BufRsrc rw = dasr_make_rsrc(wsrc, (size_t)9 * a.Cin * a.Cout * sizeof(bf16_t));
    unsigned offx[NIN], offw[NWT];
    {
        // piece u of this thread is pixel (tid / PPX + u * NTHR / PPX) of the halo tile: the pixel index is decoded ONCE (one
        // division by the tile width) and advanced by a compile-time (rows, columns) step per piece - 32-bit adds and
        // selects.  (Decoded per piece - a division, two 32-bit multiplies lowered to v_mad_u64_u32, an exec-masked select -
        // the prologue cost ~1500 cycles per wave: a third of the MFMA time of a 64-channel layer's workgroup.)
        constexpr int DPIX = NTHR / PPX, DROW = DPIX / CB_HALO_W, DCOL = DPIX % CB_HALO_W;
        const int pixb = a.Cin * (int)sizeof(bf16_t), rowb = a.W * pixb;
        int prow = (tid / PPX) / CB_HALO_W, pcol = (tid / PPX) % CB_HALO_W;
        int off = (y0 - 1 + prow) * rowb + (x0 - 1 + pcol) * pixb + 8 * (tid % PPX) * (int)sizeof(bf16_t);
#pragma unroll
        for (int u = 0; u < NIN; ++u) {
            const bool ok = ((u + 1) * NTHR <= NINP || tid + NTHR * u < NINP) && (unsigned)(y0 - 1 + prow) < (unsigned)a.H &&
                            (unsigned)(x0 - 1 + pcol) < (unsigned)a.W;
            offx[u] = ok ? (unsigned)off : DASR_OOB;
            prow += DROW;
            pcol += DCOL;
            off += DROW * rowb + DCOL * pixb;
            const bool wrap = pcol >= CB_HALO_W;
            prow += wrap ? 1 : 0;
            pcol -= wrap ? CB_HALO_W : 0;
            off += wrap ? rowb - CB_HALO_W * pixb : 0;
        }
        // kernel slice as [tap][n][k], k contiguous in both modes:
        //   forward: second half of the packed kernel, [tap][co][ci];  dgrad: first half (HWIO = [tap][n = ci_f][k = co_f]),
        //   taps flipped.  Piece idx = tid + NTHR * u: column piece idx % PPX, row (idx / PPX) % NTILE of tap idx / (NTILE * PPX)
        //   - powers of two: shifts; one 24-bit multiply for the tap.
        const int tapb = a.Cout * a.Cin * (int)sizeof(bf16_t);
#pragma unroll
        for (int u = 0; u < NWT; ++u) {
            const int idx = tid + NTHR * u;
            const int q4 = idx % PPX, nl = (idx / PPX) % NTILE, tap = idx / (NTILE * PPX);
            const int tsrc = WMODE == 0 ? tap : 8 - tap;
            offw[u] = ((u + 1) * NTHR <= NWTP || idx < NWTP)
                          ? (unsigned)(DASR_MUL24(tsrc, tapb) + ((n0 + nl) * a.Cin + 8 * q4) * (int)sizeof(bf16_t))
                          : DASR_OOB;
        }
    }
    auto prefetch = [&](int c0) {
        const unsigned cb = (unsigned)c0 * (unsigned)sizeof(bf16_t);
#pragma unroll
        for (int u = 0; u < NIN; ++u) pin[u] = dasr_buffer_load16(rx, offx[u] + cb);
#pragma unroll
        for (int u = 0; u < NWT; ++u) pwt[u] = dasr_buffer_load16(rw, offw[u] + cb);
    };
    // LDS images: piece idx of either tile goes to (idx / PPX) * CKP + 8 * (idx % PPX): a per-thread base plus a
    // compile-time step per u
    bf16_t* const lin = sIn + (tid / PPX) * CKP + 8 * (tid % PPX);
    bf16_t* const lwt = sW + (tid / PPX) * CKP + 8 * (tid % PPX);
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < NIN; ++u)
            if ((u + 1) * NTHR <= NINP || tid + NTHR * u < NINP) *(u32x4*)(lin + u * (NTHR / PPX) * CKP) = pin[u];
#pragma unroll
        for (int u = 0; u < NWT; ++u)
            if ((u + 1) * NTHR <= NWTP || tid + NTHR * u < NWTP) *(u32x4*)(lwt + u * (NTHR / PPX) * CKP) = pwt[u];
    };
    prefetch(0);
    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        __syncthreads();                       // every wave is done reading the previous chunk
        commit();
        __syncthreads();
        if (c0 + CK < a.Cin) prefetch(c0 + CK);
        // 9 taps x 2 K-steps of 16 channels; the operands of step j+1 are read while the MFMAs of step j run
        constexpr int KS = CK / 16, NJ = 9 * KS;      // K steps of 16 channels per tap, fragment steps per chunk
        auto ldfrag = [&](int j, bf16x8 (&A)[CB_MTW], bf16x8 (&Bf)[NT]) {
            const int tap = j / KS, q = j % KS;
            const int dy = tap / 3, dx = tap - 3 * dy;
#ifdef DASR_CB_FAKE_SHARE     // timing experiment only (wrong results): what sharing the A reads of a kernel row would buy (measured: nothing - the loop is not bound by LDS read bandwidth)
            if (dx == 0)
#endif
#pragma unroll
            for (int m = 0; m < CB_MTW; ++m)
                A[m] = *(const bf16x8*)(sIn + ((CB_MTW * wv + m + dy) * CB_HALO_W + li + dx) * CKP + 16 * q + 8 * lh);
#pragma unroll
            for (int n = 0; n < NT; ++n)
                Bf[n] = *(const bf16x8*)(sW + (tap * NTILE + 32 * n + li) * CKP + 16 * q + 8 * lh);
        };
        auto mma = [&](const bf16x8 (&A)[CB_MTW], const bf16x8 (&Bf)[NT]) {
#pragma unroll
            for (int m = 0; m < CB_MTW; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = TD ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bf[n], A[m], acc[m][n], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m], Bf[n], acc[m][n], 0, 0, 0);
        };
        if (NT * CB_MTW < 8) {
            bf16x8 A0[CB_MTW], B0[NT], A1[CB_MTW], B1[NT];
            ldfrag(0, A0, B0);
#pragma unroll
            for (int j = 0; j < NJ; j += 2) {
                if (j + 1 < NJ) ldfrag(j + 1, A1, B1);
                mma(A0, B0);
                if (j + 2 < NJ) ldfrag(j + 2, A0, B0);
                if (j + 1 < NJ) mma(A1, B1);
            }
        } else {
            // 128 accumulator registers: one fragment set (the SIMD's other wave covers the read latency)
            bf16x8 A0[CB_MTW], B0[NT];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                ldfrag(j, A0, B0);
                mma(A0, B0);
                DASR_SCHED_BARRIER();      // keep the scheduler from hoisting later steps' reads (register budget)
            }
        }
    }

    // ---- epilogue
#ifdef DASR_CB_NOEPI              // timing experiment only (wrong results): every accumulator stays live, nothing is stored
    {
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < CB_MTW; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[m][n][r];
        if (sum == 12345.678f) a.y[blockIdx.x] = dasr_f2bf(sum);
        return;
    }
#endif
    if (TD) {
        // Transposed accumulators (MFMA(B, A)): lane (li = pixel of the tile row, lh), registers 4g .. 4g+3 of acc[m][n] =
        // channels n0 + 32n + 8g + 4lh .. +3 - four CONSECUTIVE channels of one pixel.  NOT launched: storing straight from
        // this layout (8 bytes per lane) measured 15-30 % slower than the pixel-major layout's LDS transposition, and a
        // bounce through LDS from this layout (8-byte writes, 16-byte reads) measured equal to it (fwd 64->64 301 vs 280 us,
        // dgrad 289 vs 296 us): the epilogue's cost is not its LDS instruction count.
        const bool is_relu = a.act == DASR_ACT_RELU;
        const float slope = a.act == DASR_ACT_LRELU02 ? 0.2f : 1.f;
        const bool xin = x0 + li < a.W;
#pragma unroll
        for (int m = 0; m < CB_MTW; ++m) {
            const int gy = y0 + CB_MTW * wv + m;
            if (gy >= a.H || !xin) continue;
            bf16_t* yp = a.y + (((size_t)b * a.H + gy) * a.W + x0 + li) * a.Cout + n0 + 4 * lh;
            const bf16_t* rp = a.residual ? a.residual + (((size_t)b * a.H + gy) * a.W + x0 + li) * a.Cout + n0 + 4 * lh : nullptr;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                bf16x4 rv[4], av[4];
                if (rp) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) rv[g] = *(const bf16x4*)(rp + 32 * n + 8 * g);
                }
                if (a.accumulate) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) av[g] = *(const bf16x4*)(yp + 32 * n + 8 * g);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float o[4] = {acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
                    bf16x4 ov;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (rp) o[t] += dasr_bf2f(rv[g][t]);
                        const float neg = is_relu ? 0.f : o[t] * slope;
                        o[t] = o[t] > 0.f ? o[t] : neg;
                        if (a.accumulate) o[t] += dasr_bf2f(av[g][t]);
                        ov[t] = dasr_f2bf(o[t]);
                    }
                    *(bf16x4*)(yp + 32 * n + 8 * g) = ov;
                }
            }
        }
        return;
    }
    const int ps = a.ps_r, rr = ps * ps;
    // (ragged last tile column included: the stores are predicated per pixel.  It used to take the element-by-element
    // path below - a third of the tiles of an 80-pixel-wide image, 5.7 ms instead of 0.6 for the encoder's 128 -> 4x256
    // PixelShuffle convolution)
    const bool fast = ps == 1 || (ps == 2 && a.residual == nullptr && !a.accumulate);
    const int wvalid = a.W - x0;                     // pixels of this tile row inside the image (>= 32: all)
    if (fast) {
        // Every wave passes its rows through its OWN slice of LDS, so the hand-off between its lanes needs no s_barrier,
        // only program order (DASR_WAVE_SYNC).  One barrier first: every wave must be done with the last chunk's operands.
        __syncthreads();
        const bool is_relu = a.act == DASR_ACT_RELU;
        const float slope = a.act == DASR_ACT_LRELU02 ? 0.2f : 1.f;
        if (ps == 1 && a.residual == nullptr && !a.accumulate) {
            // plain layers (the DGB convolutions, gamma_o|beta_o, most dgrads): bias + activation in the accumulator
            // layout, rounded to bf16 BEFORE the transposition: half the LDS bytes, and the read-back is the store data
            constexpr int EPH = NTILE + 8;                            // bf16 elements per pixel row (16-byte aligned rows)
            bf16_t* sH = (bf16_t*)smem + wv * (32 * EPH);
#pragma unroll
            for (int m = 0; m < CB_MTW; ++m) {
                const int gy = y0 + CB_MTW * wv + m;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    // activation as one or two instructions per element (bias is already in the accumulator): ReLU =
                    // max(v, 0); LeakyReLU / none = max(v, slope * v) with slope 0.2 / 1 (slope <= 1).  Two elements per
                    // v_cvt_pk_bf16_f32; the halves go to their rows with ds_write_b16 / ds_write_b16_d16_hi.
                    auto put = [&](int r, float v0, float v1) {
                        const bf16x2_t pk = dasr_f2bf2(v0, v1);
                        sH[((r & 3) + 8 * (r >> 2) + 4 * lh) * EPH + 32 * n + li] = pk[0];
                        sH[(((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * lh) * EPH + 32 * n + li] = pk[1];
                    };
                    if (is_relu) {                  // one wave-uniform branch per 16 elements, not one per pair
#pragma unroll
                        for (int r = 0; r < 16; r += 2) put(r, fmaxf(acc[m][n][r], 0.f), fmaxf(acc[m][n][r + 1], 0.f));
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; r += 2)
                            put(r, fmaxf(acc[m][n][r], acc[m][n][r] * slope), fmaxf(acc[m][n][r + 1], acc[m][n][r + 1] * slope));
                    }
                }
                DASR_WAVE_SYNC();
                if (gy < a.H) {
                    constexpr int GP = NTILE / 8;
#pragma unroll
                    for (int u = 0; u < (32 * GP) / 64; ++u) {
                        const int v = lane + 64 * u, p = v / GP, cg = v % GP;
                        const bf16x8 ov = *(const bf16x8*)(sH + p * EPH + 8 * cg);
                        if (p < wvalid) *(bf16x8*)(a.y + (((size_t)b * a.H + gy) * a.W + x0 + p) * a.Cout + n0 + 8 * cg) = ov;
                    }
                }
                DASR_WAVE_SYNC();                  // the slice is rewritten by the next row
            }
            return;
        }
        // per wave: [32 pixels][NTILE + 4] fp32 (the +4 keeps the 16-byte alignment and spreads the rows over the banks)
        constexpr int EP = NTILE + 4;
        float* sE = (float*)smem + wv * (32 * EP);
#pragma unroll
        for (int m = 0; m < CB_MTW; ++m) {
            const int gy = y0 + CB_MTW * wv + m;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sE[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP + 32 * n + li] = acc[m][n][r];
            }
            DASR_WAVE_SYNC();
            if (gy < a.H) {
                if (ps == 1) {
                    constexpr int GP = NTILE / 8;                 // 8-channel groups per pixel
#pragma unroll
                    for (int u = 0; u < (32 * GP) / 64; ++u) {
                        const int v = lane + 64 * u, p = v / GP, cg = v % GP;
                        if (p >= wvalid) continue;
                        const float4 lo = *(const float4*)(sE + p * EP + 8 * cg), hi = *(const float4*)(sE + p * EP + 8 * cg + 4);
                        float o[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        const size_t idx = (((size_t)b * a.H + gy) * a.W + x0 + p) * a.Cout + n0 + 8 * cg;
                        if (a.residual) {
                            const bf16x8 rv = *(const bf16x8*)(a.residual + idx);
#pragma unroll
                            for (int t = 0; t < 8; ++t) o[t] += dasr_bf2f(rv[t]);
                        }
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const float neg = is_relu ? 0.f : o[t] * slope;
                            o[t] = o[t] > 0.f ? o[t] : neg;
                        }
                        if (a.accumulate) {
                            const bf16x8 av = *(const bf16x8*)(a.y + idx);
#pragma unroll
                            for (int t = 0; t < 8; ++t) o[t] += dasr_bf2f(av[t]);
                        }
                        bf16x8 ov;
#pragma unroll
                        for (int t = 0; t < 8; ++t) ov[t] = dasr_f2bf(o[t]);
                        *(bf16x8*)(a.y + idx) = ov;
                    }
                } else {
                    // PixelShuffle(2): out[b, 2gy+i, 2gx+j, c] = conv[b, gy, gx, 4c + 2i + j]
                    constexpr int GS = NTILE / 32;                // 8-channel groups per sub-pixel in this N slice
                    const int Cq = a.Cout / 4;
#pragma unroll
                    for (int u = 0; u < (32 * 4 * GS) / 64; ++u) {
                        const int v = lane + 64 * u;
                        const int cg = v % GS, j = (v / GS) & 1, p = (v / (2 * GS)) % 32, i = v / (64 * GS);
                        if (p >= wvalid) continue;
                        float o[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) o[t] = sE[p * EP + 4 * (8 * cg + t) + 2 * i + j];
                        bf16x8 ov;
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const float neg = is_relu ? 0.f : o[t] * slope;
                            ov[t] = dasr_f2bf(o[t] > 0.f ? o[t] : neg);
                        }
                        const size_t idx = (((size_t)b * a.H * 2 + 2 * gy + i) * ((size_t)a.W * 2) + 2 * (x0 + p) + j) * Cq +
                                           n0 / 4 + 8 * cg;
                        *(bf16x8*)(a.y + idx) = ov;
                    }
                }
            }
            DASR_WAVE_SYNC();                      // the slice is rewritten by the next row
        }
        return;
    }
    // generic path (PixelShuffle(3), PixelShuffle with residual): element by element.  The
    // 4-rows-per-wave variant is only launched where the fast path applies (launch_conv_bf16): compiling this path for
    // it makes the compiler index its 128 accumulators dynamically (they go to scratch).
    if (CB_MTW > 2) return;
#pragma unroll
    for (int m = 0; m < CB_MTW; ++m) {
        const int gy = y0 + CB_MTW * wv + m;
        if (gy >= a.H) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = n0 + 32 * n + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gx = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (gx >= a.W) continue;
                const size_t pidx = (((size_t)b * a.H + gy) * a.W + gx) * a.Cout + co;
                float v = acc[m][n][r];
                if (a.residual) v += dasr_bf2f(a.residual[pidx]);
                v = dasr_act(v, a.act);
                size_t o = pidx;
                if (ps > 1) {
                    const int c = co / rr, i = (co / ps) % ps, j = co % ps;
                    o = (((size_t)b * a.H * ps + (size_t)gy * ps + i) * ((size_t)a.W * ps) + (size_t)gx * ps + j) *
                            (a.Cout / rr) + c;
                }
                if (a.accumulate) v += dasr_bf2f(a.y[o]);
                a.y[o] = dasr_f2bf(v);
            }
        }
    }
}

static size_t conv_bf16_lds(int NT, int NW, int MTW, int CK) {
    const size_t main_b = sizeof(bf16_t) * (size_t)((MTW * NW + 2) * CB_HALO_W * (CK + 8) + 9 * 32 * NT * (CK + 8));
    const size_t ep_b = sizeof(float) * (size_t)(NW * 32 * (32 * NT + 4));
    return main_b > ep_b ? main_b : ep_b;
}

// (tensors are addressed through buffer descriptors with 32-bit byte offsets: a sample below 2 GiB; one tap of the kernel
// below 2^23 bytes: its offset is a 24-bit multiply)
static bool conv_bf16_ranges_ok(const ConvGeom& g) {
    const size_t cmax = g.Cin > g.Cout ? g.Cin : g.Cout;
    return (size_t)g.H * g.W * cmax * sizeof(bf16_t) < ((size_t)1 << 31) && (size_t)g.Cin * g.Cout * sizeof(bf16_t) < ((size_t)1 << 23);
}
bool conv_bf16_supported(const ConvGeom& g) {
    return g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && !g.transposed && (g.Cin % 32) == 0 &&
           (g.Cout % 32) == 0 && g.H == g.Ho && g.W == g.Wo && conv_bf16_ranges_ok(g);
}
bool conv_bf16_dgrad_supported(const ConvGeom& g) {
    return g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && !g.transposed && (g.Cout % 32) == 0 &&
           (g.Cin % 32) == 0 && g.H == g.Ho && g.W == g.Wo && conv_bf16_ranges_ok(g);
}

// Tile variants measured on the MI355X at the x4 / B=32 shapes (tools/bench_ops_bf16.py, profiles/r02_conv_bf16_variants.txt):
// 8-row tiles with two 256-thread workgroups per CU beat one 512-thread workgroup with 16- or 32-row tiles, and two rows x
// two N tiles per wave beat 128-accumulator register tiles (their lower LDS traffic did not pay: the limiter was the
// epilogue, see TD above).
template <int WMODE>
static int launch_conv_bf16(ConvBf16Args& a, void* stream) {
    const int NT = (a.Cout % 64) == 0 ? 2 : 1;
    const int TH = 8;
    const int tiles = ((a.W + CB_TW - 1) / CB_TW) * ((a.H + TH - 1) / TH);
    const int G8 = (tiles * a.B + 7) / 8 * 8;                 // pixel tiles, padded to whole rounds over the 8 XCDs
    const dim3 grid(G8 * (a.Cout / (32 * NT)));
    const size_t lds = conv_bf16_lds(NT, 4, 2, 32);
    if (NT == 2) DASR_LAUNCH((k_conv3x3_bf16<2, WMODE, 4, 2, 32, false>), grid, dim3(256), lds, stream, a);
    else         DASR_LAUNCH((k_conv3x3_bf16<1, WMODE, 4, 2, 32, false>), grid, dim3(256), lds, stream, a);
    DASR_RETURN_LAUNCH_STATUS();
}

int conv_bf16_fwd(const ConvGeom& g, const bf16_t* x, const bf16_t* w, const float* bias, const bf16_t* residual,
                  bf16_t* y, int act, int ps_r, void* stream) {
    if (conv_bf16_v2_supported(g.H, g.W, g.Cin, g.Cout, ps_r, residual != nullptr, false))
        return conv_bf16_v2_fwd(g, x, w, bias, residual, y, act, ps_r, stream);
    ConvBf16Args a{x, w, bias, residual, y, g.B, g.H, g.W, g.Cin, g.Cout, act, ps_r, 0};
    return launch_conv_bf16<0>(a, stream);
}
// dx[p, ci] (+)= sum_{tap, co} dconv[p - off(tap), co] * W[tap][ci][co]: a 3x3 conv of dconv (channels Cout) to Cin
int conv_bf16_dgrad(const ConvGeom& g, const bf16_t* dconv, const bf16_t* w, bf16_t* dx, int accumulate, void* stream) {
    if (conv_bf16_v2_supported(g.H, g.W, g.Cout, g.Cin, 1, false, accumulate != 0))
        return conv_bf16_v2_dgrad(g, dconv, w, dx, accumulate, stream);
    ConvBf16Args a{dconv, w, nullptr, nullptr, dx, g.B, g.H, g.W, g.Cout, g.Cin, DASR_ACT_NONE, 1, accumulate};
    return launch_conv_bf16<1>(a, stream);
}

// ------------------------------------------------------------------------------------------ wgrad
#define WB_TW 32
__host__ __device__ constexpr int wb_th(int MT, int NTW) { return MT * NTW == 4 ? 4 : 8; }
// LDS pixel stride in bf16 elements: bytes = 64 (mod 128), so that the four pixel rows of a transposed-read block
// land in four disjoint 64-byte bank ranges
__host__ __device__ constexpr int wb_stride(int ch) { return ch == 32 ? 32 : ch + 32; }

struct WgradBf16Args {
    const bf16_t* x;     // [B,H,W,Cin]
    const bf16_t* dy;    // [B,H,W,Cout]
    float* slabs;        // [P][9][Cin][Cout]
    float* bslabs;       // [P][Cout] or null
    int B, H, W, Cin, Cout;
    int P, ntiles;
};

template <int MT, int NTW>
__global__ void __launch_bounds__(256, 2) k_conv3x3_wgrad_bf16(WgradBf16Args a) {
    DASR_DYN_SMEM(smem);
    constexpr int CIG = 32 * MT, COG = 32 * NTW, NTHR = 256;
    constexpr int TH = wb_th(MT, NTW);
    constexpr int SXP = wb_stride(CIG), SDP = wb_stride(COG);
    constexpr int G = 4 / (MT * NTW);           // waves sharing one (mt, nt) pair: they split the K-steps (pixels) of a tile
    constexpr int NACC = 9;                     // every wave holds all nine taps (see the K loop)
    bf16_t* sX = (bf16_t*)smem;                                   // [(TH+2)*(TW+2)][SXP]
    bf16_t* sD = sX + (TH + 2) * (WB_TW + 2) * SXP;               // [TH*TW][SDP]
    // wave-uniform values in SGPRs: the tap split below branches on them around MFMAs, and an MFMA ignores EXEC - a
    // condition the compiler believes to be per-lane would be "guarded" by an exec mask only
    const int tid = threadIdx.x, lane = tid & 63, wv = DASR_UNIFORM((int)(tid >> 6));
    const int li = lane & 31, lh = lane >> 5;
    const int pair = wv % (MT * NTW), grp = wv / (MT * NTW);
    const int mt = pair / NTW, nt = pair % NTW;
    const int cgroups = a.Cout / COG;
    const int ci0 = (blockIdx.x / cgroups) * CIG, co0 = (blockIdx.x % cgroups) * COG;
    const int tiles_x = (a.W + WB_TW - 1) / WB_TW, tiles_y = (a.H + TH - 1) / TH;

    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bool do_bias = a.bslabs != nullptr && ci0 == 0 && tid < COG;
    float bsum = 0.f;
    // transposed-read lane roles: lane 4q+p of its 16-lane group passes the address of pixel row q, channels 4p..4p+3 of the
    // group's 16 channels; groups 0,1 cover channels 0..15 / 16..31 of pixels 0..7 of the K-step, groups 2,3 pixels 8..15
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    const int xoff = 32 * mt + 16 * tg + 4 * tp, doff = 32 * nt + 16 * tg + 4 * tp;

    // Staging addresses, computed once: piece u of this thread sits at a fixed byte offset from the tile's origin pixel
    // (relx / reld), in a fixed tile column (colx / cold) and at a fixed LDS address; per tile only the origin changes.
    // Rows outside the image fall outside the sample's buffer range (negative offsets wrap to >= 2^31) and load zeros by
    // themselves; columns outside the image are turned into DASR_OOB by one compare.
    constexpr int NX = (TH + 2) * (WB_TW + 2) * (CIG / 8), NXI = (NX + 255) / 256;
    constexpr int ND = TH * WB_TW * (COG / 8), NDI = (ND + 255) / 256;
    int relx[NXI], colx[NXI], reld[NDI], cold[NDI];
    bf16_t* ldsx[NXI];
    bf16_t* ldsd[NDI];
#pragma unroll
    for (int u = 0; u < NXI; ++u) {
        const int idx = tid + 256 * u;
        const int c8 = idx % (CIG / 8), pix = idx / (CIG / 8);
        const int py = pix / (WB_TW + 2) - 1, px = pix % (WB_TW + 2) - 1;
        relx[u] = ((py * a.W + px) * a.Cin + ci0 + 8 * c8) * (int)sizeof(bf16_t);
        colx[u] = idx < NX ? px : -(1 << 20);                    // a column that is never inside the image
        ldsx[u] = sX + pix * SXP + 8 * c8;
    }
#pragma unroll
    for (int u = 0; u < NDI; ++u) {
        const int idx = tid + 256 * u;
        const int c8 = idx % (COG / 8), pix = idx / (COG / 8);
        const int py = pix / WB_TW, px = pix % WB_TW;
        reld[u] = ((py * a.W + px) * a.Cout + co0 + 8 * c8) * (int)sizeof(bf16_t);
        cold[u] = idx < ND ? px : -(1 << 20);
        ldsd[u] = sD + pix * SDP + 8 * c8;
    }
    for (int tile = blockIdx.y; tile < a.ntiles; tile += a.P) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int x0 = tx * WB_TW, y0 = ty * TH;
        const BufRsrc rx = dasr_make_rsrc(a.x + (size_t)b * a.H * a.W * a.Cin, (size_t)a.H * a.W * a.Cin * sizeof(bf16_t));
        const BufRsrc rd = dasr_make_rsrc(a.dy + (size_t)b * a.H * a.W * a.Cout, (size_t)a.H * a.W * a.Cout * sizeof(bf16_t));
        const int ox = (y0 * a.W + x0) * a.Cin * (int)sizeof(bf16_t), od = (y0 * a.W + x0) * a.Cout * (int)sizeof(bf16_t);
        u32x4 vx[NXI], vd[NDI];
#pragma unroll
        for (int u = 0; u < NXI; ++u)
            vx[u] = dasr_buffer_load16(rx, (unsigned)(x0 + colx[u]) < (unsigned)a.W ? (unsigned)(ox + relx[u]) : DASR_OOB);
#pragma unroll
        for (int u = 0; u < NDI; ++u)
            vd[u] = dasr_buffer_load16(rd, (unsigned)(x0 + cold[u]) < (unsigned)a.W ? (unsigned)(od + reld[u]) : DASR_OOB);
        __syncthreads();                        // every wave is done with the previous tile
#pragma unroll
        for (int u = 0; u < NXI; ++u)
            if ((u + 1) * 256 <= NX || tid + 256 * u < NX) *(u32x4*)ldsx[u] = vx[u];
#pragma unroll
        for (int u = 0; u < NDI; ++u)
            if ((u + 1) * 256 <= ND || tid + 256 * u < ND) *(u32x4*)ldsd[u] = vd[u];
        __syncthreads();
        if (do_bias) {                          // bias gradient: column sums of the staged dy tile
#pragma unroll 8
            for (int px = 0; px < TH * WB_TW; ++px) bsum += dasr_bf2f(sD[px * SDP + tid]);
        }
        // K-step s = 16 consecutive pixels of one tile row.  EXEC is all ones here (the tile loop and the tap split are
        // wave-uniform), as ds_read_b64_tr_b16 requires.
        constexpr int NS = TH * (WB_TW / 16);
#pragma unroll 2
        for (int s0 = 0; s0 < NS; s0 += G) {
            const int s = s0 + grp;
            const int py = s / (WB_TW / 16), px0 = 16 * (s % (WB_TW / 16)) + 8 * lh + tq;
            const bf16_t* dp = sD + (py * WB_TW + px0) * SDP + doff;
            const bf16x4 b0 = lds_read_tr16(dp), b1 = lds_read_tr16(dp + 4 * SDP);
            bf16x8 bv;
#pragma unroll
            for (int t = 0; t < 4; ++t) { bv[t] = b0[t]; bv[4 + t] = b1[t]; }
            const bf16_t* xp = sX + (py * (WB_TW + 2) + px0) * SXP + xoff;
            // All nine taps in this wave: the three taps of a kernel row read 8-pixel windows that start one pixel
            // apart, so ONE 12-pixel read per kernel row (three transposed reads: pixels 0-3, 4-7, 8-11 of this lane's
            // half K-step) serves all three - dx = 0: dwords 0..3, dx = 2: dwords 1..4, dx = 1: v_alignbit of neighbours -
            // instead of two reads per tap.  The A-operand reads were 9 of the 10 KB of LDS traffic per K-step and wave
            // (1.1 KB per MFMA: over the 128 B/clk of the LDS with four waves issuing one MFMA per 32 clk each).  Blocks
            // with fewer than four (mt, nt) pairs used to split the TAPS over their waves, which cannot share reads
            // (1.8 KB per MFMA, 350 TF on 32->32): they split the K-steps instead and add their accumulators at the end.
            // (Pixels 10, 11 of the last window of a tile row belong to the next row / lie past the tile: read, unused.)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const bf16_t* ap = xp + dy * (WB_TW + 2) * SXP;
                const bf16x4 a0 = lds_read_tr16(ap), a1 = lds_read_tr16(ap + 4 * SXP), a2 = lds_read_tr16(ap + 8 * SXP);
                unsigned R[6];
                __builtin_memcpy(&R[0], &a0, 8);
                __builtin_memcpy(&R[2], &a1, 8);
                __builtin_memcpy(&R[4], &a2, 8);
                unsigned V0[4] = {R[0], R[1], R[2], R[3]}, V2[4] = {R[1], R[2], R[3], R[4]}, V1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) V1[e] = (R[e] >> 16) | (R[e + 1] << 16);
                bf16x8 av0, av1, av2;
                __builtin_memcpy(&av0, V0, 16);
                __builtin_memcpy(&av1, V1, 16);
                __builtin_memcpy(&av2, V2, 16);
                acc[3 * dy + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av0, bv, acc[3 * dy + 0], 0, 0, 0);
                acc[3 * dy + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av1, bv, acc[3 * dy + 1], 0, 0, 0);
                acc[3 * dy + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av2, bv, acc[3 * dy + 2], 0, 0, 0);
            }
        }
    }
    if (G > 1) {
        // waves grp = 1..G-1 hand their accumulators to wave grp = 0 of the same (mt, nt) pair, one tap at a time through
        // (G-1) * pairs * 4 KB of the (now idle) staging LDS
        float* sR = (float*)smem;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            __syncthreads();
            if (grp > 0) {
                float* q = sR + ((grp - 1) * (MT * NTW) + pair) * 1024 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) q[64 * r] = acc[t][r];
            }
            __syncthreads();
            if (grp == 0) {
                for (int gg = 0; gg < G - 1; ++gg) {
                    const float* q = sR + (gg * (MT * NTW) + pair) * 1024 + lane;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += q[64 * r];
                }
            }
        }
    }
    if (do_bias) a.bslabs[(size_t)blockIdx.y * a.Cout + co0 + tid] = bsum;
    float* slab = a.slabs + (size_t)blockIdx.y * 9 * a.Cin * a.Cout;
    if (grp != 0) return;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
            slab[((size_t)tap * a.Cin + ci) * a.Cout + co0 + 32 * nt + li] = acc[tap][r];
        }
    }
}

bool conv_bf16_wgrad_supported(const ConvGeom& g) {
    return g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && !g.transposed && (g.Cin % 32) == 0 &&
           (g.Cout % 32) == 0 && g.H == g.Ho && g.W == g.Wo;
}
static void wb_plan(const ConvGeom& g, int& MT, int& NTW, int& groups, int& ntiles, int& P) {
    MT = (g.Cin % 64) == 0 ? 2 : 1;
    NTW = (g.Cout % 64) == 0 ? 2 : 1;
    groups = (g.Cin / (32 * MT)) * (g.Cout / (32 * NTW));
    const int th = wb_th(MT, NTW);
    ntiles = g.B * ((g.H + th - 1) / th) * ((g.W + WB_TW - 1) / WB_TW);
    P = 512 / groups;
    if (P < 1) P = 1;
    if (P > ntiles) P = ntiles;
}
size_t conv_bf16_wgrad_workspace(const ConvGeom& g) {
    int MT, NTW, groups, ntiles, P;
    wb_plan(g, MT, NTW, groups, ntiles, P);
    return sizeof(float) * ((size_t)P * 9 * g.Cin * g.Cout + (size_t)P * g.Cout);
}
int conv_bf16_wgrad(const ConvGeom& g, const bf16_t* x, const bf16_t* dconv, float* dw, float* dbias, void* workspace,
                    void* stream) {
    int MT, NTW, groups, ntiles, P;
    wb_plan(g, MT, NTW, groups, ntiles, P);
    const size_t nW = (size_t)9 * g.Cin * g.Cout;
    float* slabs = (float*)workspace;
    float* bslabs = dbias ? slabs + (size_t)P * nW : nullptr;
    WgradBf16Args a{x, dconv, slabs, bslabs, g.B, g.H, g.W, g.Cin, g.Cout, P, ntiles};
    const int th = wb_th(MT, NTW);
    const size_t lds = sizeof(bf16_t) * (size_t)((th + 2) * (WB_TW + 2) * wb_stride(32 * MT) + th * WB_TW * wb_stride(32 * NTW));
    dim3 grid(groups, P);
    if (MT == 2 && NTW == 2)      DASR_LAUNCH((k_conv3x3_wgrad_bf16<2, 2>), grid, dim3(256), lds, stream, a);
    else if (MT == 2 && NTW == 1) DASR_LAUNCH((k_conv3x3_wgrad_bf16<2, 1>), grid, dim3(256), lds, stream, a);
    else if (MT == 1 && NTW == 2) DASR_LAUNCH((k_conv3x3_wgrad_bf16<1, 2>), grid, dim3(256), lds, stream, a);
    else                          DASR_LAUNCH((k_conv3x3_wgrad_bf16<1, 1>), grid, dim3(256), lds, stream, a);
    return wgrad_reduce_launch(slabs, dw, nW, P, stream, bslabs, dbias, g.Cout);
}
