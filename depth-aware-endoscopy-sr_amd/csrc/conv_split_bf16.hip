// conv_split_bf16.hip — fp32 3x3 / stride 1 / pad 1 convolutions (forward, dgrad, weight gradient) at fp32 accuracy on the
// 16-bit matrix cores: operands split into 16-bit pieces.  For the fp32 path's trunk convolutions: mlp_gamma_o | mlp_beta_o
// (normalization.py:41-42,73-74: 128 -> 128, 41 % of the x8 step), the DGB convolutions (sftmd_arch.py:811-820: 64 -> 64), the
// classic blocks and the upscale tail (:128-146, :891-908).  Two schemes, selected by the template parameter NP (pieces per
// operand): NP = 3, three bf16 pieces / six products (first); NP = 2, two scaled fp16 pieces / three products (the default of
// graph.py, below).  File layout: scale / split helpers; forward + dgrad kernels (k_conv3x3_split, _n32) with the in-place
// halo split (sp_presplit) and the shared epilogue; k_split_weights, k_absmax; the C ABI; the weight-gradient kernels
// (k_conv3x3_wgrad_split: fp32 tiles split per K-step; k_conv3x3_wgrad_split2: fp16 planes split at staging time).
//
// Why.  v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate and the fp32 kernels of conv_mfma.hip already sit at 81-85 % of
// that peak: the fp32 step has no other lever left.  An fp32 value is the exact sum of three bf16 pieces
// (x = x0 + x1 + x2, x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1): 3 x 8 significant bits + signs >= 24), a
// product of two bf16 values is exact in fp32, and
//     x * w = x0 w0 + (x0 w1 + x1 w0) + (x0 w2 + x1 w1 + x2 w0) + O(2^-26 |x w|)
// so SIX bf16 MFMAs into one fp32 accumulator give the fp32 product to a relative 2^-26 per term - below the fp32 rounding of
// the accumulation itself, which is all the exact-fp32 MFMA offers too (the same scheme k_sean_bwd_a_onehot<float> uses for
// its G operand).  6/16 of the fp32 MFMA time, and with six MFMAs per operand byte the kernel is matrix-bound with slack on
// LDS and DMA.
//
// Structure: the persistent LDS-DMA kernel of conv_bf16_v2.hip in its 8-wave form (16 x 32 output pixels x 32 NT channels per
// item, one workgroup per CU, counted vmcnt, raw barriers, transposed accumulators).  Differences:
//   * activations stay fp32 in HBM and in LDS: a halo chunk is 16 channels (64 bytes per pixel, the same LDS image and
//     swizzle as a 32-channel bf16 chunk); a lane reads its 8 channels (2 x ds_read_b128) and splits them in registers
//     (~44 VALU instructions per fragment triple, hidden under 48 MFMAs per wave and step);
//   * the kernel is split ONCE per step by dasr_conv3x3_split_weights into a K-step-major image: for each (mode, slice,
//     tap, 16-channel chunk) the three pieces of [32 NT rows][16 channels] lie contiguous and already swizzled, so a
//     K-step's slice is one linear 12 KB (6 KB) LDS-DMA copy;
//   * K-step = (tap, 16 channels): 2 rows x NT tiles x 6 products = 48 (24) MFMAs per wave, small pieces first;
//   * fp32 output through the fp32 scratch path of the epilogue, 32 bytes per lane.
//
// Round 3, second scheme: TWO fp16 pieces, THREE products (template parameter NP = 2; NP = 3 is the scheme above).
// fp16 carries 11 significant bits: with x0 = fp16(x), x1 = fp16(x - x0) the pair holds 22-23 of the 24 bits (the
// representation is exact for three values in four and one fp32 ulp off for the fourth - 0.5 ulp RMS, the size of one
// more fp32 rounding), and  x w = x0 w0 + (x0 w1 + x1 w0) + O(2^-24 |x w|).  Half the matrix work of the bf16 scheme and
// HALF the roundings into the fp32 accumulator, which is where both schemes and the exact-fp32 MFMA lose their accuracy
// (measured against float64: fp16 x 2 is the closest of the three, tests/parity_checks.py check_split_conv).  What fp16
// lacks is range (5 exponent bits): every tensor is scaled by a power of two that puts its largest magnitude in
// [2^14, 2^15) - dasr_absmax leaves the maximum in device memory, the kernels derive the scale from its exponent, nothing
// goes through the host - so that x1 stays a normal number down to 2^-18 of the tensor's maximum and loses absolute, never
// relative-to-the-sum, precision below (floor: 2^-40 of the maximum).  The result is multiplied by the exact inverse of
// the two scales in the epilogue (bias added there).  Differences of the NP = 2 kernels from the structure above: the
// landed halo chunk is split once, in place, by the threads that fetched it (sp_presplit) instead of at every read; a K-step
// is a whole kernel row (three taps, 72 MFMAs between barriers); the maxima arrive as amax buffers (dasr_common.h) filled by
// the kernels that produced the tensors, and the epilogue leaves the maximum of what it stores for the next convolution.
#include "bf16.h"
#include "conv_kernels.h"

typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

#define SP_HW 34
#define SP_NWV 8
#define SP_NTHR 512
#define SP_TH 16
#define SP_HPIX (18 * SP_HW)          // 612 halo pixels x 64 bytes (16 fp32 channels)
#define SP_NHP 5                      // halo pieces per thread per chunk (2560 >= 2448)
#define SP_HBYTES (SP_NHP * SP_NTHR * 16)
#define SP_EPITCH 144

DASR_DEVICE_CONST __attribute__((aligned(64))) unsigned sp_zero_page[64] = {0};

struct ConvSplitArgs {
    const float* x;          // [B,H,W,Cin] fp32 (dgrad: dconv, Cin = the layer's Cout)
    const bf16_t* ws;        // this mode's split kernel image (dasr_conv3x3_split_weights)
    const float* bias;       // [Cout] or null
    const float* residual;   // [B,H,W,Cout] or null (added before the activation)
    float* y;                // [B,H,W,Cout], or [B,2H,2W,Cout/4] with the PixelShuffle(2) store
    int B, H, W, Cin, Cout, accumulate, act, ps_r;
    int tiles_x, tiles_y, nsl, nitems, Q, G8;
    const float* xmax;       // NP = 2: max |x| and max |w| in device memory (dasr_absmax); NP = 3: unused
    const float* wmax;
    float* ymax;             // (may be null) raised to max |y| of what the epilogue stores
};

template <int N>
__device__ __forceinline__ void sp_wait_vm() {
#ifndef DASR_HIPEMU
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

// the same with a count that constant-folds after unrolling (a wave's slice-piece count is a run-time, wave-uniform value)
__device__ __forceinline__ void sp_wait_vm_n(int n) {
    switch (n) {
#define SP_W(N) case N: sp_wait_vm<N>(); break;
        SP_W(0) SP_W(1) SP_W(2) SP_W(3) SP_W(4) SP_W(5) SP_W(6) SP_W(7) SP_W(8) SP_W(9) SP_W(10) SP_W(11) SP_W(12)
#undef SP_W
        default: sp_wait_vm<0>(); break;
    }
}

// x = a0 + a1 + a2 exactly (each piece the bf16 rounding of what the previous ones left)
__device__ __forceinline__ void sp_split3(const float (&x)[8], bf16x8& a0, bf16x8& a1, bf16x8& a2) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const bf16x2_t h0 = dasr_f2bf2(x[e], x[e + 1]);
        const float r0 = x[e] - dasr_bf2f(h0[0]), r1 = x[e + 1] - dasr_bf2f(h0[1]);
        const bf16x2_t h1 = dasr_f2bf2(r0, r1);
        const float s0 = r0 - dasr_bf2f(h1[0]), s1 = r1 - dasr_bf2f(h1[1]);
        const bf16x2_t h2 = dasr_f2bf2(s0, s1);
        a0[e] = h0[0]; a0[e + 1] = h0[1];
        a1[e] = h1[0]; a1[e + 1] = h1[1];
        a2[e] = h2[0]; a2[e + 1] = h2[1];
    }
}

// ---- the fp16 x 2 scheme: scale exponents.  A tensor whose largest magnitude is m (bits of a non-negative float) gets
// the scale 2^k, k = 14 - floor(log2 m) clamped to [-60, 60] (m = 0: 2^60, harmless), so that m 2^k is in [2^14, 2^15).
__host__ __device__ static inline int sp_scale_exp(float m) {
    unsigned u;
    memcpy(&u, &m, 4);
    int k = 14 - ((int)((u >> 23) & 0xffu) - 127);
    return k < -60 ? -60 : (k > 60 ? 60 : k);
}
__host__ __device__ static inline float sp_pow2(int k) {        // 2^k, |k| <= 126
    const unsigned u = (unsigned)(127 + k) << 23;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// x s = a0 + a1 to 22-23 bits (a0 the fp16 rounding of x s, a1 of what a0 left)
__device__ __forceinline__ void sp_split2(const float (&x)[8], float s, f16x8& a0, f16x8& a1) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const float v0 = x[e] * s, v1 = x[e + 1] * s;
        const f16_t h0 = (f16_t)v0, g0 = (f16_t)v1;
        a0[e] = h0; a0[e + 1] = g0;
        a1[e] = (f16_t)(v0 - (float)h0); a1[e + 1] = (f16_t)(v1 - (float)g0);
    }
}
// operand fragments of one 8-element K slice and the products of a (kernel, activation) fragment pair, smallest first
template <int NP> struct SpFrag;
template <> struct SpFrag<3> {
    typedef bf16x8 type;
    typedef bf16_t elem;
    static __device__ __forceinline__ void split(const float (&x)[8], float, type (&a)[3]) { sp_split3(x, a[0], a[1], a[2]); }
    static __device__ __forceinline__ f32x16 mma(const type (&w)[3], const type (&x)[3], f32x16 c) {
        // x0 w2 + x1 w1 + x2 w0, then x0 w1 + x1 w0, then x0 w0
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], c, 0, 0, 0);
        return c;
    }
};
template <> struct SpFrag<2> {
    typedef f16x8 type;
    typedef f16_t elem;
    static __device__ __forceinline__ void split(const float (&x)[8], float s, type (&a)[2]) { sp_split2(x, s, a[0], a[1]); }
    static __device__ __forceinline__ f32x16 mma(const type (&w)[2], const type (&x)[2], f32x16 c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1], x[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[0], c, 0, 0, 0);
        return c;
    }
};
// per-launch scale state of the fp16 scheme: the activation scale and the inverse of (activation scale x kernel scale).
// The maxima come as amax buffers (dasr_common.h: word 0 = n, then n partial maxima, one per workgroup of the producer): every
// thread folds a strided share, waves meet through s_red (2 x 16 floats of LDS), and after the workgroup's next barrier
// sp_scales_take has both maxima.  NP = 3: nothing to do.
struct SpScale { float sx, inv; };
__device__ __forceinline__ float sp_amax_share(const float* buf) {
    int n;
    memcpy(&n, buf, 4);
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, buf[1 + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}
template <int NP>
__device__ __forceinline__ void sp_scales_gather(const float* xmax, const float* wmax, float* s_red) {
    if (NP == 2) {
        const float mx = sp_amax_share(xmax), mw = sp_amax_share(wmax);
        if ((threadIdx.x & 63) == 0) {
            s_red[threadIdx.x >> 6] = mx;
            s_red[16 + (threadIdx.x >> 6)] = mw;
        }
    }
}
template <int NP>
__device__ __forceinline__ SpScale sp_scales_take(const float* s_red) {
    SpScale r = {1.f, 1.f};
    if (NP == 2) {
        const int nw = (int)(blockDim.x >> 6);
        float mx = s_red[0], mw = s_red[16];
        for (int w = 1; w < nw; ++w) { mx = fmaxf(mx, s_red[w]); mw = fmaxf(mw, s_red[16 + w]); }
        const int kx = sp_scale_exp(mx), kw = sp_scale_exp(mw);
        r.sx = sp_pow2(kx);
        r.inv = sp_pow2(-(kx + kw));
    }
    return r;
}

// ---- NP = 2: the landed fp32 halo chunk is split ONCE, in place.  A tap's fragment used to be read as fp32 and split in
// registers by the wave that needed it - every halo value up to nine times (once per tap), 48 VALU per tap and wave beside
// 24 / 12 / 6 MFMAs (NT = 4 / 2 / 1).  Now each thread rewrites the five 16-byte pieces IT fetched (piece s of pixel P holds
// the channel quad c4 = s ^ key, key = (P >> 2) & 3) as [fp16(x s) of the four channels | what that rounding left]: nobody
// else touches those bytes, and the thread's own counted wait says they have landed - so the conversion sits between the
// chunk's wait and the barrier it had anyway (with an lgkmcnt(0) in front: the barrier now also publishes LDS writes).  A
// fragment (plane pl, channels 8 lh .. + 7) is two ds_read_b64: bytes 8 pl of the pieces (2 lh) ^ key and (2 lh + 1) ^ key.
// ~100 VALU per thread and chunk instead of ~430.  (First form of this: fp16 planes in 16-byte slots - one ds_read_b128 per
// fragment, but all pieces had to be read before any was written: two more barriers per chunk.  Same speed on every shape,
// A/B on one MI355X; this form has nothing to synchronise.)
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int NWV = SP_NWV>
__device__ __forceinline__ void sp_presplit(char* hb, float sx, int tid) {
    constexpr int NHP = SP_HBYTES / (1024 * NWV);              // pieces per thread: 5 (512 threads) / 10 (256)
    float4 raw[NHP];
#pragma unroll
    for (int u = 0; u < NHP; ++u) raw[u] = *(const float4*)(hb + 1024 * NWV * u + 16 * tid);
#pragma unroll
    for (int u = 0; u < NHP; ++u) {
        const float f[4] = {raw[u].x * sx, raw[u].y * sx, raw[u].z * sx, raw[u].w * sx};
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f16_t h0 = (f16_t)f[j];
            o[j] = h0;
            o[4 + j] = (f16_t)(f[j] - (float)h0);
        }
        *(f16x8*)(hb + 1024 * NWV * u + 16 * tid) = o;
    }
}
// fragment of plane pl for the lane's K-half lh at pixel P (byte address hb + 64 P)
__device__ __forceinline__ f16x8 sp_frag(const char* px, int lh, int key, int pl) {
    const f16x4 lo = *(const f16x4*)(px + ((((2 * lh) ^ key)) << 4) + 8 * pl);
    const f16x4 hi = *(const f16x4*)(px + ((((2 * lh + 1) ^ key)) << 4) + 8 * pl);
    f16x8 r;
    __builtin_memcpy(&r, &lo, 8);
    __builtin_memcpy((char*)&r + 8, &hi, 8);
    return r;
}

// ---- epilogue of one item: 32 channels per pass through [pixel][32 + 4] fp32 of the wave's scratch, 32 bytes per lane out;
// residual, activation, accumulate and the PixelShuffle(2) store as in conv_mfma.hip's epilogue.  NP = 2: the sums are in
// scaled units - times inv (the exact inverse of the two scales), plus the bias.
template <int NT, int NP, int RW = 2>
__device__ __forceinline__ void sp_epilogue(f32x16 (&acc)[RW][NT], const ConvSplitArgs& a, char* const scr, const float* sBias,
                                            const float inv, const int x0, const int y0, const int n0, const int bb,
                                            const int rbase, const int lane, const int li, const int lh, float& om) {
    const int wvalid = a.W - x0;
    const bool is_relu = a.act == DASR_ACT_RELU;
    const float slope = a.act == DASR_ACT_LRELU02 ? 0.2f : 1.f;
    const bool has_act = a.act != DASR_ACT_NONE;
#pragma unroll
    for (int m = 0; m < RW; ++m) {
        const int gy = y0 + rbase + m;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 pk = {acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
                *(f32x4*)(scr + li * SP_EPITCH + (8 * g + 4 * lh) * 4) = pk;
            }
            DASR_WAVE_SYNC();
            if (gy < a.H) {
                if (a.ps_r == 1) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int v = lane + 64 * u, pix = v >> 2, cg = v & 3;
                        if (pix >= wvalid) continue;
                        const float4 lo = *(const float4*)(scr + pix * SP_EPITCH + 32 * cg);
                        const float4 hi = *(const float4*)(scr + pix * SP_EPITCH + 32 * cg + 16);
                        float o[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        if (NP == 2) {
                            const float* bp = sBias + n0 + 32 * n + 8 * cg;
#pragma unroll
                            for (int t = 0; t < 8; ++t) o[t] = fmaf(o[t], inv, bp[t]);
                        }
                        const size_t idx = (((size_t)bb * a.H + gy) * a.W + x0 + pix) * a.Cout + n0 + 32 * n + 8 * cg;
                        if (a.residual) {
                            const float4 r0 = *(const float4*)(a.residual + idx), r1 = *(const float4*)(a.residual + idx + 4);
                            o[0] += r0.x; o[1] += r0.y; o[2] += r0.z; o[3] += r0.w;
                            o[4] += r1.x; o[5] += r1.y; o[6] += r1.z; o[7] += r1.w;
                        }
                        if (has_act) {
#pragma unroll
                            for (int t = 0; t < 8; ++t) o[t] = is_relu ? fmaxf(o[t], 0.f) : fmaxf(o[t], o[t] * slope);
                        }
                        float* yp = a.y + idx;
                        if (a.accumulate) {
                            const float4 o0 = *(const float4*)yp, o1 = *(const float4*)(yp + 4);
                            o[0] += o0.x; o[1] += o0.y; o[2] += o0.z; o[3] += o0.w;
                            o[4] += o1.x; o[5] += o1.y; o[6] += o1.z; o[7] += o1.w;
                        }
                        *(float4*)yp = make_float4(o[0], o[1], o[2], o[3]);
                        *(float4*)(yp + 4) = make_float4(o[4], o[5], o[6], o[7]);
                        om = dasr_amax4(dasr_amax4(om, make_float4(o[0], o[1], o[2], o[3])), make_float4(o[4], o[5], o[6], o[7]));
                    }
                } else {
                    // PixelShuffle(2): out[b, 2gy+i, 2gx+j, c] = conv[b, gy, gx, 4c + 2i + j]: 8 values of c per (pixel, sub-pixel)
                    const int Cq = a.Cout / 4;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int v = lane + 64 * u, j = v & 1, pix = (v >> 1) & 31, i = v >> 6;
                        if (pix >= wvalid) continue;
                        const float* sp = (const float*)(scr + pix * SP_EPITCH) + 2 * i + j;
                        float o[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const float q = NP == 2 ? fmaf(sp[4 * t], inv, sBias[n0 + 32 * n + 4 * t + 2 * i + j]) : sp[4 * t];
                            o[t] = !has_act ? q : (is_relu ? fmaxf(q, 0.f) : fmaxf(q, q * slope));
                        }
                        float* yp = a.y + (((size_t)bb * a.H * 2 + 2 * gy + i) * ((size_t)a.W * 2) + 2 * (x0 + pix) + j) * Cq +
                                    (n0 + 32 * n) / 4;
                        *(float4*)yp = make_float4(o[0], o[1], o[2], o[3]);
                        *(float4*)(yp + 4) = make_float4(o[4], o[5], o[6], o[7]);
                        om = dasr_amax4(dasr_amax4(om, make_float4(o[0], o[1], o[2], o[3])), make_float4(o[4], o[5], o[6], o[7]));
                    }
                }
            }
            DASR_WAVE_SYNC();
        }
    }
}

// RW = tile rows per wave.  2: eight waves (two per SIMD).  4 (fp16 x 2, 128 produced channels): FOUR waves, one per SIMD, each
// with 4 x 4 accumulator tiles (256 registers: the unified 512-register file of a single-wave SIMD) - a kernel fragment read
// from LDS then serves four tile rows instead of two (64 KB instead of 96 KB of operand reads per tap and CU).
// CW = 2 (fp16 x 2): the eight waves as 4 row groups x 2 channel halves - a wave owns FOUR tile rows of HALF the produced
// channels (the same 128 / 64 accumulator registers) - and a K-step is a kernel COLUMN: the six halo rows a wave's four tile
// rows meet under the three taps of a column are read once (12 KB) instead of once per tap and row pair (3 x 4 KB for half the
// rows), and each kernel fragment serves four rows: 24 KB of operand reads per wave and step (72 MFMAs) instead of 36 KB.
template <int NT, int NP, int RW = 2, int CW = 1>
__global__ void __launch_bounds__(64 * (SP_TH / RW), RW == 4 ? 1 : 2) k_conv3x3_split(ConvSplitArgs a) {
    DASR_DYN_SMEM(smem);
    typedef SpFrag<NP> F;
    static_assert(CW == 1 || (CW == 2 && NP == 2 && RW == 2 && (NT % 2) == 0), "channel halves: fp16 x 2, eight waves");
    constexpr int MR = RW * CW, NW = NT / CW;           // a wave's tile rows / 32-channel tiles
    constexpr int NWV = SP_TH / RW, NTHR = 64 * NWV, NHP = SP_HBYTES / (16 * NTHR);
    constexpr int NTILE = 32 * NT, PIECE = NTILE * 32, SLAB = NP * PIECE;    // bytes: one 16-bit piece [NTILE][16], a K-step's NP
    constexpr int WP = SLAB / 16;                                            // DMA pieces per slice: 768 / 384 (NP = 2: 512 / 256)
    // a thread's pieces of a slice: WQ whole rounds of NTHR pieces (every wave), then waves 0 .. WR-1 one more
    constexpr int WQ = WP / NTHR, WR = (WP - WQ * NTHR) / 64;
    char* const sH = smem;                              // [2][SP_HBYTES]
    // K-step = TPS taps of one 16-channel chunk between two barriers.  NP = 3: one tap (48 MFMAs per wave at NT = 4).  NP = 2:
    // a whole kernel ROW (three taps, 72 MFMAs): with half the products a one-tap step was as long as its own wait + barrier +
    // operand reads, and the compiler cannot overlap those across a barrier; inside a three-tap step it pipelines the reads
    // and splits of one tap under the MFMAs of the previous one.  Kernel slices: a ring of R step-slots (TPS slices each)
    // running D steps ahead.  (Measured for one-tap steps at NP = 2: nine slots / five steps ahead instead - no change.)
    constexpr int TPS = NP == 2 ? 3 : 1, NS = 9 / TPS;
    constexpr int R = 3, D = 2;
    char* const sW = smem + 2 * SP_HBYTES;              // [R][TPS][SLAB]
    float* const sBias = (float*)(sW + R * TPS * SLAB); // [Cout]
    float* const sRed = sBias + a.Cout;                 // [32]: the two maxima's per-wave parts, later the ymax scratch
    const int tid = threadIdx.x, lane = tid & 63, wv = DASR_UNIFORM((int)(tid >> 6));
    const int li = lane & 31, lh = lane >> 5;
    const int wr = wv / CW, wc = wv % CW;
    // DMA pieces of a kernel slice per thread, NP = 3: NT = 4: 768 = 512 + 256 (waves 0-3 two, waves 4-7 one); NT = 2: 384
    // (waves 0-5 one, waves 6-7 none).  NP = 2: NT = 4: 512 (every wave one); NT = 2: 256 (waves 0-3).  A wave's count is static.
    const int nwq = WQ + (wv < WR ? 1 : 0);
    const dasr_lds_addr_t ldsH = DASR_LDS_ADDR(sH) + 1024 * wv, ldsW = DASR_LDS_ADDR(sW) + 1024 * wv;

    const int xcd = blockIdx.x & 7, jwg = blockIdx.x >> 3;
    const int ibeg = xcd * a.Q;
    const int iend = ibeg + a.Q < a.nitems ? ibeg + a.Q : a.nitems;
    int item = ibeg + jwg;
    if (item >= iend) {
        if (a.ymax) dasr_amax_commit_idle(a.ymax, dasr_flat_wg(), dasr_flat_nwg());
        return;
    }
    const int NC = a.Cin >> 4;
    const int pixb = a.Cin * 4, rowb = a.W * pixb;
    const size_t sampb = (size_t)a.H * rowb;
    const char* const zp = (const char*)sp_zero_page;

    for (int i = tid; i < a.Cout; i += NTHR) sBias[i] = a.bias ? a.bias[i] : 0.f;
    sp_scales_gather<NP>(a.xmax, a.wmax, sRed);
    __syncthreads();
    const SpScale sc = sp_scales_take<NP>(sRed);

    int x0, y0, n0, bb;
    auto decode = [&](int it, int& ox0, int& oy0, int& on0, int& ob) {
        const int ns = it % a.nsl, pt = it / a.nsl;
        const int tile = pt % (a.tiles_x * a.tiles_y);
        ob = pt / (a.tiles_x * a.tiles_y);
        ox0 = (tile % a.tiles_x) * 32;
        oy0 = (tile / a.tiles_x) * SP_TH;
        on0 = ns * NTILE;
    };
    decode(item, x0, y0, n0, bb);
    int hoff[NHP];
    unsigned hok = 0;
    const char* hxb;
    auto halo_setup = [&](int fx0, int fy0, int fb, bool real) {
        hxb = (const char*)a.x + (size_t)fb * sampb;
        const int org = (fy0 - 1) * rowb + (fx0 - 1) * pixb;
        int t = tid;
#ifndef DASR_HIPEMU
        asm volatile("" : "+v"(t));
#endif
        hok = 0;
#pragma unroll
        for (int u = 0; u < NHP; ++u) {
            const int P = (t >> 2) + (NTHR / 4) * u;
            const int pr = P / SP_HW, pc = P - pr * SP_HW;
            hoff[u] = org + pr * rowb + pc * pixb + 16 * ((t & 3) ^ ((P >> 2) & 3));
            const bool ok = real && P < SP_HPIX && (unsigned)(fy0 - 1 + pr) < (unsigned)a.H && (unsigned)(fx0 - 1 + pc) < (unsigned)a.W;
            hok |= ok ? (1u << u) : 0u;
        }
    };
    auto halo_issue = [&](int u, int cc, int buf) {
        const char* src = ((hok >> u) & 1u) ? hxb + hoff[u] + 64 * cc : zp;
        DASR_GLDS16(src, ldsH + buf * SP_HBYTES + 1024 * NWV * u);
    };
    // slices of K-step st (taps TPS st .. TPS st + TPS - 1) of chunk cc, slice fn0: SLAB bytes each at
    // ((ns * 9 + tap) * NC + cc) * SLAB, into ring slot st % R
    auto w_issue = [&](int cc, int st, int fn0) {
#pragma unroll
        for (int j = 0; j < TPS; ++j) {
            const int tap = CW == 2 ? 3 * j + st : TPS * st + j;      // (CW = 2: step st = kernel column st, its three rows)
            const char* src = (const char*)a.ws + (size_t)(((fn0 / NTILE) * 9 + tap) * NC + cc) * SLAB + 16 * tid;
            const dasr_lds_addr_t dst = ldsW + ((st % R) * TPS + j) * SLAB;
#pragma unroll
            for (int q = 0; q < WQ; ++q) DASR_GLDS16(src + 16 * NTHR * q, dst + 1024 * NWV * q);
            if (WR > 0 && wv < WR) DASR_GLDS16(src + 16 * NTHR * WQ, dst + 1024 * NWV * WQ);
        }
    };
    // halo pieces of the NEXT chunk issued in step st: all of them at least D - 1 = one whole step before that chunk's first
    // step (whose wait lets the operations of the step before it stay in flight): one per step in steps 0 .. 4 of nine, three
    // and two in steps 0 and 1 of three
    constexpr int HC0 = (NHP + 1) / 2;
    auto halo_first = [](int st) { return TPS == 1 ? st : (st == 0 ? 0 : (st == 1 ? HC0 : NHP)); };
    auto halo_count = [](int st) { return TPS == 1 ? (st < NHP ? 1 : 0) : (st == 0 ? HC0 : (st == 1 ? NHP - HC0 : 0)); };

    const int Pl = MR * wr * SP_HW + li;
    const int boff = li * 32 + ((lh ^ ((li >> 3) & 1)) << 4) + wc * NW * 1024;

    int par = 0;
    float om = 0.f;                                     // running max |y| of this lane (a.ymax)
    halo_setup(x0, y0, bb, true);
#pragma unroll
    for (int u = 0; u < NHP; ++u) halo_issue(u, 0, 0);
#pragma unroll
    for (int t = 0; t < D; ++t) w_issue(0, t, n0);

    for (;;) {
        const int nitem = item + a.G8;
        const bool has_next = nitem < iend;
        int nx0 = x0, ny0 = y0, nn0 = n0, nb = bb;
        if (has_next) decode(nitem, nx0, ny0, nn0, nb);

        f32x16 acc[MR][NW];
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 bv = *(const float4*)(sBias + n0 + 32 * n + 8 * g + 4 * lh);
                if (NP == 2) bv = make_float4(0.f, 0.f, 0.f, 0.f);         // (scaled sums: the bias joins in the epilogue)
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    acc[m][n][4 * g] = bv.x; acc[m][n][4 * g + 1] = bv.y;
                    acc[m][n][4 * g + 2] = bv.z; acc[m][n][4 * g + 3] = bv.w;
                }
            }

        for (int cc = 0; cc < NC; ++cc) {
            const bool last = cc == NC - 1;
            if (last) halo_setup(nx0, ny0, nb, has_next);
            const int fcc = last ? 0 : cc + 1;
            const char* const hb = sH + par * SP_HBYTES;
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                // the slices of this step (issued D steps ago) and this chunk's halo pieces have landed once all but the
                // operations this wave issued in the last D - 1 steps (their halo pieces, TPS nwq slice pieces per step) are done
                {
                    int nh = 0;
#pragma unroll
                    for (int k = 1; k < D; ++k) nh += halo_count((st - k + NS) % NS);
                    sp_wait_vm_n(nh + (D - 1) * TPS * nwq);
                }
                if (NP == 2 && st == 0) {              // this chunk's halo pieces - this thread's own - have landed: split them
                    sp_presplit<NWV>(sH + par * SP_HBYTES, sc.sx, tid);
                    DASR_LDS_BARRIER();
                } else {
                    DASR_RAW_BARRIER();
                }
#pragma unroll
                for (int u = 0; u < halo_count(st); ++u) halo_issue(halo_first(st) + u, fcc, par ^ 1);
                if (st + D < NS) w_issue(cc, st + D, n0);
                else             w_issue(fcc, st + D - NS, last ? nn0 : n0);
                if constexpr (CW == 2) {
                    int Pq = Pl;
#ifndef DASR_HIPEMU
                    asm volatile("" : "+v"(Pq));
#endif
                    typename F::type A6[MR + 2][NP];
#pragma unroll
                    for (int r = 0; r < MR + 2; ++r) {
                        const int P = Pq + r * SP_HW + st;
                        const int key = (P >> 2) & 3;
                        A6[r][0] = sp_frag(hb + P * 64, lh, key, 0);
                        A6[r][1] = sp_frag(hb + P * 64, lh, key, 1);
                    }
                    DASR_SETPRIO(1);
#pragma unroll
                    for (int tj = 0; tj < 3; ++tj) {
                        const char* const wb = sW + ((st % R) * TPS + tj) * SLAB + boff;
#pragma unroll
                        for (int n = 0; n < NW; ++n) {
                            typename F::type Bw[NP];
#pragma unroll
                            for (int j = 0; j < NP; ++j) Bw[j] = *(const typename F::type*)(wb + j * PIECE + n * 1024);
#pragma unroll
                            for (int m = 0; m < MR; ++m) acc[m][n] = F::mma(Bw, A6[m + tj], acc[m][n]);
                        }
                    }
                    DASR_SETPRIO(0);
                } else {
#pragma unroll
                for (int tj = 0; tj < TPS; ++tj) {
                const int tap = TPS * st + tj;
                const int dy = tap / 3, dx = tap - 3 * dy;
                const char* const wb = sW + ((st % R) * TPS + tj) * SLAB + boff;
                int Pq = Pl;
#ifndef DASR_HIPEMU
                asm volatile("" : "+v"(Pq));
#endif
                typename F::type A[RW][NP];
#pragma unroll
                for (int m = 0; m < RW; ++m) {
                    const int P = Pq + (m + dy) * SP_HW + dx;
                    const int key = (P >> 2) & 3;
                    if constexpr (NP == 2) {            // pre-split pieces (sp_presplit)
                        A[m][0] = sp_frag(hb + P * 64, lh, key, 0);
                        A[m][1] = sp_frag(hb + P * 64, lh, key, 1);
                    } else {
                        const float4 lo = *(const float4*)(hb + P * 64 + (((2 * lh) ^ key) << 4));
                        const float4 hi = *(const float4*)(hb + P * 64 + (((2 * lh + 1) ^ key) << 4));
                        const float xv[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        F::split(xv, sc.sx, A[m]);
                    }
                }
                DASR_SETPRIO(1);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    typename F::type Bw[NP];
#pragma unroll
                    for (int j = 0; j < NP; ++j) Bw[j] = *(const typename F::type*)(wb + j * PIECE + n * 1024);
#pragma unroll
                    for (int m = 0; m < RW; ++m) acc[m][n] = F::mma(Bw, A[m], acc[m][n]);     // smallest terms first
                }
                DASR_SETPRIO(0);
                }
                }
            }
            par ^= 1;
        }

        // ---- epilogue: 32 channels per pass through [pixel][32 + 4] fp32, 32 bytes per lane out; residual, activation,
        // accumulate and the PixelShuffle(2) store as in conv_mfma.hip's epilogue
        DASR_RAW_BARRIER();
        char* const scr = sH + (par ^ 1) * SP_HBYTES + wv * (32 * SP_EPITCH);
        sp_epilogue<NW, NP, MR>(acc, a, scr, sBias, sc.inv, x0, y0, n0 + 32 * NW * wc, bb, MR * wr, lane, li, lh, om);
        if (!has_next) break;
        item = nitem; x0 = nx0; y0 = ny0; n0 = nn0; bb = nb;
    }
    sp_wait_vm<0>();
    if (a.ymax) dasr_amax_commit(a.ymax, om, sRed, dasr_flat_wg(), dasr_flat_nwg());     // (at most 256 workgroups)
}

// ---- 32 produced channels (the HR tail: 32 -> 32, 64 -> 32, and the dgrads of 32 -> 32 / 32 -> 128).  With one 32-channel tile a
// wave has only 12 MFMAs per (tap, 16-channel) step, and a barrier per step costs as much as the step (681 us average
// against ~420 us of matrix time on the x8 bench's fourteen launches).  The kernel slices of a chunk are tiny here (9 taps x
// 3 KB), so ALL NINE are fetched with the halo chunk, double-buffered by chunk: one wait + barrier per CHUNK (108 MFMAs per
// wave), the nine taps run free of synchronisation and the compiler pipelines reads, splits and MFMAs across them.
// (NT = 2, fp16 x 2 only, A/B behind impl + 512: the same one-barrier-per-chunk form for 64 produced channels - nine 4 KB slices.)
template <int NP, int NT = 1>
__global__ void __launch_bounds__(SP_NTHR, 2) k_conv3x3_split_n32(ConvSplitArgs a) {
    DASR_DYN_SMEM(smem);
    typedef SpFrag<NP> F;
    constexpr int NTILE = 32 * NT, PIECE = NTILE * 32, SLAB = NP * PIECE;     // NT = 1: 3 KB (NP = 2: 2 KB) per (tap, chunk)
    constexpr int WCH = 9 * SLAB, WPC = WCH / 16;                              // a chunk's nine slices: 27 KB = 1728 DMA pieces (18 KB = 1152)
    constexpr int WPT = SLAB / 16;                                             // pieces per tap: 192 (128)
    constexpr int WU = (WPC + SP_NTHR - 1) / SP_NTHR, WLAST = (WPC - (WU - 1) * SP_NTHR) / 64;   // trips; waves in the last one
    char* const sH = smem;                              // [2][SP_HBYTES]
    char* const sW = smem + 2 * SP_HBYTES;              // [2][WCH]
    float* const sBias = (float*)(sW + 2 * WCH);        // [Cout]
    float* const sRed = sBias + a.Cout;                 // [32]
    const int tid = threadIdx.x, lane = tid & 63, wv = DASR_UNIFORM((int)(tid >> 6));
    const int li = lane & 31, lh = lane >> 5;
    const dasr_lds_addr_t ldsH = DASR_LDS_ADDR(sH) + 1024 * wv, ldsW = DASR_LDS_ADDR(sW) + 1024 * wv;

    const int xcd = blockIdx.x & 7, jwg = blockIdx.x >> 3;
    const int ibeg = xcd * a.Q;
    const int iend = ibeg + a.Q < a.nitems ? ibeg + a.Q : a.nitems;
    int item = ibeg + jwg;
    if (item >= iend) {
        if (a.ymax) dasr_amax_commit_idle(a.ymax, dasr_flat_wg(), dasr_flat_nwg());
        return;
    }
    const int NC = a.Cin >> 4;
    const int pixb = a.Cin * 4, rowb = a.W * pixb;
    const size_t sampb = (size_t)a.H * rowb;
    const char* const zp = (const char*)sp_zero_page;

    for (int i = tid; i < a.Cout; i += SP_NTHR) sBias[i] = a.bias ? a.bias[i] : 0.f;
    sp_scales_gather<NP>(a.xmax, a.wmax, sRed);
    __syncthreads();
    const SpScale sc = sp_scales_take<NP>(sRed);

    int x0, y0, n0, bb;
    auto decode = [&](int it, int& ox0, int& oy0, int& on0, int& ob) {
        const int ns = it % a.nsl, pt = it / a.nsl;
        const int tile = pt % (a.tiles_x * a.tiles_y);
        ob = pt / (a.tiles_x * a.tiles_y);
        ox0 = (tile % a.tiles_x) * 32;
        oy0 = (tile / a.tiles_x) * SP_TH;
        on0 = ns * NTILE;
    };
    decode(item, x0, y0, n0, bb);
    int hoff[SP_NHP];
    unsigned hok = 0;
    const char* hxb;
    auto halo_setup = [&](int fx0, int fy0, int fb, bool real) {
        hxb = (const char*)a.x + (size_t)fb * sampb;
        const int org = (fy0 - 1) * rowb + (fx0 - 1) * pixb;
        int t = tid;
#ifndef DASR_HIPEMU
        asm volatile("" : "+v"(t));
#endif
        hok = 0;
#pragma unroll
        for (int u = 0; u < SP_NHP; ++u) {
            const int P = (t >> 2) + (SP_NTHR / 4) * u;
            const int pr = P / SP_HW, pc = P - pr * SP_HW;
            hoff[u] = org + pr * rowb + pc * pixb + 16 * ((t & 3) ^ ((P >> 2) & 3));
            const bool ok = real && P < SP_HPIX && (unsigned)(fy0 - 1 + pr) < (unsigned)a.H && (unsigned)(fx0 - 1 + pc) < (unsigned)a.W;
            hok |= ok ? (1u << u) : 0u;
        }
    };
    // everything chunk cc of (the item whose halo state is set up, slice fn0) needs: 5 halo pieces + the nine kernel slices
    // (pieces p = tid + 512 u < 1728: tap p / 192, 16-byte piece p % 192 of that tap's slice; waves 0-2 carry a fourth;
    // NP = 2: 1152 = 2 x 512 + 128 pieces, 128 per tap, waves 0-1 carry a third)
    auto chunk_issue = [&](int cc, int fn0, int buf) {
#pragma unroll
        for (int u = 0; u < SP_NHP; ++u) {
            const char* src = ((hok >> u) & 1u) ? hxb + hoff[u] + 64 * cc : zp;
            DASR_GLDS16(src, ldsH + buf * SP_HBYTES + 1024 * SP_NWV * u);
        }
        const char* wsrc = (const char*)a.ws + (size_t)(((fn0 / NTILE) * 9) * NC + cc) * SLAB;
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int p = tid + SP_NTHR * u;
            if (u == WU - 1 && wv >= WLAST) break;              // (wave-uniform) 1728 = 3 x 512 + 192, 1152 = 2 x 512 + 128
            const int tap = p / WPT, q = p - tap * WPT;
            DASR_GLDS16(wsrc + (size_t)tap * NC * SLAB + 16 * q, ldsW + buf * WCH + 1024 * SP_NWV * u);
        }
    };

    const int Pl = 2 * wv * SP_HW + li;
    const int boff = li * 32 + ((lh ^ ((li >> 3) & 1)) << 4);

    int par = 0;
    float om = 0.f;                                     // running max |y| of this lane (a.ymax)
    halo_setup(x0, y0, bb, true);
    chunk_issue(0, n0, 0);

    for (;;) {
        const int nitem = item + a.G8;
        const bool has_next = nitem < iend;
        int nx0 = x0, ny0 = y0, nn0 = n0, nb = bb;
        if (has_next) decode(nitem, nx0, ny0, nn0, nb);

        f32x16 acc[2][NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 bv = *(const float4*)(sBias + n0 + 32 * n + 8 * g + 4 * lh);
                if (NP == 2) bv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    acc[m][n][4 * g] = bv.x; acc[m][n][4 * g + 1] = bv.y;
                    acc[m][n][4 * g + 2] = bv.z; acc[m][n][4 * g + 3] = bv.w;
                }
            }

        for (int cc = 0; cc < NC; ++cc) {
            const bool last = cc == NC - 1;
            if (last) halo_setup(nx0, ny0, nb, has_next);
            // this chunk's halo and slices were issued a whole chunk ago (or in the prologue): everything this wave has in
            // flight is exactly that (plus the previous item's output stores)
            sp_wait_vm<0>();
            if (NP == 2) {
                sp_presplit(sH + par * SP_HBYTES, sc.sx, tid);
                DASR_LDS_BARRIER();
            } else {
                DASR_RAW_BARRIER();
            }
            chunk_issue(last ? 0 : cc + 1, last ? nn0 : n0, par ^ 1);
            const char* const hb = sH + par * SP_HBYTES;
            const char* const wc = sW + par * WCH + boff;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - 3 * dy;
                int Pq = Pl;
#ifndef DASR_HIPEMU
                asm volatile("" : "+v"(Pq));
#endif
                typename F::type A[2][NP];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int P = Pq + (m + dy) * SP_HW + dx;
                    const int key = (P >> 2) & 3;
                    if constexpr (NP == 2) {            // pre-split pieces (sp_presplit)
                        A[m][0] = sp_frag(hb + P * 64, lh, key, 0);
                        A[m][1] = sp_frag(hb + P * 64, lh, key, 1);
                    } else {
                        const float4 lo = *(const float4*)(hb + P * 64 + (((2 * lh) ^ key) << 4));
                        const float4 hi = *(const float4*)(hb + P * 64 + (((2 * lh + 1) ^ key) << 4));
                        const float xv[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        F::split(xv, sc.sx, A[m]);
                    }
                }
                const char* const wb = wc + tap * SLAB;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    typename F::type Bw[NP];
#pragma unroll
                    for (int j = 0; j < NP; ++j) Bw[j] = *(const typename F::type*)(wb + j * PIECE + n * 1024);
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][n] = F::mma(Bw, A[m], acc[m][n]);
                }
            }
            par ^= 1;
        }

        // ---- epilogue: 32 channels per pass through [pixel][32 + 4] fp32, 32 bytes per lane out; residual, activation,
        // accumulate and the PixelShuffle(2) store as in conv_mfma.hip's epilogue
        DASR_RAW_BARRIER();
        char* const scr = sH + (par ^ 1) * SP_HBYTES + wv * (32 * SP_EPITCH);
        sp_epilogue<NT, NP>(acc, a, scr, sBias, sc.inv, x0, y0, n0, bb, 2 * wv, lane, li, lh, om);
        if (!has_next) break;
        item = nitem; x0 = nx0; y0 = ny0; n0 = nn0; bb = nb;
    }
    sp_wait_vm<0>();
    if (a.ymax) dasr_amax_commit(a.ymax, om, sRed, dasr_flat_wg(), dasr_flat_nwg());     // (at most 256 workgroups)
}

// ---- the kernel split: fp32 packed [2][9][Cin][Cout] (plane 0 = HWIO) -> 16-bit image of both modes
//   ws[mode][slice][tap][chunk][piece j][row r][16],  element kk of row r at half (kk >> 3) ^ ((r >> 3) & 1)
//   mode 0 (forward): rows = output channels, K = input channels:  w[tap][16 chunk + kk][slice * NTILE + r]
//   mode 1 (dgrad):   rows = input channels,  K = output channels: w[8 - tap][slice * NTILE + r][16 chunk + kk]
// NP = 3: three bf16 pieces; NP = 2: two fp16 pieces of w 2^k, k from *wmax (the max |w| of plane 0, dasr_absmax)
__host__ __device__ static inline int sp_ntile(int N) { return (N % 128) == 0 ? 128 : ((N % 64) == 0 ? 64 : 32); }
template <int NP>
__device__ __forceinline__ void sp_split_weights(const float* __restrict__ w, unsigned short* __restrict__ ws, int Cin, int Cout,
                                                 const float* __restrict__ wmax, int slab_first, int slab_step) {
    // one workgroup per (mode, slice, tap, chunk) slab: the index arithmetic is per workgroup, a thread walks (row, kk) pairs
    // (the first form decoded every element with five integer divisions: 17.6 us for 128 x 128, fifty-five times a step)
    const size_t per_mode = (size_t)9 * NP * Cin * Cout;
    float sw = 1.f;
    if (NP == 2) {
        __shared__ float s_red[16];
        const float m0 = sp_amax_share(wmax);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m0;
        __syncthreads();
        float m = s_red[0];
        for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) m = fmaxf(m, s_red[wv]);
        sw = sp_pow2(sp_scale_exp(m));
    }
    const int nslab0 = (Cout / sp_ntile(Cout)) * 9 * (Cin / 16);          // slabs of mode 0
    const int nslab1 = (Cin / sp_ntile(Cin)) * 9 * (Cout / 16);
    for (int sl = slab_first; sl < nslab0 + nslab1; sl += slab_step) {
        const int mode = sl >= nslab0;
        const int q = sl - mode * nslab0;
        const int N = mode == 0 ? Cout : Cin, K = mode == 0 ? Cin : Cout;
        const int NTILE = sp_ntile(N), NCk = K / 16;
        const int cc = q % NCk, tap = (q / NCk) % 9, ns = q / (9 * NCk);
        unsigned short* const slab = ws + (size_t)mode * per_mode + (size_t)q * NP * NTILE * 16;
        for (int e = threadIdx.x; e < NTILE * 16; e += 256) {
            const int kk = e & 15, r = e >> 4;
            const int n = ns * NTILE + r, k = 16 * cc + kk;
            const float v = mode == 0 ? w[((size_t)tap * Cin + k) * Cout + n] : w[((size_t)(8 - tap) * Cin + n) * Cout + k];
            const int pos = r * 16 + (((kk >> 3) ^ ((r >> 3) & 1)) << 3) + (kk & 7);
            unsigned short* o = slab + pos;
            if (NP == 3) {
                const bf16_t h0 = dasr_f2bf(v);
                const float r1 = v - dasr_bf2f(h0);
                const bf16_t h1 = dasr_f2bf(r1);
                const bf16_t h2 = dasr_f2bf(r1 - dasr_bf2f(h1));
                memcpy(o, &h0, 2);
                memcpy(o + (size_t)NTILE * 16, &h1, 2);
                memcpy(o + (size_t)2 * NTILE * 16, &h2, 2);
            } else {
                const float vs = v * sw;
                const f16_t h0 = (f16_t)vs;
                const f16_t h1 = (f16_t)(vs - (float)h0);
                memcpy(o, &h0, 2);
                memcpy(o + (size_t)NTILE * 16, &h1, 2);
            }
        }
    }
}

template <int NP>
__global__ void __launch_bounds__(256) k_split_weights(const float* __restrict__ w, unsigned short* __restrict__ ws, int Cin, int Cout,
                                                       const float* __restrict__ wmax) {
    sp_split_weights<NP>(w, ws, Cin, Cout, wmax, blockIdx.x, gridDim.x);
}
// the fp16 x 2 images of every split convolution of a network in one launch: a job table in device memory (dasr.h:
// dasr_split_job), one workgroup per slab of every job
__global__ void __launch_bounds__(256) k_split_weights_multi(const dasr_split_job* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;                  // the last job whose wg_begin <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].wg_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const dasr_split_job J = jobs[lo];
    sp_split_weights<2>(J.w_packed, J.w_split, J.Cin, J.Cout, J.wmax, blockIdx.x - J.wg_begin, 1 << 30);
}

// ---- max |x| of a tensor as an amax buffer (dasr_common.h): one partial maximum per workgroup, no atomics, nothing to clear
__global__ void __launch_bounds__(256) k_absmax(const float* __restrict__ x, size_t n4, size_t n, float* __restrict__ out) {
    __shared__ float s_part[16];
    unsigned m = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    // four loads in flight per trip; indices past the end re-read the last piece (harmless for a maximum)
    if (n4 > 0)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t j = i + u * stride;
                v[u] = ((const float4*)x)[j < n4 ? j : n4 - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned a, b, c, d;
                memcpy(&a, &v[u].x, 4); memcpy(&b, &v[u].y, 4); memcpy(&c, &v[u].z, 4); memcpy(&d, &v[u].w, 4);
                a &= 0x7fffffffu; b &= 0x7fffffffu; c &= 0x7fffffffu; d &= 0x7fffffffu;
                a = a > b ? a : b; c = c > d ? c : d; a = a > c ? a : c;
                m = m > a ? m : a;
            }
        }
    if (blockIdx.x == 0)
        for (size_t i = 4 * n4 + threadIdx.x; i < n; i += 256) {
            unsigned a;
            memcpy(&a, x + i, 4);
            a &= 0x7fffffffu;
            m = m > a ? m : a;
        }
    // (magnitude bits order like the floats; NaN patterns sort above infinity and are kept as they are)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)m, o);
        m = m > t ? m : t;
    }
    if ((threadIdx.x & 63) == 0) memcpy(&s_part[threadIdx.x >> 6], &m, 4);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned r[4];
        memcpy(r, s_part, 16);
        unsigned a = r[0] > r[1] ? r[0] : r[1], b = r[2] > r[3] ? r[2] : r[3];
        a = a > b ? a : b;
        memcpy(&out[1 + blockIdx.x], &a, 4);
        if (blockIdx.x == 0) {
            const int np = (int)gridDim.x;
            memcpy(out, &np, 4);
        }
    }
}

// ------------------------------------------------------------------------------------------ C ABI
static bool sp_ok(int H, int W, int K, int N) {
    return H > 0 && W > 0 && (K % 16) == 0 && (N % 32) == 0 && N <= 1024 && (size_t)H * W * (K > N ? K : N) * 4 < ((size_t)1 << 31);
}
extern "C" int dasr_conv3x3_split_supported(int H, int W, int Cin, int Cout) {
    return (sp_ok(H, W, Cin, Cout) && sp_ok(H, W, Cout, Cin)) ? 1 : 0;      // forward and dgrad
}
// (the name is historical: the buffer is written, not raised - see dasr_common.h for its layout)
int absmax_raise(const float* x, size_t n, float* amax, void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(amax);
    DASR_CHECK_SHAPE(n > 0 && (((size_t)x) & 15) == 0);
    const size_t n4 = n / 4;
    size_t g = (n4 + 256 * 8 - 1) / (256 * 8);           // ~8 float4 per thread
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    DASR_LAUNCH(k_absmax, dim3((unsigned)g), dim3(256), 0, stream, x, n4, n, amax);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_absmax(const float* x, size_t n, float* amax, void* stream) { return absmax_raise(x, n, amax, stream); }
static unsigned sp_weight_slabs_all(int Cin, int Cout) {
    return (unsigned)((Cout / sp_ntile(Cout)) * 9 * (Cin / 16) + (Cin / sp_ntile(Cin)) * 9 * (Cout / 16));
}
static unsigned sp_weight_slabs(int Cin, int Cout) {        // (the single-tensor launch: workgroups walk the slabs)
    const unsigned n = sp_weight_slabs_all(Cin, Cout);
    return n > 2048 ? 2048 : n;
}
extern "C" size_t dasr_conv3x3_split_weights_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return sizeof(bf16_t) * (size_t)54 * Cin * Cout;
}
extern "C" size_t dasr_conv3x3_split2_weights_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return sizeof(f16_t) * (size_t)36 * Cin * Cout;
}
extern "C" int dasr_conv3x3_split_weights(const float* w_packed, unsigned short* w_split, int Cin, int Cout, void* stream) {
    DASR_CHECK_PTR(w_packed); DASR_CHECK_PTR(w_split);
    DASR_CHECK_SHAPE(Cin > 0 && Cout > 0 && (Cin % 32) == 0 && (Cout % 32) == 0);
    DASR_LAUNCH((k_split_weights<3>), dim3(sp_weight_slabs(Cin, Cout)), dim3(256), 0, stream, w_packed, w_split, Cin, Cout,
                (const float*)nullptr);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_conv3x3_split2_weights(const float* w_packed, const float* wmax, unsigned short* w_split, int Cin, int Cout,
                                           void* stream) {
    DASR_CHECK_PTR(w_packed); DASR_CHECK_PTR(w_split); DASR_CHECK_PTR(wmax);
    DASR_CHECK_SHAPE(Cin > 0 && Cout > 0 && (Cin % 32) == 0 && (Cout % 32) == 0);
    DASR_LAUNCH((k_split_weights<2>), dim3(sp_weight_slabs(Cin, Cout)), dim3(256), 0, stream, w_packed, w_split, Cin, Cout,
                wmax);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_conv3x3_split2_weights_slabs(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || (Cin % 32) != 0 || (Cout % 32) != 0) return 0;
    return (int)sp_weight_slabs_all(Cin, Cout);
}
extern "C" int dasr_conv3x3_split2_weights_multi(const dasr_split_job* jobs_host, const dasr_split_job* jobs_device, int njobs,
                                                 void* stream) {
    DASR_CHECK_PTR(jobs_host); DASR_CHECK_PTR(jobs_device);
    DASR_CHECK_SHAPE(njobs > 0);
    long long total = 0;
    for (int j = 0; j < njobs; ++j) {
        const dasr_split_job& J = jobs_host[j];
        DASR_CHECK_PTR(J.w_packed); DASR_CHECK_PTR(J.w_split); DASR_CHECK_PTR(J.wmax);
        DASR_CHECK_SHAPE(J.Cin > 0 && J.Cout > 0 && (J.Cin % 32) == 0 && (J.Cout % 32) == 0);
        DASR_CHECK_SHAPE(J.wg_begin == total);
        total += sp_weight_slabs_all(J.Cin, J.Cout);
        DASR_CHECK_SHAPE(total < (1ll << 30));
    }
    DASR_LAUNCH(k_split_weights_multi, dim3((unsigned)total), dim3(256), 0, stream, jobs_device, njobs);
    DASR_RETURN_LAUNCH_STATUS();
}
template <int NP>
static int sp_launch(ConvSplitArgs& a, void* stream) {
    const int NT = sp_ntile(a.Cout) / 32;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + SP_TH - 1) / SP_TH;
    a.nsl = a.Cout / (32 * NT);
    a.nitems = a.tiles_x * a.tiles_y * a.B * a.nsl;
    a.Q = (a.nitems + 7) / 8;
    a.G8 = a.Q < 32 ? a.Q : 32;
    if ((dasr_get_conv_bf16_impl() & 3) == 2) a.G8 = 1;        // tests: one workgroup per XCD walks every item of it
    const bool chunk2 = NT == 2 && NP == 2 && (dasr_get_conv_bf16_impl() & 512) != 0;     // (A/B: 64 channels on the chunk form)
    const size_t lds = 2 * (size_t)SP_HBYTES +
                       (NT == 1 || chunk2 ? 2 * (size_t)(9 * NP * 32 * NT * 32) : 3 * (size_t)((NP == 2 ? 3 : 1) * NP * 32 * NT * 32)) +
                       sizeof(float) * (size_t)(a.Cout + 32);
    if (lds > 160 * 1024) return DASR_E_UNSUPPORTED;
    const dim3 grid(8 * a.G8);
    // (impl + 256, A/B: the four-wave form - four tile rows per wave - of the fp16 x 2 kernel at 128 produced channels)
    if (NT == 4 && NP == 2 && (dasr_get_conv_bf16_impl() & 256) != 0)
        DASR_LAUNCH((k_conv3x3_split<4, 2, 4>), grid, dim3(256), lds, stream, a);
    // fp16 x 2, 128 produced channels: the channel-halves / kernel-column form (128 -> 128 forward 284 -> 279 us, 64 -> 256 286 ->
    // 278, 32 -> 128 at 512 x 640 1342 -> 1295; dgrads unchanged); at 64 produced channels it is no faster (64 -> 32 at 256 x 320:
    // 195 -> 204 us) and stays an A/B.  impl + 1024 swaps the two forms (tests, tools/bench_split.py).
    else if (NT == 4 && NP == 2 && (dasr_get_conv_bf16_impl() & 1024) == 0)
        DASR_LAUNCH((k_conv3x3_split<4, 2, 2, 2>), grid, dim3(SP_NTHR), lds, stream, a);
    else if (NT == 2 && NP == 2 && (dasr_get_conv_bf16_impl() & 1024) != 0)
        DASR_LAUNCH((k_conv3x3_split<2, 2, 2, 2>), grid, dim3(SP_NTHR), lds, stream, a);
    else if (NT == 4) DASR_LAUNCH((k_conv3x3_split<4, NP>), grid, dim3(SP_NTHR), lds, stream, a);
    else if (chunk2)  DASR_LAUNCH((k_conv3x3_split_n32<2, 2>), grid, dim3(SP_NTHR), lds, stream, a);
    else if (NT == 2) DASR_LAUNCH((k_conv3x3_split<2, NP>), grid, dim3(SP_NTHR), lds, stream, a);
    else              DASR_LAUNCH((k_conv3x3_split_n32<NP>), grid, dim3(SP_NTHR), lds, stream, a);
    DASR_RETURN_LAUNCH_STATUS();
}
static int sp_fwd(const float* x, const float* xmax, const unsigned short* w_split, const float* wmax, const float* bias,
                  const float* residual, float* y, float* ymax, int B, int H, int W, int Cin, int Cout, int act, int ps_r,
                  void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(w_split); DASR_CHECK_PTR(y);
    DASR_CHECK_SHAPE(B > 0);
    if (!dasr_conv3x3_split_supported(H, W, Cin, Cout)) return DASR_E_UNSUPPORTED;
    if (act < 0 || act > 2) return DASR_E_UNSUPPORTED;
    if (ps_r < 1) ps_r = 1;
    if (ps_r > 2 || (ps_r == 2 && (residual != nullptr || (Cout % 128) != 0))) return DASR_E_UNSUPPORTED;
    ConvSplitArgs a{x, (const bf16_t*)w_split, bias, residual, y, B, H, W, Cin, Cout, 0, act, ps_r, 0, 0, 0, 0, 0, 0, xmax, wmax, ymax};
    return xmax ? sp_launch<2>(a, stream) : sp_launch<3>(a, stream);
}
extern "C" int dasr_conv3x3_fwd_split(const float* x, const unsigned short* w_split, const float* bias,
                                      const float* residual, float* y, int B, int H, int W, int Cin, int Cout, int act,
                                      int ps_r, void* stream) {
    return sp_fwd(x, nullptr, w_split, nullptr, bias, residual, y, nullptr, B, H, W, Cin, Cout, act, ps_r, stream);
}
extern "C" int dasr_conv3x3_fwd_split2(const float* x, const float* xmax, const unsigned short* w_split, const float* wmax,
                                       const float* bias, const float* residual, float* y, float* y_amax, int B, int H,
                                       int W, int Cin, int Cout, int act, int ps_r, void* stream) {
    DASR_CHECK_PTR(xmax); DASR_CHECK_PTR(wmax);
    return sp_fwd(x, xmax, w_split, wmax, bias, residual, y, y_amax, B, H, W, Cin, Cout, act, ps_r, stream);
}
// dx[p, ci] (+)= sum_{tap, co} dconv[p - off(tap), co] * w[tap][ci][co]: mode 1 of the split image (taps already flipped)
static int sp_dgrad(const float* dconv, const float* dmax, const unsigned short* w_split, const float* wmax, float* dx,
                    int accumulate, int B, int H, int W, int Cin, int Cout, void* stream) {
    DASR_CHECK_PTR(dconv); DASR_CHECK_PTR(w_split); DASR_CHECK_PTR(dx);
    DASR_CHECK_SHAPE(B > 0);
    if (!dasr_conv3x3_split_supported(H, W, Cin, Cout)) return DASR_E_UNSUPPORTED;
    const size_t mode1 = (size_t)9 * (dmax ? 2 : 3) * Cin * Cout;
    ConvSplitArgs a{dconv, (const bf16_t*)w_split + mode1, nullptr, nullptr, dx, B, H, W, Cout, Cin,
                    accumulate, DASR_ACT_NONE, 1, 0, 0, 0, 0, 0, 0, dmax, wmax, nullptr};
    return dmax ? sp_launch<2>(a, stream) : sp_launch<3>(a, stream);
}
extern "C" int dasr_conv3x3_dgrad_split(const float* dconv, const unsigned short* w_split, float* dx, int accumulate, int B,
                                        int H, int W, int Cin, int Cout, void* stream) {
    return sp_dgrad(dconv, nullptr, w_split, nullptr, dx, accumulate, B, H, W, Cin, Cout, stream);
}
extern "C" int dasr_conv3x3_dgrad_split2(const float* dconv, const float* dmax, const unsigned short* w_split, const float* wmax,
                                         float* dx, int accumulate, int B, int H, int W, int Cin, int Cout, void* stream) {
    DASR_CHECK_PTR(dmax); DASR_CHECK_PTR(wmax);
    return sp_dgrad(dconv, dmax, w_split, wmax, dx, accumulate, B, H, W, Cin, Cout, stream);
}

// ------------------------------------------------------------------------------------------ weight gradient
// dW[tap][ci][co] = sum_p x[p + off(tap), ci] * dconv[p, co] with the same three-piece operands: M = ci, N = co, K = pixels,
// 16 per MFMA.  Structure of k_conv3x3_wgrad_mfma<2, 2> (conv_mfma.hip): a workgroup of four waves owns a 64 x 64 block of
// (ci, co) for all nine taps (nine resident 32 x 32 accumulators per wave), walks a strip of 2 x 32 pixel tiles staged in
// LDS as fp32 ([pixel][channel], 51 KB, two workgroups per CU), writes its partial dW as a slab; k_wgrad_reduce sums the slabs.
// What changes is the K loop: a K-step is 16 consecutive pixels of a tile row; a lane reads ITS channel of 8 of them for the
// dconv operand and of 10 of them per kernel row for the three taps of that row (their 8-pixel windows start one pixel
// apart), 38 conflict-free ds_read_b32, splits them in registers and issues 9 taps x 6 products = 54 bf16 MFMAs - against
// 8 x 9 = 72 exact-fp32 MFMAs of twice the duration each for the same 16 pixels.
#define SW_TW 32
// A workgroup owns a (32 MT) x (32 NTW) block of (ci, co).  With four (ci-tile, co-tile) pairs every wave takes one; blocks with
// fewer pairs (32-channel layers of the HR tail) give each pair G = 4 / pairs waves, which split the K-STEPS of a tile (every
// wave keeps all nine taps: the three taps of a kernel row share one 10-pixel window) and add their accumulators through
// LDS at the end.  Tile rows by pairs, so that a tile always holds the same amount of matrix work and ~50-76 KB of LDS.
__host__ __device__ constexpr int sw_th(int MT, int NTW) { return MT * NTW == 4 ? 2 : (MT * NTW == 2 ? 4 : 8); }

struct SplitWgradArgs {
    const float* xmax;   // NP = 2: max |x|, max |dy| in device memory (dasr_absmax)
    const float* dmax;
    const float* x;      // [B,H,W,Cin]
    const float* dy;     // [B,H,W,Cout]
    float* slabs;        // [P][9][Cin][Cout]
    float* bslabs;       // [P][Cout] or null
    int B, H, W, Cin, Cout;
    int P, ntiles;
};

// one value -> its three bf16 pieces, two values at a time (one v_cvt_pk_bf16_f32 per piece and pair)
__device__ __forceinline__ void sp_split_pair(float v0, float v1, bf16x2_t& h0, bf16x2_t& h1, bf16x2_t& h2) {
    h0 = dasr_f2bf2(v0, v1);
    const float r0 = v0 - dasr_bf2f(h0[0]), r1 = v1 - dasr_bf2f(h0[1]);
    h1 = dasr_f2bf2(r0, r1);
    h2 = dasr_f2bf2(r0 - dasr_bf2f(h1[0]), r1 - dasr_bf2f(h1[1]));
}

// (NP = 2) two values times the tensor's scale -> their two fp16 pieces
__device__ __forceinline__ void sp_split_pair2(float v0, float v1, float s, f16x2_t& h0, f16x2_t& h1) {
    v0 *= s; v1 *= s;
    h0[0] = (f16_t)v0; h0[1] = (f16_t)v1;
    h1[0] = (f16_t)(v0 - (float)h0[0]); h1[1] = (f16_t)(v1 - (float)h0[1]);
}

template <int MT, int NTW, int NP>
__global__ void __launch_bounds__(256, 2) k_conv3x3_wgrad_split(SplitWgradArgs a) {
    DASR_DYN_SMEM(smem);
    typedef SpFrag<NP> F;
    typedef typename F::elem E;
    constexpr int CIG = 32 * MT, COG = 32 * NTW, TH = sw_th(MT, NTW), HW = SW_TW + 2;
    constexpr int PAIRS = MT * NTW, G = 4 / PAIRS;
    float* sX = (float*)smem;                                   // [(TH+2)*HW][CIG]
    float* sD = sX + (TH + 2) * HW * CIG;                       // [TH*TW][COG]
    const int tid = threadIdx.x, lane = tid & 63, wv = DASR_UNIFORM((int)(tid >> 6));
    const int li = lane & 31, lh = lane >> 5;
    const int pair = wv % PAIRS, grp = wv / PAIRS;              // (wave-uniform: branches around MFMAs below)
    const int mt = pair / NTW, nt = pair % NTW;
    const int cgroups = a.Cout / COG;
    // (measured and dropped: an XCD-aware mapping that puts the gridDim.x (ci, co) blocks of one partial index - they read
    // the same pixel tiles - on ONE XCD's L2: no change at 128 -> 128, 6 % slower at 64 -> 64; the kernel is not waiting for HBM)
    const int bgrp = blockIdx.x, bpar = blockIdx.y;
    const int ci0 = (bgrp / cgroups) * CIG, co0 = (bgrp % cgroups) * COG;
    const int tiles_x = (a.W + SW_TW - 1) / SW_TW, tiles_y = (a.H + TH - 1) / TH;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bool do_bias = a.bslabs != nullptr && ci0 == 0 && tid < COG;
    float bsum = 0.f;
    float sx = 1.f, sd = 1.f, inv = 1.f;
    if (NP == 2) {
        __shared__ float s_red[32];
        sp_scales_gather<2>(a.xmax, a.dmax, s_red);
        __syncthreads();
        const int nw = (int)(blockDim.x >> 6);
        float mx = s_red[0], md = s_red[16];
        for (int w = 1; w < nw; ++w) { mx = fmaxf(mx, s_red[w]); md = fmaxf(md, s_red[16 + w]); }
        const int kx = sp_scale_exp(mx), kd = sp_scale_exp(md);
        sx = sp_pow2(kx); sd = sp_pow2(kd); inv = sp_pow2(-(kx + kd));
    }

    for (int tile = bpar; tile < a.ntiles; tile += a.P) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int x0 = tx * SW_TW, y0 = ty * TH;
        // stage the tile: every global load of a thread before the first LDS write (see k_conv3x3_wgrad_mfma)
        constexpr int NX = (TH + 2) * HW * (CIG / 4), NXI = (NX + 255) / 256;        // 64 x 64 block: 2176 float4 pieces, 9 per thread
        constexpr int ND = TH * SW_TW * (COG / 4), NDI = (ND + 255) / 256;           //                1024: 4 per thread
        // (144 accumulator registers are resident: the x pieces go in two batches, the second one after the barrier)
        constexpr int XA = NDI >= 8 ? 1 : (8 - NDI < NXI ? 8 - NDI : NXI);
        float4 vx[NXI - XA > XA ? NXI - XA : XA], vd[NDI];
        auto ldx = [&](int u) {
            const int idx = tid + 256 * u;
            const int c4 = idx % (CIG / 4), pix = idx / (CIG / 4);
            const int gy = y0 + pix / HW - 1, gx = x0 + pix % HW - 1;
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
            const int gy = y0 + pix / SW_TW, gx = x0 + pix % SW_TW;
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
#pragma unroll
        for (int u = XA; u < NXI; ++u) vx[u - XA] = ldx(u);
#pragma unroll
        for (int u = XA; u < NXI; ++u) stx(u, vx[u - XA]);
        __syncthreads();
        if (do_bias) {
#pragma unroll 8
            for (int px = 0; px < TH * SW_TW; ++px) bsum += sD[px * COG + tid];
        }
#pragma unroll 1
        for (int s = grp; s < TH * 2; s += G) {
            const int py = s >> 1, pc = 16 * (s & 1) + 8 * lh;       // this lane's first pixel column of the K-step
            // dconv operand: 8 pixels of channel co0 + 32 nt + li
            typename F::type Bd[NP];
            {
                const float* dp = sD + (py * SW_TW + pc) * COG + 32 * nt + li;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    if constexpr (NP == 3) {
                        bf16x2_t h0, h1, h2;
                        sp_split_pair(dp[e * COG], dp[(e + 1) * COG], h0, h1, h2);
                        Bd[0][e] = h0[0]; Bd[0][e + 1] = h0[1];
                        Bd[1][e] = h1[0]; Bd[1][e + 1] = h1[1];
                        Bd[2][e] = h2[0]; Bd[2][e + 1] = h2[1];
                    } else {
                        f16x2_t h0, h1;
                        sp_split_pair2(dp[e * COG], dp[(e + 1) * COG], sd, h0, h1);
                        Bd[0][e] = h0[0]; Bd[0][e + 1] = h0[1];
                        Bd[1][e] = h1[0]; Bd[1][e + 1] = h1[1];
                    }
                }
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                // x operand: halo columns pc .. pc + 9 of halo row py + dy, channel ci0 + 32 mt + li: the three taps of this
                // kernel row use columns dx .. dx + 7
                const float* xp = sX + ((py + dy) * HW + pc) * CIG + 32 * mt + li;
                E pc_[NP][10];
#pragma unroll
                for (int e = 0; e < 10; e += 2) {
                    if constexpr (NP == 3) {
                        bf16x2_t h0, h1, h2;
                        sp_split_pair(xp[e * CIG], xp[(e + 1) * CIG], h0, h1, h2);
                        pc_[0][e] = h0[0]; pc_[0][e + 1] = h0[1];
                        pc_[1][e] = h1[0]; pc_[1][e + 1] = h1[1];
                        pc_[2][e] = h2[0]; pc_[2][e + 1] = h2[1];
                    } else {
                        f16x2_t h0, h1;
                        sp_split_pair2(xp[e * CIG], xp[(e + 1) * CIG], sx, h0, h1);
                        pc_[0][e] = h0[0]; pc_[0][e + 1] = h0[1];
                        pc_[1][e] = h1[0]; pc_[1][e + 1] = h1[1];
                    }
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    typename F::type Ax[NP];
#pragma unroll
                    for (int j = 0; j < NP; ++j)
#pragma unroll
                        for (int e = 0; e < 8; ++e) Ax[j][e] = pc_[j][e + dx];
                    // (M = ci: the x fragment is the MFMA's first operand; the product set is symmetric in the two)
                    acc[3 * dy + dx] = F::mma(Ax, Bd, acc[3 * dy + dx]);
                }
            }
        }
    }
    if (G > 1) {
        // waves grp = 1 .. G-1 hand their accumulators to wave grp = 0 of the same pair, one tap at a time through
        // (G - 1) * PAIRS * 4 KB of the (now idle) staging LDS
        float* sR = (float*)smem;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            __syncthreads();
            if (grp > 0) {
                float* q = sR + ((grp - 1) * PAIRS + pair) * 1024 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) q[64 * r] = acc[t][r];
            }
            __syncthreads();
            if (grp == 0) {
                for (int gg = 0; gg < G - 1; ++gg) {
                    const float* q = sR + (gg * PAIRS + pair) * 1024 + lane;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += q[64 * r];
                }
            }
        }
    }
    if (do_bias) a.bslabs[(size_t)bpar * a.Cout + co0 + tid] = bsum;
    float* slab = a.slabs + (size_t)bpar * 9 * a.Cin * a.Cout;
    if (grp != 0) return;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
            slab[((size_t)tap * a.Cin + ci) * a.Cout + co0 + 32 * nt + li] = NP == 2 ? acc[tap][r] * inv : acc[tap][r];
        }
    }
}

// ---- the weight gradient of the fp16 x 2 scheme, operands split ONCE when the tile is staged (round 3, second version).
// In k_conv3x3_wgrad_split<.., 2> every K-step splits its 38 fp32 values in registers: ~160 VALU beside 27 MFMAs, and the two
// waves that share an x (or dconv) block each split it again - as long as the matrix work, 0.69 PF.  Here the staging pass
// (global -> registers -> LDS, which exists anyway) writes TWO fp16 planes [pixel][channel] (pixel stride 64 bytes mod 128),
// and the K loop is the one of k_conv3x3_wgrad_bf16 (conv_bf16_mfma.hip) on them: operands come out of LDS as MFMA
// fragments through ds_read_b64_tr_b16 (K = pixels: the transpose of the staged layout), the three taps of a kernel row
// share ONE 12-pixel read per plane (dx = 0: dwords 0..3, dx = 2: dwords 1..4, dx = 1: v_alignbit of neighbours): per K-step
// 22 transposed reads + 24 VALU + 27 MFMAs.  The bias gradient is summed from the staging registers (exact fp32 values).
__host__ __device__ constexpr int sw2_stride(int ch) { return ch == 32 ? 32 : ch + 32; }    // fp16 elements per pixel and plane
__device__ __forceinline__ s16x4 sp_tr16(const f16_t* p) {
    const bf16x4 r = lds_read_tr16((const bf16_t*)p);
    s16x4 v;
    __builtin_memcpy(&v, &r, 8);
    return v;
}
template <int MT, int NTW>
__global__ void __launch_bounds__(256, 2) k_conv3x3_wgrad_split2(SplitWgradArgs a) {
    DASR_DYN_SMEM(smem);
    constexpr int CIG = 32 * MT, COG = 32 * NTW, TH = sw_th(MT, NTW), HW = SW_TW + 2;
    constexpr int PAIRS = MT * NTW, G = 4 / PAIRS;
    constexpr int SXP = sw2_stride(CIG), SDP = sw2_stride(COG);
    constexpr int NXE = (TH + 2) * HW * SXP, NDE = TH * SW_TW * SDP;
    f16_t* sX0 = (f16_t*)smem;                                  // [(TH+2)*HW][SXP]: fp16(x s)
    f16_t* sX1 = sX0 + NXE;                                     //                    what that rounding left
    f16_t* sD0 = sX1 + NXE;                                     // [TH*TW][SDP]
    f16_t* sD1 = sD0 + NDE;
    const int tid = threadIdx.x, lane = tid & 63, wv = DASR_UNIFORM((int)(tid >> 6));
    const int li = lane & 31, lh = lane >> 5;
    const int pair = wv % PAIRS, grp = wv / PAIRS;              // (wave-uniform: branches around MFMAs below)
    const int mt = pair / NTW, nt = pair % NTW;
    const int cgroups = a.Cout / COG;
    const int ci0 = (blockIdx.x / cgroups) * CIG, co0 = (blockIdx.x % cgroups) * COG;
    const int tiles_x = (a.W + SW_TW - 1) / SW_TW, tiles_y = (a.H + TH - 1) / TH;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bool do_bias = a.bslabs != nullptr && ci0 == 0;       // (workgroup-uniform)
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);                // this thread's channel quad (tid % (COG / 4)) of sum dconv
    float sx, sd, inv;
    {
        __shared__ float s_red[32];
        sp_scales_gather<2>(a.xmax, a.dmax, s_red);
        __syncthreads();
        float mx = s_red[0], md = s_red[16];
        for (int w = 1; w < 4; ++w) { mx = fmaxf(mx, s_red[w]); md = fmaxf(md, s_red[16 + w]); }
        const int kx = sp_scale_exp(mx), kd = sp_scale_exp(md);
        sx = sp_pow2(kx); sd = sp_pow2(kd); inv = sp_pow2(-(kx + kd));
    }
    // transposed-read lane roles: lane 4q+p of its 16-lane group passes the address of pixel row q, channels 4p..4p+3 of the
    // group's 16 channels; groups 0,1 cover channels 0..15 / 16..31 of pixels 0..7 of the K-step, groups 2,3 pixels 8..15
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    const int xoff = 32 * mt + 16 * tg + 4 * tp, doff = 32 * nt + 16 * tg + 4 * tp;
    // four values times the tensor's scale -> 4 + 4 fp16 (one 8-byte LDS write per plane)
    auto put = [&](f16_t* p0, f16_t* p1, int off, float4 v, float s) {
        const float f[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
        f16_t h0[4], h1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h0[j] = (f16_t)f[j];
            h1[j] = (f16_t)(f[j] - (float)h0[j]);
        }
        __builtin_memcpy(p0 + off, h0, 8);
        __builtin_memcpy(p1 + off, h1, 8);
    };

    for (int tile = blockIdx.y; tile < a.ntiles; tile += a.P) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int x0 = tx * SW_TW, y0 = ty * TH;
        // stage the tile: every global load of a thread before the first LDS write (see k_conv3x3_wgrad_mfma)
        constexpr int NX = (TH + 2) * HW * (CIG / 4), NXI = (NX + 255) / 256;        // 64 x 64 block: 2176 float4 pieces, 9 per thread
        constexpr int ND = TH * SW_TW * (COG / 4), NDI = (ND + 255) / 256;           //                1024: 4 per thread
        // (144 accumulator registers are resident: the x pieces go in two batches, the second one after the barrier)
        constexpr int XA = NDI >= 8 ? 1 : (8 - NDI < NXI ? 8 - NDI : NXI);
        float4 vx[NXI - XA > XA ? NXI - XA : XA], vd[NDI];
        // (buffer loads: a 32-bit offset per piece, the zero padding from the descriptor's range check - SQ counters of the
        // first form, with 64-bit predicated loads: 6.9 VALU per MFMA, most of them this staging pass)
        const BufRsrc rx = dasr_make_rsrc(a.x + (size_t)b * a.H * a.W * a.Cin, (size_t)a.H * a.W * a.Cin * sizeof(float));
        const BufRsrc rd = dasr_make_rsrc(a.dy + (size_t)b * a.H * a.W * a.Cout, (size_t)a.H * a.W * a.Cout * sizeof(float));
        auto ldx = [&](int u) {
            const int idx = tid + 256 * u;
            const int c4 = idx % (CIG / 4), pix = idx / (CIG / 4);
            const int gy = y0 + pix / HW - 1, gx = x0 + pix % HW - 1;
            const bool ok = idx < NX && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const u32x4_t r = dasr_buffer_load16(rx, ok ? (unsigned)(((gy * a.W + gx) * a.Cin + ci0 + 4 * c4) * (int)sizeof(float)) : DASR_OOB);
            float4 v;
            __builtin_memcpy(&v, &r, 16);
            return v;
        };
        auto stx = [&](int u, float4 v) {
            const int idx = tid + 256 * u;
            if (idx < NX) put(sX0, sX1, (idx / (CIG / 4)) * SXP + 4 * (idx % (CIG / 4)), v, sx);
        };
#pragma unroll
        for (int u = 0; u < XA; ++u) vx[u] = ldx(u);
#pragma unroll
        for (int u = 0; u < NDI; ++u) {
            const int idx = tid + 256 * u;
            const int c4 = idx % (COG / 4), pix = idx / (COG / 4);
            const int gy = y0 + pix / SW_TW, gx = x0 + pix % SW_TW;
            const bool ok = idx < ND && gy < a.H && gx < a.W;
            const u32x4_t r = dasr_buffer_load16(rd, ok ? (unsigned)(((gy * a.W + gx) * a.Cout + co0 + 4 * c4) * (int)sizeof(float)) : DASR_OOB);
            __builtin_memcpy(&vd[u], &r, 16);
        }
        __syncthreads();                        // every wave is done with the previous tile
#pragma unroll
        for (int u = 0; u < XA; ++u) stx(u, vx[u]);
#pragma unroll
        for (int u = 0; u < NDI; ++u) {
            const int idx = tid + 256 * u;
            if (idx < ND) put(sD0, sD1, (idx / (COG / 4)) * SDP + 4 * (idx % (COG / 4)), vd[u], sd);
            if (do_bias) { bs.x += vd[u].x; bs.y += vd[u].y; bs.z += vd[u].z; bs.w += vd[u].w; }   // (zeros outside the image)
        }
#pragma unroll
        for (int u = XA; u < NXI; ++u) vx[u - XA] = ldx(u);
#pragma unroll
        for (int u = XA; u < NXI; ++u) stx(u, vx[u - XA]);
        __syncthreads();
        // K-step s = 16 consecutive pixels of one tile row.  EXEC is all ones here (the tile loop and the K-step split are
        // wave-uniform), as ds_read_b64_tr_b16 requires.
        constexpr int NS = TH * (SW_TW / 16);
#pragma unroll 1
        for (int s = grp; s < NS; s += G) {
            const int py = s / (SW_TW / 16), px0 = 16 * (s % (SW_TW / 16)) + 8 * lh + tq;
            f16x8 Bd[2];
            {
                const int o = (py * SW_TW + px0) * SDP + doff;
                const s16x4 l0 = sp_tr16(sD0 + o), u0 = sp_tr16(sD0 + o + 4 * SDP);
                const s16x4 l1 = sp_tr16(sD1 + o), u1 = sp_tr16(sD1 + o + 4 * SDP);
                __builtin_memcpy(&Bd[0], &l0, 8); __builtin_memcpy((char*)&Bd[0] + 8, &u0, 8);
                __builtin_memcpy(&Bd[1], &l1, 8); __builtin_memcpy((char*)&Bd[1] + 8, &u1, 8);
            }
            const int xo = (py * HW + px0) * SXP + xoff;
            // (pixels 10, 11 of the last window of a tile row belong to the next row / lie past the tile: read, unused)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                f16x8 A[3][2];                                  // [dx][plane]
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const f16_t* ap = (pl == 0 ? sX0 : sX1) + xo + dy * HW * SXP;
                    const s16x4 a0 = sp_tr16(ap), a1 = sp_tr16(ap + 4 * SXP), a2 = sp_tr16(ap + 8 * SXP);
                    unsigned R[6];
                    __builtin_memcpy(&R[0], &a0, 8);
                    __builtin_memcpy(&R[2], &a1, 8);
                    __builtin_memcpy(&R[4], &a2, 8);
                    const unsigned V0[4] = {R[0], R[1], R[2], R[3]}, V2[4] = {R[1], R[2], R[3], R[4]};
                    unsigned V1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) V1[e] = (R[e] >> 16) | (R[e + 1] << 16);
                    __builtin_memcpy(&A[0][pl], V0, 16);
                    __builtin_memcpy(&A[1][pl], V1, 16);
                    __builtin_memcpy(&A[2][pl], V2, 16);
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[3 * dy + dx] = SpFrag<2>::mma(A[dx], Bd, acc[3 * dy + dx]);
            }
        }
    }
    float* sR = (float*)smem;
    if (G > 1) {
        // waves grp = 1 .. G-1 hand their accumulators to wave grp = 0 of the same pair, one tap at a time through
        // (G - 1) * PAIRS * 4 KB of the (now idle) staging LDS
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            __syncthreads();
            if (grp > 0) {
                float* q = sR + ((grp - 1) * PAIRS + pair) * 1024 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) q[64 * r] = acc[t][r];
            }
            __syncthreads();
            if (grp == 0) {
                for (int gg = 0; gg < G - 1; ++gg) {
                    const float* q = sR + (gg * PAIRS + pair) * 1024 + lane;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += q[64 * r];
                }
            }
        }
    }
    if (do_bias) {                              // the threads' channel-quad sums meet through LDS: threads q, q + COG/4, ... own quad q
        __syncthreads();
        *(float4*)(sR + 4 * tid) = bs;
        __syncthreads();
        if (tid < COG) {
            float r = 0.f;
            for (int t = tid >> 2; t < 256; t += COG / 4) r += sR[4 * t + (tid & 3)];
            a.bslabs[(size_t)blockIdx.y * a.Cout + co0 + tid] = r;
        }
    }
    float* slab = a.slabs + (size_t)blockIdx.y * 9 * a.Cin * a.Cout;
    if (grp != 0) return;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
            slab[((size_t)tap * a.Cin + ci) * a.Cout + co0 + 32 * nt + li] = acc[tap][r] * inv;
        }
    }
}

static void sw_plan(int B, int H, int W, int Cin, int Cout, int& MT, int& NTW, int& groups, int& ntiles, int& P) {
    MT = (Cin % 64) == 0 ? 2 : 1;
    NTW = (Cout % 64) == 0 ? 2 : 1;
    if (MT == 1 && (Cout % 128) == 0) NTW = 4;      // 32 input channels x a multiple of 128 outputs: four pairs again
    groups = (Cin / (32 * MT)) * (Cout / (32 * NTW));
    const int th = sw_th(MT, NTW);
    ntiles = B * ((H + th - 1) / th) * ((W + SW_TW - 1) / SW_TW);
    P = 512 / groups;
    if (P < 1) P = 1;
    if (P > ntiles) P = ntiles;
}
extern "C" size_t dasr_conv3x3_wgrad_split_workspace(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin % 32) || (Cout % 32)) return 0;
    int MT, NTW, groups, ntiles, P;
    sw_plan(B, H, W, Cin, Cout, MT, NTW, groups, ntiles, P);
    return sizeof(float) * ((size_t)P * 9 * Cin * Cout + (size_t)P * Cout);
}
template <int NP>
static int sw_launch(const float* x, const float* xmax, const float* dconv, const float* dmax, float* dw, float* dbias,
                     void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(dconv); DASR_CHECK_PTR(dw); DASR_CHECK_PTR(workspace);
    DASR_CHECK_SHAPE(B > 0 && H > 0 && W > 0);
    if ((Cin % 32) != 0 || (Cout % 32) != 0) return DASR_E_UNSUPPORTED;
    if (workspace_bytes < dasr_conv3x3_wgrad_split_workspace(B, H, W, Cin, Cout)) return DASR_E_WORKSPACE;
    int MT, NTW, groups, ntiles, P;
    sw_plan(B, H, W, Cin, Cout, MT, NTW, groups, ntiles, P);
    const size_t nW = (size_t)9 * Cin * Cout;
    float* slabs = (float*)workspace;
    float* bslabs = dbias ? slabs + (size_t)P * nW : nullptr;
    SplitWgradArgs a{xmax, dmax, x, dconv, slabs, bslabs, B, H, W, Cin, Cout, P, ntiles};
    const int th = sw_th(MT, NTW);
    const dim3 grid(groups, P);
    // fp16 x 2: the staged-split kernel (128 -> 128 424 -> 317 us, 64 -> 64 134 -> 107, 32 -> 128 at 512 x 640 1716 -> 1349, 32 -> 32
    // there 544 -> 435, 64 -> 32 277 -> 265; before its staging went through buffer descriptors the small blocks lost with it).
    // impl + 64 (A/B, tests): the first version - fp32 tiles, operands split per K-step.
    const int impl = dasr_get_conv_bf16_impl();
    const bool fits32 = (size_t)H * W * (Cin > Cout ? Cin : Cout) * sizeof(float) < ((size_t)1 << 31);    // (its 32-bit buffer offsets)
    if (NP == 2 && (impl & 64) == 0 && fits32) {
        size_t lds2 = sizeof(f16_t) * 2 * (size_t)((th + 2) * (SW_TW + 2) * sw2_stride(32 * MT) + th * SW_TW * sw2_stride(32 * NTW));
        if (lds2 < 4 * 4096) lds2 = 4 * 4096;                     // (the bias-gradient exchange: 256 float4)
        if (MT == 2 && NTW == 2)      DASR_LAUNCH((k_conv3x3_wgrad_split2<2, 2>), grid, dim3(256), lds2, stream, a);
        else if (MT == 1 && NTW == 4) DASR_LAUNCH((k_conv3x3_wgrad_split2<1, 4>), grid, dim3(256), lds2, stream, a);
        else if (MT == 2 && NTW == 1) DASR_LAUNCH((k_conv3x3_wgrad_split2<2, 1>), grid, dim3(256), lds2, stream, a);
        else if (MT == 1 && NTW == 2) DASR_LAUNCH((k_conv3x3_wgrad_split2<1, 2>), grid, dim3(256), lds2, stream, a);
        else                          DASR_LAUNCH((k_conv3x3_wgrad_split2<1, 1>), grid, dim3(256), lds2, stream, a);
        return wgrad_reduce_launch(slabs, dw, nW, P, stream, bslabs, dbias, Cout);
    }
    const size_t lds = sizeof(float) * (size_t)((th + 2) * (SW_TW + 2) * 32 * MT + th * SW_TW * 32 * NTW);
    if (MT == 2 && NTW == 2)      DASR_LAUNCH((k_conv3x3_wgrad_split<2, 2, NP>), grid, dim3(256), lds, stream, a);
    else if (MT == 1 && NTW == 4) DASR_LAUNCH((k_conv3x3_wgrad_split<1, 4, NP>), grid, dim3(256), lds, stream, a);
    else if (MT == 2 && NTW == 1) DASR_LAUNCH((k_conv3x3_wgrad_split<2, 1, NP>), grid, dim3(256), lds, stream, a);
    else if (MT == 1 && NTW == 2) DASR_LAUNCH((k_conv3x3_wgrad_split<1, 2, NP>), grid, dim3(256), lds, stream, a);
    else                          DASR_LAUNCH((k_conv3x3_wgrad_split<1, 1, NP>), grid, dim3(256), lds, stream, a);
    return wgrad_reduce_launch(slabs, dw, nW, P, stream, bslabs, dbias, Cout);
}
extern "C" int dasr_conv3x3_wgrad_split(const float* x, const float* dconv, float* dw, float* dbias, void* workspace,
                                        size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, void* stream) {
    return sw_launch<3>(x, nullptr, dconv, nullptr, dw, dbias, workspace, workspace_bytes, B, H, W, Cin, Cout, stream);
}
extern "C" int dasr_conv3x3_wgrad_split2(const float* x, const float* xmax, const float* dconv, const float* dmax, float* dw,
                                         float* dbias, void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin,
                                         int Cout, void* stream) {
    DASR_CHECK_PTR(xmax); DASR_CHECK_PTR(dmax);
    return sw_launch<2>(x, xmax, dconv, dmax, dw, dbias, workspace, workspace_bytes, B, H, W, Cin, Cout, stream);
}
