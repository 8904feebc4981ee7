// conv_c1.hip — 3x3 convolution of a ONE-channel image (the depth map) to Cout channels with fused ReLU:
// SEAN.mlp_mask (normalization.py:36-40,61), forward and weight/bias gradient - and, as the CIN = 3 instantiation, of
// the 3-channel LR frame: the encoder's first layer (sftmd_arch.py:745, weight-normed 3 -> 32, LeakyReLU), which ran on
// the generic direct kernels (1.7 ms forward, 6.3 ms weight gradient at 32 frames of 256x320).  There is next to no reduction over
// input channels, so this is HBM-bound elementwise work: the forward writes B*H*W*Cout floats, the backward
// reads the incoming gradient and the saved activation once (ReLU backward fused) and reduces over pixels.
// Lane layout: a lane owns 4 consecutive output channels (float4), 256-byte runs per 16 lanes.
#include "dasr_common.h"
#include "bf16.h"
#include "conv_kernels.h"

// Both kernels walk image ROWS: a workgroup stages the three depth-map rows around row (b, y) in LDS (zero padded,
// W + 2 floats each) and its threads - channel quad q = tid % nq, pixel lane pl = tid / nq - step through the row's
// pixels with plain 32-bit increments.  (The first version derived (b, y, x) of every pixel from a flat 64-bit index -
// three 64-bit divisions per pixel - and fetched the 9 taps with predicated global loads: VALU-bound at 1.7 TB/s.)
#define C1_MAXW 4096
// three image rows around row y, [3][(W+2)*CIN] (NHWC input: a row is W*CIN consecutive floats), zero padded
// Four loads per thread are issued before the first LDS write, from clamped addresses with the padding selected in
// afterwards: the plain `lds[i] = inside ? x[..] : 0` loop is one exec-masked global round trip per element per thread.
template <int CIN>
__device__ __forceinline__ void c1_stage_rows(const float* __restrict__ x, float* sRow, int b, int y, int H, int W) {
    const float* xb = x + (size_t)b * H * W * CIN;
    const int RW = (W + 2) * CIN, n = 3 * RW;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * 256) {
        float v[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u < n ? i0 + 256 * u : 0;
            const int r = i / RW, c = i - r * RW;
            const int iy = y + r - 1, ix = c / CIN - 1;
            in[u] = iy >= 0 && iy < H && ix >= 0 && ix < W;
            v[u] = xb[in[u] ? ((size_t)iy * W + ix) * CIN + c % CIN : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + 256 * u < n) sRow[i0 + 256 * u] = in[u] ? v[u] : 0.f;
    }
}

// T = storage type of the Cout-channel activation (y, dy, yact): float or bf16_t; the depth map and the kernel are fp32
template <typename T, int CIN>
__global__ void __launch_bounds__(256) k_conv3x3_c1_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, T* __restrict__ y, int B,
                                                        int H, int W, int Cout, int act, float* __restrict__ amax) {
    DASR_DYN_SMEM(smem);
    float* sRow = (float*)smem;                    // [3][(W+2)*CIN]
    const int nq = Cout / 4;                       // channel quads
    const int q = threadIdx.x % nq, pl = threadIdx.x / nq, npl = 256 / nq;
    constexpr int NT = 9 * CIN;                    // (tap, ci) terms: HWIO kernel rows
    const int RW = (W + 2) * CIN;
    float4 wt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wt[t] = *(const float4*)(w + (size_t)t * Cout + 4 * q);
    const float4 bv = bias ? *(const float4*)(bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    // ReLU: 0 (not 0 * v: keeps +0 and does not pass NaN through the multiply), LeakyReLU: 0.2 v, none: v
    const float slope = act == DASR_ACT_LRELU02 ? 0.2f : 1.f;
    const bool relu = act == DASR_ACT_RELU;
    auto neg = [&](float v) { return relu ? 0.f : slope * v; };
    float om = 0.f;                                // running max |y| of this lane (amax != null)
    for (int row = blockIdx.x; row < B * H; row += gridDim.x) {
        const int b = row / H, py = row - b * H;
        __syncthreads();
        c1_stage_rows<CIN>(x, sRow, b, py, H, W);
        __syncthreads();
        T* yrow = y + ((size_t)row * W) * Cout + 4 * q;
        for (int px = pl; px < W; px += npl) {
            float4 acc = bv;
#pragma unroll
            for (int t = 0; t < NT; ++t) {     // t = (kh*3 + kw)*CIN + ci
                const float d = sRow[(t / (3 * CIN)) * RW + (px + (t / CIN) % 3) * CIN + t % CIN];
                acc.x = fmaf(d, wt[t].x, acc.x); acc.y = fmaf(d, wt[t].y, acc.y);
                acc.z = fmaf(d, wt[t].z, acc.z); acc.w = fmaf(d, wt[t].w, acc.w);
            }
            // branch-free: the activation kind is one uniform slope (ReLU 0, LeakyReLU 0.2, none 1)
            acc.x = acc.x > 0.f ? acc.x : neg(acc.x); acc.y = acc.y > 0.f ? acc.y : neg(acc.y);
            acc.z = acc.z > 0.f ? acc.z : neg(acc.z); acc.w = acc.w > 0.f ? acc.w : neg(acc.w);
            st4(yrow + (size_t)px * Cout, acc);
            om = dasr_amax4(om, acc);
        }
    }
    if (amax) dasr_amax_commit(amax, om, sRow, dasr_flat_wg(), dasr_flat_nwg());     // (grid <= 2048 workgroups)
}

// dw[tap][co] = sum_p x[p+tap] * dconv[p][co], dbias[co] = sum_p dconv[p][co], dconv = dy * act'(y)
//
// HASY: the saved activation is read and its derivative applied (a template parameter, and the derivative a select
// against one uniform slope: with `yact ? load : 1` and the activation kind tested per element the compiler put every y
// load in its own exec-masked block followed by s_waitcnt vmcnt(0) - eight serial round trips per trip - and several
// scalar branches per element: 559 us for 32 frames of 256x320 in bf16, 2.4 TB/s).  Loads are unconditional (clamped
// pixel, contribution zeroed) and software-pipelined: the next trip's 2*NPX loads - of the next ROW when the row is
// done, the addresses do not depend on the staged depth rows - are in flight while this trip is reduced.
template <typename T, int CIN, bool HASY>
__global__ void __launch_bounds__(256) k_conv3x3_c1_wgrad(const float* __restrict__ x, const T* __restrict__ dy,
                                                          const T* __restrict__ yact, float* __restrict__ dw,
                                                          float* __restrict__ dbias, int B, int H, int W, int Cout,
                                                          int act) {
    DASR_DYN_SMEM(smem);
    constexpr int NT = 9 * CIN;                    // (tap, ci) terms; accumulator NT is the bias gradient
    const int RW = (W + 2) * CIN;
    float* red = (float*)smem;                     // [256][4] partials of ONE term at a time
    float* sRow = red + 256 * 4;                   // [3][(W+2)*CIN]
    const int nq = Cout / 4;
    const int q = threadIdx.x % nq, pl = threadIdx.x / nq, npl = 256 / nq;
    const float slope = act == DASR_ACT_RELU ? 0.f : act == DASR_ACT_LRELU02 ? 0.2f : 1.f;
    float4 acc[NT + 1];
#pragma unroll
    for (int t = 0; t < NT + 1; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    // NPX pixels per trip; bf16 loads carry half the bytes, so twice the pixels keep the same bytes in flight
    constexpr int NPX = sizeof(T) == 2 ? 8 : 4;
    typedef typename raw4<T>::type Raw;
    Raw gq[NPX], yq[NPX];
    const int nrows = B * H, step = NPX * npl;
    auto issue = [&](int row, int base) {
        const size_t ro = ((size_t)row * W) * Cout + 4 * q;
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
            const int px = base + pl + u * npl < W ? base + pl + u * npl : W - 1;
            gq[u] = ld4_raw(dy + ro + (size_t)px * Cout);
            if (HASY) yq[u] = ld4_raw(yact + ro + (size_t)px * Cout);
        }
    };
    if ((int)blockIdx.x < nrows) issue(blockIdx.x, 0);
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int b = row / H, py = row - b * H;
        __syncthreads();
        c1_stage_rows<CIN>(x, sRow, b, py, H, W);
        __syncthreads();
        for (int base = 0; base < W; base += step) {
            float4 g[NPX];
#pragma unroll
            for (int u = 0; u < NPX; ++u) {
                g[u] = cvt4<T>(gq[u]);
                if (HASY) {
                    const float4 yv = cvt4<T>(yq[u]);
                    g[u].x *= yv.x > 0.f ? 1.f : slope; g[u].y *= yv.y > 0.f ? 1.f : slope;
                    g[u].z *= yv.z > 0.f ? 1.f : slope; g[u].w *= yv.w > 0.f ? 1.f : slope;
                }
            }
            // the registers are free again: next trip (same row, or the first of this workgroup's next row).  Issued
            // unconditionally - after the very last trip it re-reads a trip of this row - because a branch around the loads
            // makes the waitcnt pass assume the fewest outstanding loads at the join and wait for the NEW loads too.
            {
                int nb = base + step, nr = row;
                if (nb >= W) {
                    nb = 0;
                    nr = row + (int)gridDim.x < nrows ? row + (int)gridDim.x : row;
                }
                DASR_SCHED_BARRIER();      // conversions above, loads here, the reduction below: otherwise the scheduler
                issue(nr, nb);             // sinks the loads to the end of the trip and nothing is in flight during it
                DASR_SCHED_BARRIER();
            }
#pragma unroll
            for (int u = 0; u < NPX; ++u) {
                const int px = base + pl + u * npl;
                if (px >= W) g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                const int pc = px < W ? px : W - 1;
                acc[NT].x += g[u].x; acc[NT].y += g[u].y; acc[NT].z += g[u].z; acc[NT].w += g[u].w;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float d = sRow[(t / (3 * CIN)) * RW + (pc + (t / CIN) % 3) * CIN + t % CIN];
                    acc[t].x = fmaf(d, g[u].x, acc[t].x); acc[t].y = fmaf(d, g[u].y, acc[t].y);
                    acc[t].z = fmaf(d, g[u].z, acc[t].z); acc[t].w = fmaf(d, g[u].w, acc[t].w);
                }
            }
        }
    }
    // reduce over the pixel lanes of the workgroup (one term at a time through 4 KB of LDS), then one float atomic per
    // output per workgroup
#pragma unroll
    for (int t = 0; t < NT + 1; ++t) {
        __syncthreads();
        *(float4*)(red + threadIdx.x * 4) = acc[t];
        __syncthreads();
        for (int co = threadIdx.x; co < Cout; co += 256) {
            const int qq = co / 4, j = co % 4;
            float s = 0.f;
            for (int l = 0; l < npl; ++l) s += red[(l * nq + qq) * 4 + j];
            if (t < NT) atomicAdd(&dw[(size_t)t * Cout + co], s);
            else if (dbias) atomicAdd(&dbias[co], s);
        }
    }
}

__global__ void __launch_bounds__(256) k_c1_zero(float* __restrict__ dw, int nw, float* __restrict__ dbias, int nb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nw) dw[i] = 0.f;
    else if (dbias && i < nw + nb) dbias[i - nw] = 0.f;
}

bool conv_c1_supported(const ConvGeom& g) {
    return g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && !g.transposed && (g.Cin == 1 || g.Cin == 3) &&
           (g.Cout % 4) == 0 && g.Cout <= (g.Cin == 1 ? 1024 : 256) && (256 % (g.Cout / 4)) == 0 && g.H == g.Ho &&
           g.W == g.Wo && g.W <= C1_MAXW;
}
template <typename T>
static int conv_c1_fwd_impl(const ConvGeom& g, const float* x, const float* w, const float* bias, T* y, int act, void* stream,
                            float* amax = nullptr) {
    unsigned grid = (unsigned)(g.B * g.H);
    if (grid > 256 * 8) grid = 256 * 8;
    const size_t lds = sizeof(float) * 3 * (g.W + 2) * g.Cin;
    if (g.Cin == 3) DASR_LAUNCH((k_conv3x3_c1_fwd<T, 3>), dim3(grid), dim3(256), lds, stream, x, w, bias, y, g.B, g.H, g.W, g.Cout, act, amax);
    else            DASR_LAUNCH((k_conv3x3_c1_fwd<T, 1>), dim3(grid), dim3(256), lds, stream, x, w, bias, y, g.B, g.H, g.W, g.Cout, act, amax);
    DASR_RETURN_LAUNCH_STATUS();
}
int conv_c1_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, int act, void* stream,
                float* amax) {
    return conv_c1_fwd_impl<float>(g, x, w, bias, y, act, stream, amax);
}
int conv_c1_fwd_bf16(const ConvGeom& g, const float* x, const float* w, const float* bias, bf16_t* y, int act, void* stream) {
    return conv_c1_fwd_impl<bf16_t>(g, x, w, bias, y, act, stream);
}
// yact may be null (dy is already the gradient w.r.t. the convolution output)
template <typename T>
static int conv_c1_wgrad_impl(const ConvGeom& g, const float* x, const T* dy, const T* yact, int act, float* dw, float* dbias,
                              void* stream) {
    // one launch clears both accumulators (two memsets were two more dispatches on a 130 us kernel)
    DASR_LAUNCH(k_c1_zero, dim3(dasr_cdiv((size_t)(9 * g.Cin + 1) * g.Cout, 256)), dim3(256), 0, stream, dw, 9 * g.Cin * g.Cout,
                dbias, g.Cout);
    unsigned grid = (unsigned)(g.B * g.H);
#ifndef C1_WGRAD_WGS_BF16
#define C1_WGRAD_WGS_BF16 1024
#endif
#ifndef C1_WGRAD_WGS_F32
#define C1_WGRAD_WGS_F32 512
#endif
    // fp32: two workgroups per CU (more only adds float atomics at the end, measured); bf16: four (124 VGPRs)
    const unsigned cap = sizeof(T) == 2 ? C1_WGRAD_WGS_BF16 : C1_WGRAD_WGS_F32;
    if (grid > cap) grid = cap;
    const size_t lds = sizeof(float) * (256 * 4 + 3 * (g.W + 2) * g.Cin);
#define C1_WGRAD(CIN, HASY) \
    DASR_LAUNCH((k_conv3x3_c1_wgrad<T, CIN, HASY>), dim3(grid), dim3(256), lds, stream, x, dy, yact, dw, dbias, g.B, g.H, g.W, g.Cout, act)
    const bool hasy = yact != nullptr && act != DASR_ACT_NONE;
    if (g.Cin == 3) { if (hasy) C1_WGRAD(3, true); else C1_WGRAD(3, false); }
    else            { if (hasy) C1_WGRAD(1, true); else C1_WGRAD(1, false); }
#undef C1_WGRAD
    DASR_RETURN_LAUNCH_STATUS();
}
int conv_c1_wgrad(const ConvGeom& g, const float* x, const float* dy, const float* yact, int act, float* dw, float* dbias,
                  void* stream) {
    return conv_c1_wgrad_impl<float>(g, x, dy, yact, act, dw, dbias, stream);
}
int conv_c1_wgrad_bf16(const ConvGeom& g, const float* x, const bf16_t* dy, const bf16_t* yact, int act, float* dw,
                       float* dbias, void* stream) {
    return conv_c1_wgrad_impl<bf16_t>(g, x, dy, yact, act, dw, dbias, stream);
}
