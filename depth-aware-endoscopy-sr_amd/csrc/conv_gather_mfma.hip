// conv_gather_mfma.hip — strided and transposed convolutions of the encoder (sftmd_arch.py:745-749:
// 3x3 stride-2 Conv2d 32->64->128, ConvTranspose2d 128->L, Conv2d L->L stride 2) on the fp32 matrix cores.
//
// These layers are 0.8 % of the FLOPs and work on 64x80 .. 32x40 images, so they do not get an LDS-tiled
// kernel: operands are gathered straight from global memory (L2-resident: the whole encoder state of a
// frame is a few MB), one 32-pixel M-tile per wave, no barriers.
//
//   GATHER 0 ("conv"):       src(m, tap) = (oy*stride - pad + kh, ox*stride - pad + kw)
//   GATHER 1 ("transposed"): src(m, tap) = ((oy + pad - kh)/stride, (ox + pad - kw)/stride) when divisible
// forward of Conv2d = GATHER 0, forward of ConvTranspose2d = GATHER 1; dgrad swaps them (and reads the
// HWIO kernel transposed, WT = 1).  wgrad: M = ci, N = co, K = output pixels, one (tap, ci-tile, co-pair)
// per wave, pixel range split over blockIdx.y into slabs summed by k_wgrad_reduce-style reduction.
#include "dasr_common.h"
#include "conv_kernels.h"

struct GatherArgs {
    const float* in;     // [B,Hi,Wi,Kd]   (gathered operand)
    const float* w;      // HWIO of the forward convolution
    const float* bias;
    float* out;          // [B,Ho,Wo,Nd]
    int B, Hi, Wi, Kd, Ho, Wo, Nd;
    int KH, KW, stride, pad, gather, wt, act, accumulate;
};

__device__ __forceinline__ bool gather_src(const GatherArgs& a, int oy, int ox, int kh, int kw, int& iy, int& ix) {
    if (a.gather == 0) {
        iy = oy * a.stride - a.pad + kh;
        ix = ox * a.stride - a.pad + kw;
    } else {
        int ty = oy + a.pad - kh, tx = ox + a.pad - kw;
        if (ty < 0 || tx < 0 || (ty % a.stride) != 0 || (tx % a.stride) != 0) return false;
        iy = ty / a.stride;
        ix = tx / a.stride;
    }
    return iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
}

// Output-pixel order.  For the "transposed" gather with stride s only every s-th row/column of taps lands on an
// input pixel, and which taps do depends on (oy mod s, ox mod s).  Enumerating the output pixels class by class
// (all pixels of one residue class are consecutive) makes tap validity uniform inside a 32-pixel M-tile, so a
// wave can skip the dead taps (3 of 4 at stride 2) instead of multiplying zeros.
__device__ __forceinline__ bool gather_pixel(int B, int Ho, int Wo, int stride, int by_class, size_t m, int& b,
                                             int& oy, int& ox) {
    const size_t M = (size_t)B * Ho * Wo;
    if (m >= M) return false;
    if (!by_class || stride <= 1) {
        ox = (int)(m % Wo);
        oy = (int)((m / Wo) % Ho);
        b = (int)(m / ((size_t)Wo * Ho));
        return true;
    }
    size_t rest = m;
    for (int cy = 0; cy < stride; ++cy)
        for (int cx = 0; cx < stride; ++cx) {
            const int hc = (Ho - cy + stride - 1) / stride, wc = (Wo - cx + stride - 1) / stride;
            const size_t cnt = (size_t)B * hc * wc;
            if (rest < cnt) {
                ox = cx + stride * (int)(rest % wc);
                oy = cy + stride * (int)((rest / wc) % hc);
                b = (int)(rest / ((size_t)wc * hc));
                return true;
            }
            rest -= cnt;
        }
    return false;
}

template <int NT>
__global__ void __launch_bounds__(256) k_conv_gather_mfma(GatherArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const size_t m = ((size_t)blockIdx.x * 4 + wv) * 32 + li;      // this lane's output pixel (A row)
    const int n0 = blockIdx.y * 32 * NT;
    int ox = 0, oy = 0, b = 0;
    const bool mvalid = gather_pixel(a.B, a.Ho, a.Wo, a.stride, a.gather, m, b, oy, ox);
    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    // Which taps land on a source pixel: per lane (okmask) and for any lane of this M-tile (alive, wave-uniform: dead
    // taps - 3 of 4 of a stride-2 transposed gather - are skipped, not multiplied by zeros).
    const int T = a.KH * a.KW;                                      // <= 32 (host check)
    unsigned okmask = 0, alive = 0;
    for (int tap = 0; tap < T; ++tap) {
        int iy, ix;
        const bool ok = mvalid && gather_src(a, oy, ox, tap / a.KW, tap % a.KW, iy, ix);
        okmask |= ok ? 1u << tap : 0u;
        alive |= __any(ok) ? 1u << tap : 0u;
    }
    // One trip = one tap x 32 channels = four 8-channel steps: 4 * (1 + NT) operand loads, 16 * NT MFMAs.  The trips are
    // software-pipelined over two statically named operand sets: the loads of trip i+1 are issued (unconditionally, from
    // clamped addresses - a branch around them makes the compiler wait for them at the join) before the MFMAs of trip i,
    // so the matrix pipe works while the L2 round trip is in flight (un-pipelined: 31 % of the fp32 MFMA peak).
    auto load = [&](int tap, int c0, float4 (&A)[4], float4 (&Bf)[4][NT]) {
        int iy = 0, ix = 0;
        const bool ok = (okmask >> tap) & 1u;
        gather_src(a, oy, ox, tap / a.KW, tap % a.KW, iy, ix);
        const float* src = ok ? a.in + (((size_t)b * a.Hi + iy) * a.Wi + ix) * a.Kd : a.in;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cc = c0 + 8 * u < a.Kd ? c0 + 8 * u : c0;      // clamped: the extra steps are skipped in mma()
            A[u] = *(const float4*)(src + cc + 4 * lh);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int nn = n0 + 32 * n + li;
                // [tap][n][k] with k contiguous: the transposed half for the forward, the HWIO half for dgrad
                Bf[u][n] = *(const float4*)(a.w + ((size_t)tap * a.Nd + nn) * a.Kd + cc + 4 * lh);
            }
        }
    };
    auto mma = [&](int tap, int c0, const float4 (&A)[4], const float4 (&Bf)[4][NT]) {
        const bool ok = (okmask >> tap) & 1u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c0 + 8 * u >= a.Kd) break;                          // wave-uniform
            const float4 av = ok ? A[u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, Bf[u][n].x, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, Bf[u][n].y, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, Bf[u][n].z, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, Bf[u][n].w, acc[n], 0, 0, 0);
            }
        }
    };
    // next (tap, c0) of the trip sequence; false when the sequence is over (all wave-uniform)
    auto advance = [&](int& tap, int& c0) {
        c0 += 32;
        if (c0 < a.Kd) return true;
        c0 = 0;
        for (++tap; tap < T; ++tap)
            if ((alive >> tap) & 1u) return true;
        return false;
    };
    int tap = 0, c0 = 0;
    while (tap < T && !((alive >> tap) & 1u)) ++tap;
    if (tap < T) {
        float4 A0[4], B0[4][NT], A1[4], B1[4][NT];
        load(tap, c0, A0, B0);
        while (true) {
            int tap1 = tap, c1 = c0;
            const bool more1 = advance(tap1, c1);
            load(more1 ? tap1 : tap, more1 ? c1 : c0, A1, B1);
            mma(tap, c0, A0, B0);
            if (!more1) break;
            int tap2 = tap1, c2 = c1;
            const bool more2 = advance(tap2, c2);
            load(more2 ? tap2 : tap1, more2 ? c2 : c1, A0, B0);
            mma(tap1, c1, A1, B1);
            if (!more2) break;
            tap = tap2;
            c0 = c2;
        }
    }
    // D: column = lane&31 = output channel, rows = pixels of this wave's M-tile.  Lane li (either half) decoded pixel
    // mbase + li at the top of the kernel: its linear output index is fetched from that lane (decoding it again per
    // accumulator register - divisions and the residue-class walk, 16 * NT times per lane - cost more than the MFMAs of
    // the small layers).
    const int mypix = mvalid ? (b * a.Ho + oy) * a.Wo + ox : -1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pix = __shfl(mypix, (r & 3) + 8 * (r >> 2) + 4 * lh, 64);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int nn = n0 + 32 * n + li;
            const float bv = a.bias ? a.bias[nn] : 0.f;
            if (pix < 0) continue;
            float v = dasr_act(acc[n][r] + bv, a.act);
            const size_t o = (size_t)pix * a.Nd + nn;
            if (a.accumulate) v += a.out[o];
            a.out[o] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ wgrad
struct GatherWgradArgs {
    const float* x;      // forward input  [B,H,W,Cin]
    const float* dy;     // dconv          [B,Ho,Wo,Cout]
    float* slabs;        // [P][taps][Cin][Cout]
    int B, H, W, Cin, Ho, Wo, Cout;
    int KH, KW, stride, pad, transposed;
    int P;
};

__global__ void __launch_bounds__(256) k_conv_gather_wgrad_mfma(GatherWgradArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const int ci_tiles = a.Cin / 32, co_pairs = a.Cout / 64;
    const int item = blockIdx.x * 4 + wv;
    const int nitems = a.KH * a.KW * ci_tiles * co_pairs;
    if (item >= nitems) return;     // whole wave exits together (item is wave-uniform)
    const int cp = item % co_pairs, ct = (item / co_pairs) % ci_tiles, tap = item / (co_pairs * ci_tiles);
    const int kh = tap / a.KW, kw = tap % a.KW;
    // K runs over the pixels of the SMALLER image: output pixels for a strided convolution, INPUT pixels for a
    // transposed one (oy = iy*stride - pad + kh is then always a whole pixel; walking the output pixels instead hits
    // a live (pixel, tap) pair only once in stride^2 steps: 9.3 ms for the 128 -> 256 layer at 32 frames of 256x320)
    const size_t M = a.transposed ? (size_t)a.B * a.H * a.W : (size_t)a.B * a.Ho * a.Wo;
    const size_t per = (M + a.P - 1) / a.P;
    const size_t m0 = (size_t)blockIdx.y * per;
    const size_t m1 = m0 + per < M ? m0 + per : M;
    GatherArgs ga{};
    ga.Hi = a.H; ga.Wi = a.W; ga.stride = a.stride; ga.pad = a.pad; ga.gather = a.transposed;
    f32x16 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    // Eight pixel pairs per trip: all 24 operand loads are issued before the first MFMA (one load -> MFMA per trip made
    // this loop a chain of global-memory latencies: 2.4 ms for a 3 GFLOP layer).  32-bit index arithmetic (M < 2^31).
    // (Software-pipelining the trips over two operand sets, as in the forward kernel, was measured and is worse here:
    // 208 VGPRs halve the resident waves and the 6.3 ms it took against 4.5 ms says the waves were hiding more latency.)
    constexpr int GU = 8;
    const unsigned Wm = (unsigned)(a.transposed ? a.W : a.Wo), HWm = (unsigned)(a.transposed ? a.H * a.W : a.Ho * a.Wo);
    for (size_t mb = m0; mb < m1; mb += 2 * GU) {
        float av[GU], b0[GU], b1[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const size_t m = mb + 2 * u + lh;
            av[u] = b0[u] = b1[u] = 0.f;
            if (m < m1) {
                const unsigned mm = (unsigned)m, b = mm / HWm, rem = mm - b * HWm;
                const int py = (int)(rem / Wm), px = (int)(rem - (unsigned)py * Wm);
                int iy, ix, oy, ox;
                bool ok;
                if (a.transposed) {
                    iy = py; ix = px;
                    oy = iy * a.stride - a.pad + kh;
                    ox = ix * a.stride - a.pad + kw;
                    ok = oy >= 0 && oy < a.Ho && ox >= 0 && ox < a.Wo;
                } else {
                    oy = py; ox = px;
                    ok = gather_src(ga, oy, ox, kh, kw, iy, ix);
                }
                if (ok) {
                    av[u] = a.x[(((size_t)b * a.H + iy) * a.W + ix) * a.Cin + 32 * ct + li];
                    const float* dp = a.dy + (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.Cout + 64 * cp + li;
                    b0[u] = dp[0];
                    b1[u] = dp[32];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b0[u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b1[u], acc[1], 0, 0, 0);
        }
    }
    float* slab = a.slabs + (size_t)blockIdx.y * a.KH * a.KW * a.Cin * a.Cout;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * lh;
            slab[((size_t)tap * a.Cin + ci) * a.Cout + 64 * cp + 32 * n + li] = acc[n][r];
        }
}

// ------------------------------------------------------------------------------------------ host side
bool conv_gather_fwd_supported(const ConvGeom& g) { return (g.Cin % 8) == 0 && (g.Cout % 32) == 0 && g.KH * g.KW <= 32; }
bool conv_gather_dgrad_supported(const ConvGeom& g) { return (g.Cout % 8) == 0 && (g.Cin % 32) == 0 && g.KH * g.KW <= 32; }
bool conv_gather_wgrad_supported(const ConvGeom& g) { return (g.Cin % 32) == 0 && (g.Cout % 64) == 0; }

static int launch_gather(GatherArgs& a, void* stream) {
    size_t M = (size_t)a.B * a.Ho * a.Wo;
    unsigned gx = dasr_cdiv(M, 128);
    if ((a.Nd % 64) == 0) {
        DASR_LAUNCH((k_conv_gather_mfma<2>), dim3(gx, a.Nd / 64), dim3(256), 0, stream, a);
    } else {
        DASR_LAUNCH((k_conv_gather_mfma<1>), dim3(gx, a.Nd / 32), dim3(256), 0, stream, a);
    }
    DASR_RETURN_LAUNCH_STATUS();
}
int conv_gather_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, int act,
                    void* stream) {
    GatherArgs a{x, w + (size_t)g.KH * g.KW * g.Cin * g.Cout, bias, y, g.B, g.H, g.W, g.Cin, g.Ho, g.Wo, g.Cout, g.KH,
                 g.KW, g.stride, g.pad, g.transposed ? 1 : 0, 0, act, 0};
    return launch_gather(a, stream);
}
int conv_gather_dgrad(const ConvGeom& g, const float* dconv, const float* w, float* dx, int accumulate, void* stream) {
    GatherArgs a{dconv, w, nullptr, dx, g.B, g.Ho, g.Wo, g.Cout, g.H, g.W, g.Cin, g.KH, g.KW, g.stride, g.pad,
                 g.transposed ? 0 : 1, 1, DASR_ACT_NONE, accumulate};
    return launch_gather(a, stream);
}
static int gather_wgrad_P(const ConvGeom& g) {
    int items = g.KH * g.KW * (g.Cin / 32) * (g.Cout / 64);
    int blocks = (items + 3) / 4;
    int P = 1024 / blocks;
    if (P < 1) P = 1;
    size_t M = g.transposed ? (size_t)g.B * g.H * g.W : (size_t)g.B * g.Ho * g.Wo;
    size_t maxP = (M + 255) / 256;
    if ((size_t)P > maxP) P = (int)maxP;
    if (P < 1) P = 1;
    return P;
}
size_t conv_gather_wgrad_workspace(const ConvGeom& g) {
    return sizeof(float) * (size_t)gather_wgrad_P(g) * g.KH * g.KW * g.Cin * g.Cout;
}
int conv_gather_wgrad(const ConvGeom& g, const float* x, const float* dconv, float* dw, void* workspace, void* stream) {
    int P = gather_wgrad_P(g);
    GatherWgradArgs a{x, dconv, (float*)workspace, g.B, g.H, g.W, g.Cin, g.Ho, g.Wo, g.Cout,
                      g.KH, g.KW, g.stride, g.pad, g.transposed ? 1 : 0, P};
    int items = g.KH * g.KW * (g.Cin / 32) * (g.Cout / 64);
    DASR_LAUNCH(k_conv_gather_wgrad_mfma, dim3((items + 3) / 4, P), dim3(256), 0, stream, a);
    return wgrad_reduce_launch((const float*)workspace, dw, (size_t)g.KH * g.KW * g.Cin * g.Cout, P, stream);
}
