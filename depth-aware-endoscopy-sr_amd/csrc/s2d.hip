// s2d.hip — the encoder's stride-2 layers on the bf16 stride-1 MFMA kernels (mixed-precision path only).
//
// Encoder.forward (sftmd_arch.py:745-749, 771-783): Conv2d 32->64->128 (3x3, stride 2, pad 1), ConvTranspose2d 128->L
// (3x3, stride 2, pad 1, NO output_padding: its output is (2H-1) x (2W-1)), Conv2d L->L (stride 2).  On fp32 activations they run on the gather kernels
// (conv_gather_mfma.hip: operands straight from L2, 27-50 % of the fp32 MFMA peak; 19 ms of the 174 ms bf16 step at
// 32 x 256x320).  With bf16 activations they are re-expressed as stride-1 3x3 convolutions, which the tuned kernels of
// conv_bf16_mfma.hip run 3-5x faster even though a quarter of the expanded taps are structural zeros:
//
//   stride-2 conv:   x'[i][j][(2py+px)C + c] = x[2i+py][2j+px][c]   (space-to-depth, zero beyond an odd edge)
//                    y[oy][ox] = sum_{kh,kw} x[2oy-1+kh][2ox-1+kw] w[kh][kw]
//                              = sum_{r,s} x'[oy+r-1][ox+s-1] W'[r][s],   W'[r][s][(py,px,c)][co] = w[kh(r,py)][kw(s,px)][c][co]
//                    with kh(0,1) = 0, kh(1,0) = 1, kh(1,1) = 2, nothing else (tap row / column 2 of W' is zero)
//   transposed conv: y[2i+a][2j+b][co] = conv3x3(x; W')[i][j][4co + 2a + b]  (PixelShuffle(2), fused in the conv epilogue)
//                    W'[r][s][ci][4co+2a+b] = w[kh(a,r)][kw(b,s)][ci][co],  kh(0,1) = 1, kh(1,1) = 2, kh(1,2) = 0
//                    (oy = 2 iy - 1 + kh: even rows see kh = 1 only, odd rows kh = 2 from iy = i and kh = 0 from iy = i + 1)
//                    The PixelShuffle image is 2H x 2W; the reference's layer has no output_padding, so its last row and
//                    column do not exist there: the consumer (space-to-depth of the next stride-2 layer) is given the
//                    VALID extents (2H-1, 2W-1), reads zeros beyond them - the next layer's zero padding - and its
//                    adjoint writes a zero gradient there, so nothing of the extra row / column reaches any result.
//
// This file: the two data movers (space-to-depth and its adjoint) and the weight expansions / gradient collapses.  The
// convolutions themselves, their PixelShuffle / LeakyReLU epilogues and all three gradients are the existing entry points.
#include "dasr_common.h"
#include "bf16.h"

// ---- space-to-depth: one thread = 4 channels of one output phase; blockIdx.y = (b, i)
template <typename TI>
__global__ void __launch_bounds__(256) k_space_to_depth2(const TI* __restrict__ x, bf16_t* __restrict__ y, int H, int W,
                                                         int C, int Hs, int Ws, int Hv, int Wv) {
    const int b = blockIdx.y / Hs, i = blockIdx.y - b * Hs;
    const int c4n = C / 4, n = Ws * 4 * c4n;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        const int c4 = idx % c4n, q = (idx / c4n) & 3, j = idx / (4 * c4n);
        const int h = 2 * i + (q >> 1), w = 2 * j + (q & 1);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h < Hv && w < Wv) v = ld4(x + (((size_t)b * H + h) * W + w) * C + 4 * c4);
        st4(y + ((((size_t)b * Hs + i) * Ws + j) * 4 + q) * C + 4 * c4, v);
    }
}
// adjoint: dx[b][h][w][c] (+)= dy'[b][h/2][w/2][(2(h&1) + (w&1))C + c]; blockIdx.y = (b, h)
template <typename TO>
__global__ void __launch_bounds__(256) k_depth_to_space2_bwd(const bf16_t* __restrict__ dy, TO* __restrict__ dx, int H,
                                                             int W, int C, int Hs, int Ws, int accumulate, int Hv,
                                                             int Wv) {
    const int b = blockIdx.y / H, h = blockIdx.y - b * H;
    const int c4n = C / 4, n = W * c4n;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        const int c4 = idx % c4n, w = idx / c4n;
        const int q = 2 * (h & 1) + (w & 1);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);       // beyond the valid extents the forward read zeros: zero gradient
        if (h < Hv && w < Wv) v = ld4(dy + ((((size_t)b * Hs + (h >> 1)) * Ws + (w >> 1)) * 4 + q) * C + 4 * c4);
        TO* p = dx + (((size_t)b * H + h) * W + w) * C + 4 * c4;
        if (accumulate) {
            const float4 o = ld4(p);
            v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w);
        }
        st4(p, v);
    }
}

// ---- weight expansions.  Source: fp32 HWIO [9][Cin][Cout] (first half of a packed kernel).  Destination: the bf16 packed
// layout of conv_bf16_mfma.hip, [2][9][I'][O']: HWIO, then every tap transposed ([tap][O'][I']).
__device__ __forceinline__ int s2_kh(int r, int py) { return r == 0 ? (py == 1 ? 0 : -1) : r == 1 ? (py == 0 ? 1 : 2) : -1; }
__device__ __forceinline__ int t2_kh(int a, int r) { return a == 0 ? (r == 1 ? 1 : -1) : (r == 1 ? 2 : r == 2 ? 0 : -1); }

__global__ void __launch_bounds__(256) k_weight_expand_s2(const float* __restrict__ w, bf16_t* __restrict__ out, int Cin,
                                                          int Cout) {
    const int I2 = 4 * Cin;
    const size_t n = (size_t)9 * I2 * Cout;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
        const int co = (int)(idx % Cout), ii = (int)((idx / Cout) % I2), tap = (int)(idx / ((size_t)Cout * I2));
        const int q = ii / Cin, ci = ii - q * Cin;
        const int kh = s2_kh(tap / 3, q >> 1), kw = s2_kh(tap % 3, q & 1);
        const float v = (kh >= 0 && kw >= 0) ? w[((size_t)(kh * 3 + kw) * Cin + ci) * Cout + co] : 0.f;
        st1(out + idx, v);
        st1(out + n + ((size_t)tap * Cout + co) * I2 + ii, v);
    }
}
// dW[kh][kw][ci][co] = dW'[r][s][(py,px,ci)][co] at the one expanded slot that holds it
__global__ void __launch_bounds__(256) k_weight_collapse_s2(const float* __restrict__ dwe, float* __restrict__ dw, int Cin,
                                                            int Cout) {
    const size_t n = (size_t)9 * Cin * Cout;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
        const int co = (int)(idx % Cout), ci = (int)((idx / Cout) % Cin), k = (int)(idx / ((size_t)Cout * Cin));
        const int kh = k / 3, kw = k % 3;
        const int r = kh == 0 ? 0 : 1, py = kh == 1 ? 0 : 1, s = kw == 0 ? 0 : 1, px = kw == 1 ? 0 : 1;
        dw[idx] = dwe[((size_t)(r * 3 + s) * 4 * Cin + (size_t)(2 * py + px) * Cin + ci) * Cout + co];
    }
}
__global__ void __launch_bounds__(256) k_weight_expand_t2(const float* __restrict__ w, const float* __restrict__ bias,
                                                          bf16_t* __restrict__ out, float* __restrict__ bias_out, int Cin,
                                                          int Cout) {
    const int O2 = 4 * Cout;
    const size_t n = (size_t)9 * Cin * O2;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
        const int nn = (int)(idx % O2), ci = (int)((idx / O2) % Cin), tap = (int)(idx / ((size_t)O2 * Cin));
        const int co = nn >> 2, a = (nn >> 1) & 1, b = nn & 1;
        const int kh = t2_kh(a, tap / 3), kw = t2_kh(b, tap % 3);
        const float v = (kh >= 0 && kw >= 0) ? w[((size_t)(kh * 3 + kw) * Cin + ci) * Cout + co] : 0.f;
        st1(out + idx, v);
        st1(out + n + ((size_t)tap * O2 + nn) * Cin + ci, v);
        if (bias && idx < (size_t)O2) bias_out[idx] = bias[idx >> 2];
    }
}
__global__ void __launch_bounds__(256) k_weight_collapse_t2(const float* __restrict__ dwe, const float* __restrict__ dbe,
                                                            float* __restrict__ dw, float* __restrict__ db, int Cin,
                                                            int Cout) {
    const size_t n = (size_t)9 * Cin * Cout;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
        const int co = (int)(idx % Cout), ci = (int)((idx / Cout) % Cin), k = (int)(idx / ((size_t)Cout * Cin));
        const int kh = k / 3, kw = k % 3;
        const int a = kh == 1 ? 0 : 1, r = kh == 0 ? 2 : 1, b = kw == 1 ? 0 : 1, s = kw == 0 ? 2 : 1;
        dw[idx] = dwe[((size_t)(r * 3 + s) * Cin + ci) * 4 * Cout + 4 * co + 2 * a + b];
        if (db && dbe && idx < (size_t)Cout)
            db[idx] = (dbe[4 * idx] + dbe[4 * idx + 1]) + (dbe[4 * idx + 2] + dbe[4 * idx + 3]);
    }
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int dasr_space_to_depth2_bf16(const void* x, int x_is_bf16, unsigned short* y, int B, int H, int W, int C,
                                         int Hv, int Wv, void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(y);
    DASR_CHECK_SHAPE(B > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0 && Hv >= 0 && Hv <= H && Wv >= 0 && Wv <= W);
    const int Hs = (H + 1) / 2, Ws = (W + 1) / 2;
    unsigned gx = dasr_cdiv((size_t)Ws * C, 256);
    if (gx > 64) gx = 64;
    const dim3 grid(gx, B * Hs);
    if (B * Hs > 65535) return DASR_E_SHAPE;
    if (x_is_bf16) DASR_LAUNCH((k_space_to_depth2<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, H, W, C, Hs, Ws, Hv, Wv);
    else           DASR_LAUNCH((k_space_to_depth2<float>), grid, dim3(256), 0, stream, (const float*)x, (bf16_t*)y, H, W, C, Hs, Ws, Hv, Wv);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_depth_to_space2_bwd_bf16(const unsigned short* dy, void* dx, int dx_is_bf16, int accumulate, int B,
                                             int H, int W, int C, int Hv, int Wv, void* stream) {
    DASR_CHECK_PTR(dy); DASR_CHECK_PTR(dx);
    DASR_CHECK_SHAPE(B > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0 && Hv >= 0 && Hv <= H && Wv >= 0 && Wv <= W);
    const int Hs = (H + 1) / 2, Ws = (W + 1) / 2;
    if (B * H > 65535) return DASR_E_SHAPE;
    unsigned gx = dasr_cdiv((size_t)W * (C / 4), 256);
    if (gx > 64) gx = 64;
    const dim3 grid(gx, B * H);
    if (dx_is_bf16) DASR_LAUNCH((k_depth_to_space2_bwd<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)dy, (bf16_t*)dx, H, W, C, Hs, Ws, accumulate, Hv, Wv);
    else            DASR_LAUNCH((k_depth_to_space2_bwd<float>), grid, dim3(256), 0, stream, (const bf16_t*)dy, (float*)dx, H, W, C, Hs, Ws, accumulate, Hv, Wv);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_weight_expand_s2_bf16(const float* w_hwio, unsigned short* out, int Cin, int Cout, void* stream) {
    DASR_CHECK_PTR(w_hwio); DASR_CHECK_PTR(out);
    DASR_CHECK_SHAPE(Cin > 0 && Cout > 0);
    DASR_LAUNCH(k_weight_expand_s2, dim3(dasr_ew_grid((size_t)36 * Cin * Cout)), dim3(256), 0, stream, w_hwio, (bf16_t*)out, Cin, Cout);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_weight_collapse_s2(const float* dw_expanded, float* dw_hwio, int Cin, int Cout, void* stream) {
    DASR_CHECK_PTR(dw_expanded); DASR_CHECK_PTR(dw_hwio);
    DASR_CHECK_SHAPE(Cin > 0 && Cout > 0);
    DASR_LAUNCH(k_weight_collapse_s2, dim3(dasr_ew_grid((size_t)9 * Cin * Cout)), dim3(256), 0, stream, dw_expanded, dw_hwio, Cin, Cout);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_weight_expand_t2_bf16(const float* w_hwio, const float* bias, unsigned short* out, float* bias_out,
                                          int Cin, int Cout, void* stream) {
    DASR_CHECK_PTR(w_hwio); DASR_CHECK_PTR(out);
    if (bias) DASR_CHECK_PTR(bias_out);
    DASR_CHECK_SHAPE(Cin > 0 && Cout > 0 && (size_t)9 * Cin >= 1);
    DASR_LAUNCH(k_weight_expand_t2, dim3(dasr_ew_grid((size_t)36 * Cin * Cout)), dim3(256), 0, stream, w_hwio, bias, (bf16_t*)out, bias_out, Cin, Cout);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_weight_collapse_t2(const float* dw_expanded, const float* dbias_expanded, float* dw_hwio, float* dbias,
                                       int Cin, int Cout, void* stream) {
    DASR_CHECK_PTR(dw_expanded); DASR_CHECK_PTR(dw_hwio);
    DASR_CHECK_SHAPE(Cin > 0 && Cout > 0);
    DASR_LAUNCH(k_weight_collapse_t2, dim3(dasr_ew_grid((size_t)9 * Cin * Cout)), dim3(256), 0, stream, dw_expanded, dbias_expanded, dw_hwio, dbias, Cin, Cout);
    DASR_RETURN_LAUNCH_STATUS();
}
