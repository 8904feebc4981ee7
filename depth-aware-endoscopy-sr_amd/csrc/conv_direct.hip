// conv_direct.hip — generic direct convolution (any kernel size / stride / ConvTranspose geometry /
// channel count), NHWC x HWIO, fp32 on the vector ALU.  This is the fall-back for the shapes the MFMA
// implicit-GEMM kernels (conv_mfma.hip) do not cover: the strided / transposed encoder layers, the
// 1- and 3-channel inputs, and odd channel counts.  dasr_conv2d_* in conv_api.hip dispatch between them.
#include "dasr_common.h"
#include "bf16.h"
#include "conv_kernels.h"

// ------------------------------------------------------------------------------------------ forward
__global__ void __launch_bounds__(256) k_conv_direct_fwd(ConvGeom g, const float* __restrict__ x,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ residual, float* __restrict__ y,
                                                         int act, int ps_r) {
    size_t n = (size_t)g.B * g.Ho * g.Wo * g.Cout;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        int co = (int)(idx % g.Cout);
        size_t pix = idx / g.Cout;
        int ox = (int)(pix % g.Wo), oy = (int)((pix / g.Wo) % g.Ho), b = (int)(pix / ((size_t)g.Wo * g.Ho));
        float acc = bias ? bias[co] : 0.f;
        for (int kh = 0; kh < g.KH; ++kh) {
            int iy;
            if (!g.transposed) {
                iy = oy * g.stride - g.pad + kh;
            } else {
                int ty = oy + g.pad - kh;
                if (ty < 0 || (ty % g.stride) != 0) continue;
                iy = ty / g.stride;
            }
            if (iy < 0 || iy >= g.H) continue;
            for (int kw = 0; kw < g.KW; ++kw) {
                int ix;
                if (!g.transposed) {
                    ix = ox * g.stride - g.pad + kw;
                } else {
                    int tx = ox + g.pad - kw;
                    if (tx < 0 || (tx % g.stride) != 0) continue;
                    ix = tx / g.stride;
                }
                if (ix < 0 || ix >= g.W) continue;
                const float* xp = x + (((size_t)b * g.H + iy) * g.W + ix) * g.Cin;
                const float* wp = w + ((size_t)(kh * g.KW + kw) * g.Cin) * g.Cout + co;
                for (int ci = 0; ci < g.Cin; ++ci) acc = fmaf(xp[ci], wp[(size_t)ci * g.Cout], acc);
            }
        }
        if (residual) acc += residual[idx];
        acc = dasr_act(acc, act);
        y[conv_out_index(g, b, oy, ox, co, ps_r)] = acc;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_conv_epilogue_bwd(ConvGeom g, const T* __restrict__ dy,
                                                           const T* __restrict__ y, T* __restrict__ dconv,
                                                           int act, int ps_r, float* __restrict__ amax) {
    __shared__ float s_part[16];
    size_t n = (size_t)g.B * g.Ho * g.Wo * g.Cout;
    float om = 0.f;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        int co = (int)(idx % g.Cout);
        size_t pix = idx / g.Cout;
        int ox = (int)(pix % g.Wo), oy = (int)((pix / g.Wo) % g.Ho), b = (int)(pix / ((size_t)g.Wo * g.Ho));
        size_t o = conv_out_index(g, b, oy, ox, co, ps_r);
        const float v = ld1(dy + o) * dasr_act_grad_from_out(ld1(y + o), act);
        st1(dconv + idx, v);
        om = dasr_amax1(om, v);
    }
    if (amax) dasr_amax_commit(amax, om, s_part, dasr_flat_wg(), dasr_flat_nwg());
}

// ------------------------------------------------------------------------------------------ dgrad
__global__ void __launch_bounds__(256) k_conv_direct_dgrad(ConvGeom g, const float* __restrict__ dconv,
                                                           const float* __restrict__ w, float* __restrict__ dx,
                                                           int accumulate) {
    size_t n = (size_t)g.B * g.H * g.W * g.Cin;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        int ci = (int)(idx % g.Cin);
        size_t pix = idx / g.Cin;
        int ix = (int)(pix % g.W), iy = (int)((pix / g.W) % g.H), b = (int)(pix / ((size_t)g.W * g.H));
        float acc = 0.f;
        for (int kh = 0; kh < g.KH; ++kh) {
            int oy;
            if (!g.transposed) {
                int ty = iy + g.pad - kh;
                if (ty < 0 || (ty % g.stride) != 0) continue;
                oy = ty / g.stride;
            } else {
                oy = iy * g.stride - g.pad + kh;
            }
            if (oy < 0 || oy >= g.Ho) continue;
            for (int kw = 0; kw < g.KW; ++kw) {
                int ox;
                if (!g.transposed) {
                    int tx = ix + g.pad - kw;
                    if (tx < 0 || (tx % g.stride) != 0) continue;
                    ox = tx / g.stride;
                } else {
                    ox = ix * g.stride - g.pad + kw;
                }
                if (ox < 0 || ox >= g.Wo) continue;
                const float* dp = dconv + (((size_t)b * g.Ho + oy) * g.Wo + ox) * g.Cout;
                const float* wp = w + ((size_t)(kh * g.KW + kw) * g.Cin + ci) * g.Cout;
                for (int co = 0; co < g.Cout; ++co) acc = fmaf(dp[co], wp[co], acc);
            }
        }
        dx[idx] = accumulate ? dx[idx] + acc : acc;
    }
}

// ------------------------------------------------------------------------------------------ wgrad
// One thread per (tap, ci, co); blockIdx.y splits the output pixels; partial sums go to dw with float atomics
// (dw zeroed by the caller's hipMemsetAsync).
__global__ void __launch_bounds__(256) k_conv_direct_wgrad(ConvGeom g, const float* __restrict__ x,
                                                           const float* __restrict__ dconv, float* __restrict__ dw,
                                                           size_t pix_per_split) {
    size_t nW = (size_t)g.KH * g.KW * g.Cin * g.Cout;
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nW) return;
    int co = (int)(e % g.Cout);
    int ci = (int)((e / g.Cout) % g.Cin);
    int tap = (int)(e / ((size_t)g.Cout * g.Cin));
    int kh = tap / g.KW, kw = tap % g.KW;
    size_t npix = (size_t)g.B * g.Ho * g.Wo;
    size_t p0 = (size_t)blockIdx.y * pix_per_split;
    size_t p1 = p0 + pix_per_split < npix ? p0 + pix_per_split : npix;
    float acc = 0.f;
    for (size_t p = p0; p < p1; ++p) {
        int ox = (int)(p % g.Wo), oy = (int)((p / g.Wo) % g.Ho), b = (int)(p / ((size_t)g.Wo * g.Ho));
        int iy, ix;
        if (!g.transposed) {
            iy = oy * g.stride - g.pad + kh;
            ix = ox * g.stride - g.pad + kw;
        } else {
            int ty = oy + g.pad - kh, tx = ox + g.pad - kw;
            if (ty < 0 || tx < 0 || (ty % g.stride) != 0 || (tx % g.stride) != 0) continue;
            iy = ty / g.stride;
            ix = tx / g.stride;
        }
        if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) continue;
        acc = fmaf(x[(((size_t)b * g.H + iy) * g.W + ix) * g.Cin + ci], dconv[p * g.Cout + co], acc);
    }
    atomicAdd(&dw[e], acc);
}

// dbias[co] = sum over rows of dconv [npix][Cout]; 4 row lanes x 64 channel lanes per block
__global__ void __launch_bounds__(256) k_colsum(const float* __restrict__ m, float* __restrict__ out, size_t rows,
                                                int C, size_t rows_per_split) {
    __shared__ float red[256];
    int c = blockIdx.x * 64 + (threadIdx.x & 63);
    int rl = threadIdx.x >> 6;
    size_t r0 = (size_t)blockIdx.y * rows_per_split;
    size_t r1 = r0 + rows_per_split < rows ? r0 + rows_per_split : rows;
    float acc = 0.f;
    if (c < C)
        for (size_t r = r0 + rl; r < r1; r += 4) acc += m[r * C + c];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0 && c < C) atomicAdd(&out[c], red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] +
                                                 red[threadIdx.x + 192]);
}

// narrow matrices (C <= 8, e.g. the 3-channel output gradient at HR): one thread sums whole rows
__global__ void __launch_bounds__(256) k_colsum_small(const float* __restrict__ m, float* __restrict__ out, size_t rows,
                                                      int C) {
    __shared__ float red[8][4];
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
        const float* p = m + r * C;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) acc[c] += p[c];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float v = acc[c];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < C) atomicAdd(&out[threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] +
                                                          red[threadIdx.x][3]);
}

// ------------------------------------------------------------------------------------------ host side
int conv_direct_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, const float* residual,
                    float* y, int act, int ps_r, void* stream) {
    size_t n = (size_t)g.B * g.Ho * g.Wo * g.Cout;
    DASR_LAUNCH(k_conv_direct_fwd, dim3(dasr_ew_grid(n)), dim3(256), 0, stream, g, x, w, bias, residual, y, act, ps_r);
    DASR_RETURN_LAUNCH_STATUS();
}
// Fast forms of the above (the generic kernel spends three 64-bit divisions and a scalar load pair per element).
// No PixelShuffle: the index map is the identity -> float4 streaming.
template <typename T>
__global__ void __launch_bounds__(256) k_conv_epilogue_bwd_flat4(const T* __restrict__ dy, const T* __restrict__ y,
                                                                 T* __restrict__ dconv, size_t n4, int act,
                                                                 float* __restrict__ amax) {
    __shared__ float s_part[16];
    float om = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 d = ld4(dy + 4 * i), v = ld4(y + 4 * i);
        const float4 o = make_float4(d.x * dasr_act_grad_from_out(v.x, act), d.y * dasr_act_grad_from_out(v.y, act),
                                     d.z * dasr_act_grad_from_out(v.z, act), d.w * dasr_act_grad_from_out(v.w, act));
        st4(dconv + 4 * i, o);
        om = dasr_amax4(om, o);
    }
    if (amax) dasr_amax_commit(amax, om, s_part, dasr_flat_wg(), dasr_flat_nwg());
}
// PixelShuffle(R): one thread per (conv pixel, shuffled channel c): R*R coalesced 4-byte reads of dy / y (consecutive
// lanes = consecutive c), one contiguous run of R*R floats written (co = c*R*R + i*R + j).  blockIdx.y = conv row
// (b*Ho + oy), so the only division left is e / Cq.
template <int R, typename T>
__global__ void __launch_bounds__(256) k_conv_epilogue_bwd_ps(const T* __restrict__ dy, const T* __restrict__ y,
                                                              T* __restrict__ dconv, int Wo, int Cq, int act, int rows,
                                                              float* __restrict__ amax) {
    __shared__ float s_part[16];
    const int e0 = blockIdx.x * 256 + threadIdx.x;
    const bool livee = e0 < Wo * Cq;                                      // (no early return: the workgroup meets in dasr_amax_commit)
    const int e = livee ? e0 : Wo * Cq - 1;
    const int ox = e / Cq, c = e - ox * Cq;
    const size_t srow = (size_t)Wo * R * Cq;                              // floats per shuffled row
    float om = 0.f;
    // blockIdx.y walks the conv rows (b*Ho + oy) with stride gridDim.y (the grid stays below DASR_AMAX_MAX_PARTS workgroups
    // when a maximum is wanted), TWO rows per trip: the 2 R*R loads of both are issued before the first store (one row per
    // trip left a dependent round trip per row: 1385 -> 1541 us on the 32 -> 128 PixelShuffle layer at 512 x 640)
    for (size_t row0 = blockIdx.y; row0 < (size_t)rows; row0 += 2 * (size_t)gridDim.y) {
        const size_t row1 = row0 + gridDim.y;
        const bool two = row1 < (size_t)rows;
        const size_t rws[2] = {row0, two ? row1 : row0};
        float out[2][R * R];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const T* dyp = dy + rws[h] * R * srow + (size_t)ox * R * Cq + c;
            const T* yp = y + rws[h] * R * srow + (size_t)ox * R * Cq + c;
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const size_t o = (size_t)i * srow + (size_t)j * Cq;
                    out[h][i * R + j] = ld1(dyp + o) * dasr_act_grad_from_out(ld1(yp + o), act);
                }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !two) break;
            T* dst = dconv + (rws[h] * Wo + ox) * (size_t)(Cq * R * R) + (size_t)c * (R * R);
            if (livee) {
                if (R == 2) {
                    st4(dst, make_float4(out[h][0], out[h][1], out[h][2], out[h][3]));
                } else {
#pragma unroll
                    for (int q = 0; q < R * R; ++q) st1(dst + q, out[h][q]);
                }
            }
#pragma unroll
            for (int q = 0; q < R * R; ++q) om = dasr_amax1(om, out[h][q]);
        }
    }
    if (amax) dasr_amax_commit(amax, om, s_part, dasr_flat_wg(), dasr_flat_nwg());
}

template <typename T>
static int conv_epilogue_bwd_impl(const ConvGeom& g, const T* dy, const T* y, T* dconv, int act, int ps_r, void* stream,
                                  float* amax = nullptr) {
    size_t n = (size_t)g.B * g.Ho * g.Wo * g.Cout;
    const size_t rows = (size_t)g.B * g.Ho;
    if (ps_r <= 1 && (n % 4) == 0) {
        DASR_LAUNCH((k_conv_epilogue_bwd_flat4<T>), dim3(dasr_ew_grid(n / 4)), dim3(256), 0, stream, dy, y, dconv, n / 4, act, amax);
    } else if ((ps_r == 2 || ps_r == 3) && (g.Cout % (ps_r * ps_r)) == 0 && rows < ((size_t)1 << 31) &&
               (size_t)g.Wo * (g.Cout / (ps_r * ps_r)) < (1u << 30)) {
        const int Cq = g.Cout / (ps_r * ps_r);
        const unsigned gx = dasr_cdiv((size_t)g.Wo * Cq, 256);
        // one conv row per workgroup up to 65535 rows; with a maximum to leave behind, few enough workgroups for one partial each
        size_t gy = rows < 65535 ? rows : 65535;
        if (amax && (size_t)gx * gy > DASR_AMAX_MAX_PARTS) gy = DASR_AMAX_MAX_PARTS / gx > 0 ? DASR_AMAX_MAX_PARTS / gx : 1;
        if (amax && (size_t)gx * gy > DASR_AMAX_MAX_PARTS) return DASR_E_UNSUPPORTED;
        const dim3 grid(gx, (unsigned)gy);
        if (ps_r == 2)
            DASR_LAUNCH((k_conv_epilogue_bwd_ps<2, T>), grid, dim3(256), 0, stream, dy, y, dconv, g.Wo, Cq, act, (int)rows, amax);
        else
            DASR_LAUNCH((k_conv_epilogue_bwd_ps<3, T>), grid, dim3(256), 0, stream, dy, y, dconv, g.Wo, Cq, act, (int)rows, amax);
    } else {
        DASR_LAUNCH((k_conv_epilogue_bwd<T>), dim3(dasr_ew_grid(n)), dim3(256), 0, stream, g, dy, y, dconv, act, ps_r, amax);
    }
    DASR_RETURN_LAUNCH_STATUS();
}
int conv_epilogue_bwd(const ConvGeom& g, const float* dy, const float* y, float* dconv, int act, int ps_r, void* stream,
                      float* amax) {
    return conv_epilogue_bwd_impl<float>(g, dy, y, dconv, act, ps_r, stream, amax);
}
int conv_epilogue_bwd_bf16(const ConvGeom& g, const bf16_t* dy, const bf16_t* y, bf16_t* dconv, int act, int ps_r,
                           void* stream) {
    return conv_epilogue_bwd_impl<bf16_t>(g, dy, y, dconv, act, ps_r, stream);
}
int conv_direct_dgrad(const ConvGeom& g, const float* dconv, const float* w, float* dx, int accumulate, void* stream) {
    size_t n = (size_t)g.B * g.H * g.W * g.Cin;
    DASR_LAUNCH(k_conv_direct_dgrad, dim3(dasr_ew_grid(n)), dim3(256), 0, stream, g, dconv, w, dx, accumulate);
    DASR_RETURN_LAUNCH_STATUS();
}
int conv_colsum(const float* m, float* out, size_t rows, int C, void* stream) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * C, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    if (C <= 8) {
        DASR_LAUNCH(k_colsum_small, dim3(dasr_ew_grid(rows)), dim3(256), 0, stream, m, out, rows, C);
        DASR_RETURN_LAUNCH_STATUS();
    }
    unsigned nsplit = (unsigned)((rows + 255) / 256);
    if (nsplit > 2048) nsplit = 2048;
    if (nsplit < 1) nsplit = 1;
    size_t rps = (rows + nsplit - 1) / nsplit;
    DASR_LAUNCH(k_colsum, dim3(dasr_cdiv(C, 64), nsplit), dim3(256), 0, stream, m, out, rows, C, rps);
    DASR_RETURN_LAUNCH_STATUS();
}
int conv_direct_wgrad(const ConvGeom& g, const float* x, const float* dconv, float* dw, void* stream) {
    size_t nW = (size_t)g.KH * g.KW * g.Cin * g.Cout;
    hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * nW, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    size_t npix = (size_t)g.B * g.Ho * g.Wo;
    unsigned gx = dasr_cdiv(nW, 256);
    unsigned nsplit = 4096 / gx;
    if (nsplit < 1) nsplit = 1;
    size_t max_split = (npix + 63) / 64;
    if (nsplit > max_split) nsplit = (unsigned)max_split;
    if (nsplit > 65535) nsplit = 65535;
    size_t pps = (npix + nsplit - 1) / nsplit;
    DASR_LAUNCH(k_conv_direct_wgrad, dim3(gx, nsplit), dim3(256), 0, stream, g, x, dconv, dw, pps);
    DASR_RETURN_LAUNCH_STATUS();
}
