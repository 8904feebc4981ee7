// conv_c1.hip — 3x3 convolution of a ONE-channel image (the depth map) to Cout channels with fused ReLU:
// SEAN.mlp_mask (normalization.py:36-40,61), forward and weight/bias gradient.  There is no reduction over
// input channels, so this is HBM-bound elementwise work: the forward writes B*H*W*Cout floats, the backward
// reads the incoming gradient and the saved activation once (ReLU backward fused) and reduces over pixels.
// Lane layout: a lane owns 4 consecutive output channels (float4), 256-byte runs per 16 lanes.
#include "dasr_common.h"
#include "conv_kernels.h"

__global__ void __launch_bounds__(256) k_conv3x3_c1_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int B,
                                                        int H, int W, int Cout, int act) {
    const int nq = Cout / 4;                       // channel quads
    const int q = threadIdx.x % nq, pl = threadIdx.x / nq, npl = 256 / nq;
    float4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = *(const float4*)(w + (size_t)t * Cout + 4 * q);
    const float4 bv = bias ? *(const float4*)(bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t npix = (size_t)B * H * W;
    if (pl >= npl) return;
    for (size_t p = (size_t)blockIdx.x * npl + pl; p < npix; p += (size_t)gridDim.x * npl) {
        const int px = (int)(p % W), py = (int)((p / W) % H);
        const size_t b = p / ((size_t)W * H);
        const float* xb = x + b * (size_t)H * W;
        float4 acc = bv;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = py + t / 3 - 1, ix = px + t % 3 - 1;
            const float d = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? xb[(size_t)iy * W + ix] : 0.f;
            acc.x = fmaf(d, wt[t].x, acc.x); acc.y = fmaf(d, wt[t].y, acc.y);
            acc.z = fmaf(d, wt[t].z, acc.z); acc.w = fmaf(d, wt[t].w, acc.w);
        }
        acc.x = dasr_act(acc.x, act); acc.y = dasr_act(acc.y, act);
        acc.z = dasr_act(acc.z, act); acc.w = dasr_act(acc.w, act);
        *(float4*)(y + p * Cout + 4 * q) = acc;
    }
}

// dw[tap][co] = sum_p x[p+tap] * dconv[p][co], dbias[co] = sum_p dconv[p][co], dconv = dy * act'(y)
__global__ void __launch_bounds__(256) k_conv3x3_c1_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ yact, float* __restrict__ dw,
                                                          float* __restrict__ dbias, int B, int H, int W, int Cout,
                                                          int act) {
    DASR_DYN_SMEM(smem);
    float* red = (float*)smem;                     // [256][40] partials
    const int nq = Cout / 4;
    const int q = threadIdx.x % nq, pl = threadIdx.x / nq, npl = 256 / nq;
    float4 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t npix = (size_t)B * H * W;
    const size_t stride = (size_t)gridDim.x * npl;
    if (pl < npl)
        for (size_t p0 = (size_t)blockIdx.x * npl + pl; p0 < npix; p0 += 4 * stride) {
            // four pixels per trip: all eight 16-byte loads are issued before the first is consumed
            float4 gq[4], yq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t p = p0 + u * stride;
                gq[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                yq[u] = make_float4(1.f, 1.f, 1.f, 1.f);
                if (p < npix) {
                    gq[u] = *(const float4*)(dy + p * Cout + 4 * q);
                    if (yact) yq[u] = *(const float4*)(yact + p * Cout + 4 * q);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t p = p0 + u * stride;
                if (p >= npix) continue;
                const int px = (int)(p % W), py = (int)((p / W) % H);
                const size_t b = p / ((size_t)W * H);
                const float* xb = x + b * (size_t)H * W;
                float4 g = gq[u];
                if (yact) {
                    g.x *= dasr_act_grad_from_out(yq[u].x, act); g.y *= dasr_act_grad_from_out(yq[u].y, act);
                    g.z *= dasr_act_grad_from_out(yq[u].z, act); g.w *= dasr_act_grad_from_out(yq[u].w, act);
                }
                acc[9].x += g.x; acc[9].y += g.y; acc[9].z += g.z; acc[9].w += g.w;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = py + t / 3 - 1, ix = px + t % 3 - 1;
                    const float d = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? xb[(size_t)iy * W + ix] : 0.f;
                    acc[t].x = fmaf(d, g.x, acc[t].x); acc[t].y = fmaf(d, g.y, acc[t].y);
                    acc[t].z = fmaf(d, g.z, acc[t].z); acc[t].w = fmaf(d, g.w, acc[t].w);
                }
            }
        }
    // reduce over the pixel lanes of the workgroup, then one float atomic per output per workgroup
#pragma unroll
    for (int t = 0; t < 10; ++t) *(float4*)(red + (threadIdx.x * 10 + t) * 4) = acc[t];
    __syncthreads();
    for (int e = threadIdx.x; e < 10 * Cout; e += 256) {
        const int t = e / Cout, co = e % Cout, qq = co / 4, j = co % 4;
        float s = 0.f;
        for (int l = 0; l < npl; ++l) s += red[((l * nq + qq) * 10 + t) * 4 + j];
        if (t < 9) atomicAdd(&dw[(size_t)t * Cout + co], s);
        else if (dbias) atomicAdd(&dbias[co], s);
    }
}

bool conv_c1_supported(const ConvGeom& g) {
    return g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && !g.transposed && g.Cin == 1 && (g.Cout % 4) == 0 &&
           g.Cout <= 1024 && (256 % (g.Cout / 4)) == 0 && g.H == g.Ho && g.W == g.Wo;
}
int conv_c1_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, int act, void* stream) {
    size_t npix = (size_t)g.B * g.H * g.W;
    int npl = 256 / (g.Cout / 4);
    unsigned grid = dasr_cdiv(npix, npl);
    if (grid > 256 * 16) grid = 256 * 16;
    DASR_LAUNCH(k_conv3x3_c1_fwd, dim3(grid), dim3(256), 0, stream, x, w, bias, y, g.B, g.H, g.W, g.Cout, act);
    DASR_RETURN_LAUNCH_STATUS();
}
// yact may be null (dy is already the gradient w.r.t. the convolution output)
int conv_c1_wgrad(const ConvGeom& g, const float* x, const float* dy, const float* yact, int act, float* dw, float* dbias,
                  void* stream) {
    hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * 9 * g.Cout, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    if (dbias && (e = hipMemsetAsync(dbias, 0, sizeof(float) * g.Cout, (hipStream_t)stream)) != hipSuccess) return (int)e;
    DASR_LAUNCH(k_conv3x3_c1_wgrad, dim3(1024), dim3(256), sizeof(float) * 256 * 40, stream, x, dy, yact, dw, dbias, g.B,
                g.H, g.W, g.Cout, act);
    DASR_RETURN_LAUNCH_STATUS();
}
