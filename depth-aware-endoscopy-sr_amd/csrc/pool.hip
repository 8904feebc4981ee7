// pool.hip — RegionWiseAvgPooling (sftmd_arch.py:714-733): bilinear(align_corners=True) mask resize,
// re-binarisation at 0.5, masked mean per depth region; and its backward (features only).
#include "dasr_common.h"

// maskr NCHW [B,K,h,w] from mask NCHW [B,K,H,W]
__global__ void k_pool_mask_resize(const float* __restrict__ mask, float* __restrict__ maskr, int h, int w, int H,
                                   int W, size_t n) {
    // ATen area_pixel_compute_scale(align_corners=True): (in-1)/(out-1), 0 when out == 1
    float sh = h > 1 ? (float)(H - 1) / (float)(h - 1) : 0.f;
    float sw = w > 1 ? (float)(W - 1) / (float)(w - 1) : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int x = (int)(i % w), y = (int)((i / w) % h);
        size_t bk = i / ((size_t)w * h);
        const float* m = mask + bk * (size_t)H * W;
        if (h == H && w == W) {
            maskr[i] = m[(size_t)y * W + x];
            continue;
        }
        float fy = sh * (float)y, fx = sw * (float)x;
        int y0 = (int)fy, x0 = (int)fx;
        int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        float ly1 = fminf(fmaxf(fy - (float)y0, 0.f), 1.f), lx1 = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
        float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        float top = lx0 * m[(size_t)y0 * W + x0] + lx1 * m[(size_t)y0 * W + x1];
        float bot = lx0 * m[(size_t)y1 * W + x0] + lx1 * m[(size_t)y1 * W + x1];
        float v = ly0 * top + ly1 * bot;
        maskr[i] = v >= 0.5f ? 1.f : 0.f;
    }
}
// area[b,k] = sum_p maskr ; one wave per (b,k)
__global__ void __launch_bounds__(64) k_pool_area(const float* __restrict__ maskr, float* __restrict__ area, int hw) {
    const float* m = maskr + (size_t)blockIdx.x * hw;
    float acc = 0.f;
    for (int p = threadIdx.x; p < hw; p += 64) acc += m[p];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (threadIdx.x == 0) area[blockIdx.x] = acc;
}
// out[b,k,l] = sum_p maskr[b,k,p] * feat[b,p,l] / (area[b,k] + 1e-10)
__global__ void k_pool_fwd(const float* __restrict__ feat, const float* __restrict__ maskr,
                           const float* __restrict__ area, float* __restrict__ out, int K, int L, int hw, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int l = (int)(i % L), k = (int)((i / L) % K);
        size_t b = i / ((size_t)L * K);
        const float* m = maskr + (b * K + k) * hw;
        const float* f = feat + b * (size_t)hw * L + l;
        float acc = 0.f;
        for (int p0 = 0; p0 < hw; p0 += 16) {        // sixteen (mask, feature) pairs in flight; same summation order
            float mv[16], fv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int p = p0 + u < hw ? p0 + u : hw - 1;
                mv[u] = m[p];
                fv[u] = f[(size_t)p * L];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (p0 + u < hw) acc = fmaf(mv[u], fv[u], acc);
        }
        out[i] = acc / (area[b * K + k] + 1e-10f);
    }
}
// dfeat[b,p,l] = sum_k maskr[b,k,p] * dout[b,k,l] / (area[b,k] + 1e-10)
__global__ void k_pool_bwd(const float* __restrict__ dout, const float* __restrict__ maskr,
                           const float* __restrict__ area, float* __restrict__ dfeat, int K, int L, int hw, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int l = (int)(i % L), p = (int)((i / L) % hw);
        size_t b = i / ((size_t)L * hw);
        float acc = 0.f;
        for (int k = 0; k < K; ++k)
            acc = fmaf(maskr[(b * K + k) * hw + p], dout[(b * K + k) * L + l] / (area[b * K + k] + 1e-10f), acc);
        dfeat[i] = acc;
    }
}
extern "C" int dasr_region_pool_fwd(const float* feat, const float* mask, float* maskr, float* area, float* out, int B,
                                    int K, int L, int h, int w, int H, int W, void* stream) {
    DASR_CHECK_PTR(feat); DASR_CHECK_PTR(mask); DASR_CHECK_PTR(maskr); DASR_CHECK_PTR(area); DASR_CHECK_PTR(out);
    DASR_CHECK_SHAPE(B > 0 && K > 0 && L > 0 && h > 0 && w > 0 && H > 0 && W > 0);
    size_t nm = (size_t)B * K * h * w, no = (size_t)B * K * L;
    DASR_LAUNCH(k_pool_mask_resize, dim3(dasr_ew_grid(nm)), dim3(256), 0, stream, mask, maskr, h, w, H, W, nm);
    DASR_LAUNCH(k_pool_area, dim3(B * K), dim3(64), 0, stream, maskr, area, h * w);
    DASR_LAUNCH(k_pool_fwd, dim3(dasr_ew_grid(no)), dim3(256), 0, stream, feat, maskr, area, out, K, L, h * w, no);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_region_pool_bwd(const float* dout, const float* maskr, const float* area, float* dfeat, int B,
                                    int K, int L, int h, int w, void* stream) {
    DASR_CHECK_PTR(dout); DASR_CHECK_PTR(maskr); DASR_CHECK_PTR(area); DASR_CHECK_PTR(dfeat);
    DASR_CHECK_SHAPE(B > 0 && K > 0 && L > 0 && h > 0 && w > 0);
    size_t n = (size_t)B * h * w * L;
    DASR_LAUNCH(k_pool_bwd, dim3(dasr_ew_grid(n)), dim3(256), 0, stream, dout, maskr, area, dfeat, K, L, h * w, n);
    DASR_RETURN_LAUNCH_STATUS();
}
