// instnorm.hip — per-(sample, channel) mean and biased variance of an NHWC tensor.
// Reference: nn.InstanceNorm2d(affine=False) at sftmd_arch.py:811-820 and normalization.py:16-17,56.
// Two passes over the (L2/MALL-resident) sample keep the variance free of E[x^2]-mean^2 cancellation.
#include "dasr_common.h"

__global__ void __launch_bounds__(256) k_instnorm_stats(const float* __restrict__ x, float* __restrict__ mean,
                                                        float* __restrict__ var, int HW, int C) {
    __shared__ float red[256];
    __shared__ float mu_s[64];
    int b = blockIdx.x;
    int c = blockIdx.y * 64 + (threadIdx.x & 63);
    int pl = threadIdx.x >> 6;
    const float* xb = x + (size_t)b * HW * C;
    float acc = 0.f;
    if (c < C)
        for (int p = pl; p < HW; p += 4) acc += xb[(size_t)p * C + c];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (pl == 0)
        mu_s[threadIdx.x] = (red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192]) /
                            (float)HW;
    __syncthreads();
    float mu = mu_s[threadIdx.x & 63];
    acc = 0.f;
    if (c < C)
        for (int p = pl; p < HW; p += 4) {
            float d = xb[(size_t)p * C + c] - mu;
            acc = fmaf(d, d, acc);
        }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (pl == 0 && c < C) {
        mean[(size_t)b * C + c] = mu;
        var[(size_t)b * C + c] =
            (red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192]) / (float)HW;
    }
}

extern "C" int dasr_instnorm_stats(const float* x, float* mean, float* var, int B, int HW, int C, void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(mean); DASR_CHECK_PTR(var);
    DASR_CHECK_SHAPE(B > 0 && HW > 0 && C > 0);
    DASR_LAUNCH(k_instnorm_stats, dim3(B, dasr_cdiv(C, 64)), dim3(256), 0, stream, x, mean, var, HW, C);
    DASR_RETURN_LAUNCH_STATUS();
}
