// loss.hip — one-pass sums for the harness losses: the L1 pixel loss and the per-region
// smooth-L1 numerators / areas of dynamic_weight_mask_loss (codes/models/modules/mask_loss.py:64-90,
// F_model_depthCond.py:164,188-190).  The reference makes 10 masked passes over two [B,3,sH,sW] tensors; with
// one-hot depth masks every HR pixel belongs to at most one region (nearest upsampling of the mask:
// region[y/s][x/s]), so one pass yields all K numerators, the K areas and sum|sr-hr|; the division, the softmax
// weighting and the trainable weights stay in PyTorch (harness.py).  HBM-bound: 24 B per HR pixel forward,
// 36 B backward.  sr / hr are the API's NCHW tensors.
#include "dasr_common.h"

#define LOSS_MAXK 16

// sums: [0..K-1] numerators  sum smooth_l1(sr-hr) over the region (all channels)
//       [K..2K-1] areas       C * (#pixels of the region)          (= sum of the 3-channel mask, mask_loss.py:81)
//       [2K]      sum |sr - hr|
__global__ void __launch_bounds__(256) k_loss_sums(const float* __restrict__ sr, const float* __restrict__ hr,
                                                   const unsigned char* __restrict__ region, float* __restrict__ sums,
                                                   int B, int C, int h, int w, int s, int K) {
    __shared__ float red[2 * LOSS_MAXK + 2];
    for (int i = threadIdx.x; i < 2 * K + 1; i += 256) red[i] = 0.f;
    __syncthreads();
    const int W = w * s, H = h * s;
    // one thread = one LR-pixel-wide run of s HR pixels in one row of one channel: a single region
    const size_t nruns = (size_t)B * C * H * w;
    float l1 = 0.f;
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < nruns; r += (size_t)gridDim.x * 256) {
        const int lx = (int)(r % w);
        const int Y = (int)((r / w) % H);
        const size_t bc = r / ((size_t)w * H);
        const int b = (int)(bc / C);
        const int k = region[((size_t)b * h + Y / s) * w + lx];
        const float* ps = sr + (bc * H + Y) * W + (size_t)lx * s;
        const float* ph = hr + (bc * H + Y) * W + (size_t)lx * s;
        float num = 0.f;
        for (int j = 0; j < s; ++j) {
            const float d = ps[j] - ph[j], ad = fabsf(d);
            l1 += ad;
            num += ad < 1.f ? 0.5f * d * d : ad - 0.5f;       // SmoothL1Loss(beta = 1)
        }
        if (k < K) {
            atomicAdd(&red[k], num);
            atomicAdd(&red[K + k], (float)s);
        }
    }
    for (int off = 32; off > 0; off >>= 1) l1 += __shfl_down(l1, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&red[2 * K], l1);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * K + 1; i += 256) atomicAdd(&sums[i], red[i]);
}

// dsr = dsums[2K] * sign(d) + dsums[region] * smooth_l1'(d)
__global__ void __launch_bounds__(256) k_loss_bwd(const float* __restrict__ sr, const float* __restrict__ hr,
                                                  const unsigned char* __restrict__ region,
                                                  const float* __restrict__ dsums, float* __restrict__ dsr, int B,
                                                  int C, int h, int w, int s, int K) {
    const int W = w * s, H = h * s;
    const size_t n = (size_t)B * C * H * W;
    const float dl1 = dsums[2 * K];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int X = (int)(i % W), Y = (int)((i / W) % H);
        const int b = (int)(i / ((size_t)W * H * C));
        const int k = region[((size_t)b * h + Y / s) * w + X / s];
        const float d = sr[i] - hr[i];
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        float g = dl1 * sg;
        if (k < K) g += dsums[k] * (fabsf(d) < 1.f ? d : sg);
        dsr[i] = g;
    }
}

extern "C" int dasr_loss_sums(const float* sr, const float* hr, const unsigned char* region, float* sums, int B, int C,
                              int h, int w, int scale, int K, void* stream) {
    DASR_CHECK_PTR(sr); DASR_CHECK_PTR(hr); DASR_CHECK_PTR(region); DASR_CHECK_PTR(sums);
    DASR_CHECK_SHAPE(B > 0 && C > 0 && h > 0 && w > 0 && scale > 0 && K > 0);
    if (K > LOSS_MAXK) return DASR_E_UNSUPPORTED;
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(float) * (2 * K + 1), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    size_t nruns = (size_t)B * C * h * scale * w;
    DASR_LAUNCH(k_loss_sums, dim3(dasr_ew_grid(nruns)), dim3(256), 0, stream, sr, hr, region, sums, B, C, h, w, scale, K);
    DASR_RETURN_LAUNCH_STATUS();
}

extern "C" int dasr_loss_bwd(const float* sr, const float* hr, const unsigned char* region, const float* dsums,
                             float* dsr, int B, int C, int h, int w, int scale, int K, void* stream) {
    DASR_CHECK_PTR(sr); DASR_CHECK_PTR(hr); DASR_CHECK_PTR(region); DASR_CHECK_PTR(dsums); DASR_CHECK_PTR(dsr);
    DASR_CHECK_SHAPE(B > 0 && C > 0 && h > 0 && w > 0 && scale > 0 && K > 0);
    if (K > LOSS_MAXK) return DASR_E_UNSUPPORTED;
    size_t n = (size_t)B * C * h * scale * w * scale;
    DASR_LAUNCH(k_loss_bwd, dim3(dasr_ew_grid(n)), dim3(256), 0, stream, sr, hr, region, dsums, dsr, B, C, h, w, scale, K);
    DASR_RETURN_LAUNCH_STATUS();
}
