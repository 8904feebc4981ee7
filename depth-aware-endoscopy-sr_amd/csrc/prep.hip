// prep.hip — device-side input preparation: the reference's getDepthMask binning
// (codes/data/LQGTker_Depth_dataset.py:204-225) from the depth map straight to the region byte the one-hot SEAN
// kernels read, and optionally to the K float planes the module API takes.  Only the depth map crosses PCIe.
//
// Rule restated (float32 arithmetic exactly as torch evaluates it on 0-dim float32 tensors, no FMA contraction):
//   lo, hi   = min, max of the sample's depth map            (depthFixedRange: lo = 0, hi = 1)
//   interval = (hi - lo) / K
//   bin i    = [lo + interval*i, lo + interval*(i+1))         i = 0..K-1
// A pixel equal to the maximum (or outside [0,1) in fixed-range mode) belongs to no bin: region byte = K, all planes 0.
// In fixed-range mode the reference computes the edges in Python doubles (0 + 0.1*i) and torch compares the float32
// map against them rounded to float32; the edges are passed in from the host for that mode.
#include "dasr_common.h"

#if DASR_DEVICE_BUILD
__device__ __forceinline__ float prep_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float prep_add(float a, float b) { return __fadd_rn(a, b); }
#else
static inline float prep_mul(float a, float b) { volatile float r = a * b; return r; }
static inline float prep_add(float a, float b) { volatile float r = a + b; return r; }
#endif

#define PREP_MAXK 16
#define PREP_CHUNK 4096

// partial min / max of chunk blockIdx.x of sample blockIdx.y -> ws[(b*nchunk + chunk)*2 + {0,1}]
__global__ void __launch_bounds__(256) k_depth_minmax(const float* __restrict__ depth, float* __restrict__ ws, int HW,
                                                      int nchunk) {
    __shared__ float slo[4], shi[4];
    const int b = blockIdx.y;
    const float* d = depth + (size_t)b * HW;
    const int per = (HW + nchunk - 1) / nchunk;
    const int p0 = blockIdx.x * per, p1 = p0 + per < HW ? p0 + per : HW;
    float lo = INFINITY, hi = -INFINITY;
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        const float v = d[p];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_down(lo, off, 64));
        hi = fmaxf(hi, __shfl_down(hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[((size_t)b * nchunk + blockIdx.x) * 2 + 0] = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
        ws[((size_t)b * nchunk + blockIdx.x) * 2 + 1] = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
    }
}

// region byte (and the K float planes) of every pixel
__global__ void __launch_bounds__(256) k_depth_bins(const float* __restrict__ depth, const float* __restrict__ ws,
                                                    const float* __restrict__ fixed_edges, float* __restrict__ masks,
                                                    unsigned char* __restrict__ region, int HW, int K, int nchunk) {
    __shared__ float edge[PREP_MAXK + 1];
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        if (fixed_edges) {
            for (int i = 0; i <= K; ++i) edge[i] = fixed_edges[i];
        } else {
            float lo = INFINITY, hi = -INFINITY;
            for (int c = 0; c < nchunk; ++c) {
                lo = fminf(lo, ws[((size_t)b * nchunk + c) * 2 + 0]);
                hi = fmaxf(hi, ws[((size_t)b * nchunk + c) * 2 + 1]);
            }
            const float interval = prep_add(hi, -lo) / (float)K;          // IEEE division, like torch
            for (int i = 0; i <= K; ++i) edge[i] = prep_add(lo, prep_mul(interval, (float)i));
        }
    }
    __syncthreads();
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const float v = depth[(size_t)b * HW + p];
        int idx = K;
        for (int i = 0; i < K; ++i)
            if (v >= edge[i] && v < edge[i + 1]) idx = i;
        if (region) region[(size_t)b * HW + p] = (unsigned char)idx;
        if (masks)
            for (int i = 0; i < K; ++i) masks[((size_t)b * K + i) * HW + p] = i == idx ? 1.f : 0.f;
    }
}

static int prep_nchunk(int HW) {
    int n = (HW + PREP_CHUNK - 1) / PREP_CHUNK;
    return n > 64 ? 64 : (n < 1 ? 1 : n);
}

extern "C" size_t dasr_depth_to_masks_workspace(int B, int HW) {
    if (B <= 0 || HW <= 0) return 0;
    return sizeof(float) * 2 * (size_t)B * prep_nchunk(HW);
}

extern "C" int dasr_depth_to_masks(const float* depth, const float* fixed_edges, float* masks, unsigned char* region,
                                   void* workspace, size_t workspace_bytes, int B, int HW, int K, void* stream) {
    DASR_CHECK_PTR(depth); DASR_CHECK_PTR(workspace);
    if (masks == nullptr && region == nullptr) return DASR_E_NULL;
    DASR_CHECK_SHAPE(B > 0 && HW > 0 && K > 0);
    if (K > PREP_MAXK) return DASR_E_UNSUPPORTED;
    if (workspace_bytes < dasr_depth_to_masks_workspace(B, HW)) return DASR_E_WORKSPACE;
    const int nchunk = prep_nchunk(HW);
    if (!fixed_edges)
        DASR_LAUNCH(k_depth_minmax, dim3(nchunk, B), dim3(256), 0, stream, depth, (float*)workspace, HW, nchunk);
    unsigned gx = dasr_cdiv((size_t)HW, 256);
    if (gx > 1024) gx = 1024;
    DASR_LAUNCH(k_depth_bins, dim3(gx, B), dim3(256), 0, stream, depth, (const float*)workspace, fixed_edges, masks, region,
                HW, K, nchunk);
    DASR_RETURN_LAUNCH_STATUS();
}
