// instnorm.hip — per-(sample, channel) mean and biased variance of an NHWC tensor.
// Reference: nn.InstanceNorm2d(affine=False) at sftmd_arch.py:811-820 and normalization.py:16-17,56.
//
// HBM-bound (one read of the tensor; the second pass of each chunk hits L1/L2).  The image is cut into
// chunks of IN_CHUNK pixels; a workgroup (4 pixel lanes x 64 channel lanes, 256-byte coalesced rows)
// computes the chunk's mean and its sum of squared deviations ABOUT THAT MEAN (no E[x^2]-mean^2
// cancellation), and a second tiny kernel merges the chunks in a fixed order with Chan's formula, so
// the result is bitwise reproducible.
#include "dasr_common.h"

#define IN_CHUNK 256

// 256 threads = 16 pixel lanes x 16 channel quads: a lane loads float4 (4 channels), a wave-instruction covers four
// pixels x 256 B.  blockIdx.z selects a 64-channel slice.
__global__ void __launch_bounds__(256) k_instnorm_partial(const float* __restrict__ x, float* __restrict__ part, int HW,
                                                          int C, int nchunks) {
    __shared__ float4 red[256];
    __shared__ float4 mu_s[16];
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.z * 64 + 4 * cq;
    const bool live = c + 3 < C;
    const int p0 = chunk * IN_CHUNK;
    const int p1 = p0 + IN_CHUNK < HW ? p0 + IN_CHUNK : HW;
    const float* xb = x + (size_t)b * HW * C + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live)
        for (int p = p0 + pl; p < p1; p += 16) {
            const float4 v = *(const float4*)(xb + (size_t)p * C);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (pl == 0) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l = 0; l < 16; ++l) {
            const float4 v = red[l * 16 + cq];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const float inv = 1.0f / (float)(p1 - p0);
        mu_s[cq] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
    }
    __syncthreads();
    const float4 mu = mu_s[cq];
    acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live)
        for (int p = p0 + pl; p < p1; p += 16) {
            const float4 v = *(const float4*)(xb + (size_t)p * C);
            const float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
            acc.x = fmaf(dx, dx, acc.x); acc.y = fmaf(dy, dy, acc.y);
            acc.z = fmaf(dz, dz, acc.z); acc.w = fmaf(dw, dw, acc.w);
        }
    __syncthreads();
    red[threadIdx.x] = acc;
    __syncthreads();
    if (pl == 0 && live) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l = 0; l < 16; ++l) {
            const float4 v = red[l * 16 + cq];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float* o = part + (((size_t)b * nchunks + chunk) * C + c) * 2;
        o[0] = mu.x; o[1] = s.x; o[2] = mu.y; o[3] = s.y; o[4] = mu.z; o[5] = s.z; o[6] = mu.w; o[7] = s.w;
    }
}

// scalar-lane variant for channel counts that are not a multiple of 4
__global__ void __launch_bounds__(256) k_instnorm_partial_c1(const float* __restrict__ x, float* __restrict__ part,
                                                             int HW, int C, int nchunks) {
    __shared__ float red[256];
    __shared__ float mu_s[64];
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int c = blockIdx.z * 64 + (threadIdx.x & 63);
    const int pl = threadIdx.x >> 6;
    const int p0 = chunk * IN_CHUNK;
    const int p1 = p0 + IN_CHUNK < HW ? p0 + IN_CHUNK : HW;
    const float* xb = x + (size_t)b * HW * C;
    float acc = 0.f;
    if (c < C)
        for (int p = p0 + pl; p < p1; p += 4) acc += xb[(size_t)p * C + c];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (pl == 0)
        mu_s[threadIdx.x] = (red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192]) /
                            (float)(p1 - p0);
    __syncthreads();
    const float mu = mu_s[threadIdx.x & 63];
    acc = 0.f;
    if (c < C)
        for (int p = p0 + pl; p < p1; p += 4) {
            float d = xb[(size_t)p * C + c] - mu;
            acc = fmaf(d, d, acc);
        }
    __syncthreads();
    red[threadIdx.x] = acc;
    __syncthreads();
    if (pl == 0 && c < C) {
        float m2 = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
        size_t o = (((size_t)b * nchunks + chunk) * C + c) * 2;
        part[o] = mu;
        part[o + 1] = m2;
    }
}

__global__ void k_instnorm_merge(const float* __restrict__ part, float* __restrict__ mean, float* __restrict__ var,
                                 int HW, int C, int nchunks, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = i % C, b = i / C;
    float cnt = 0.f, mu = 0.f, m2 = 0.f;
    // the merge is a serial recurrence, the loads are not: sixteen chunk records are fetched per round trip
    // (one load per iteration made this 1024-thread kernel an 80-deep chain of L2 latencies: 60 us)
    for (int k0 = 0; k0 < nchunks; k0 += 16) {
        float2 rec[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + u < nchunks ? k0 + u : nchunks - 1;
            rec[u] = *(const float2*)(part + (((size_t)b * nchunks + k) * C + c) * 2);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + u;
            if (k >= nchunks) break;
            int p0 = k * IN_CHUNK;
            float nb = (float)((p0 + IN_CHUNK < HW ? p0 + IN_CHUNK : HW) - p0);
            float mb = rec[u].x, m2b = rec[u].y;
            float tot = cnt + nb, delta = mb - mu;
            mu += delta * (nb / tot);
            m2 += m2b + delta * delta * (cnt * nb / tot);
            cnt = tot;
        }
    }
    mean[i] = mu;
    var[i] = m2 / (float)HW;
}

extern "C" size_t dasr_instnorm_stats_workspace(int B, int HW, int C) {
    if (B <= 0 || HW <= 0 || C <= 0) return 0;
    size_t nchunks = ((size_t)HW + IN_CHUNK - 1) / IN_CHUNK;
    return sizeof(float) * 2 * (size_t)B * nchunks * C;
}

extern "C" int dasr_instnorm_stats(const float* x, float* mean, float* var, void* workspace, size_t workspace_bytes,
                                   int B, int HW, int C, void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(mean); DASR_CHECK_PTR(var); DASR_CHECK_PTR(workspace);
    DASR_CHECK_SHAPE(B > 0 && HW > 0 && C > 0);
    if (workspace_bytes < dasr_instnorm_stats_workspace(B, HW, C)) return DASR_E_WORKSPACE;
    int nchunks = (HW + IN_CHUNK - 1) / IN_CHUNK;
    float* part = (float*)workspace;
    if ((C & 3) == 0) {
        DASR_LAUNCH(k_instnorm_partial, dim3(nchunks, B, dasr_cdiv(C, 64)), dim3(256), 0, stream, x, part, HW, C, nchunks);
    } else {
        DASR_LAUNCH(k_instnorm_partial_c1, dim3(nchunks, B, dasr_cdiv(C, 64)), dim3(256), 0, stream, x, part, HW, C,
                    nchunks);
    }
    int n = B * C;
    DASR_LAUNCH(k_instnorm_merge, dim3(dasr_cdiv(n, 256)), dim3(256), 0, stream, part, mean, var, HW, C, nchunks, n);
    DASR_RETURN_LAUNCH_STATUS();
}
