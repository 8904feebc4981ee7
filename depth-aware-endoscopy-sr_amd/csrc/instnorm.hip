// instnorm.hip — per-(sample, channel) mean and biased variance of an NHWC tensor.
// Reference: nn.InstanceNorm2d(affine=False) at sftmd_arch.py:811-820 and normalization.py:16-17,56.
//
// HBM-bound: ONE read of the tensor.  The image is cut into chunks of IN_CHUNK pixels; a workgroup (16 pixel lanes x
// 16 channel quads, a wave-instruction = four pixels x 64 channels contiguous) streams its chunk in batches of eight
// 16-byte loads per lane, all in flight together.  Each batch is reduced in registers to (mean, M2 about that mean) -
// two passes over eight registers, no E[x^2]-mean^2 cancellation - and folded into the lane's running pair with
// Chan's formula; the 16 pixel lanes of a channel quad are merged in a fixed order through LDS (one barrier), and a
// second small kernel merges the chunk records (16 lanes per (b, c), fixed tree), so the result is bitwise
// reproducible.  T = storage type of the tensor (float, or bf16_t on the mixed-precision path); statistics are fp32.
#include "dasr_common.h"
#include "bf16.h"

#define IN_CHUNK 1024          // pixels per workgroup (vector kernel): 64 per lane = 8 batches of 8 loads
#define IN_CHUNK_C1 256        // scalar-lane kernel (C % 4 != 0)
#define IN_NB 8

struct ChanAcc {
    float n, mean, m2;
};
// fold (nb, mb, m2b) into a
__device__ __forceinline__ void chan_merge(ChanAcc& a, float nb, float mb, float m2b) {
    const float tot = a.n + nb;
    if (tot > 0.f) {
        const float delta = mb - a.mean, r = nb / tot;
        a.mean += delta * r;
        a.m2 += m2b + delta * delta * (a.n * r);
        a.n = tot;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_instnorm_partial(const T* __restrict__ x, float* __restrict__ part, int HW,
                                                          int C, int nchunks) {
    __shared__ float red[16][16][9];               // [pixel lane][channel quad][mean x4, m2 x4, n]
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.z * 64 + 4 * cq;
    const bool live = c + 3 < C;
    const int p0 = chunk * IN_CHUNK;
    const int p1 = p0 + IN_CHUNK < HW ? p0 + IN_CHUNK : HW;
    const T* xb = x + (size_t)b * HW * C + (live ? c : 0);
    ChanAcc a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = ChanAcc{0.f, 0.f, 0.f};
    for (int pb = p0 + pl; pb < p1; pb += 16 * IN_NB) {
        float4 v[IN_NB];
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < IN_NB; ++u) {                    // clamped addresses: the loads are unconditional
            const int p = pb + 16 * u;
            v[u] = ld4(xb + (size_t)(p < p1 ? p : p1 - 1) * C);      // plain loads: the producer just wrote it (L2 / MALL hits)
            cnt += p < p1 ? 1 : 0;
        }
        const float nb = (float)cnt, inv = 1.0f / nb;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < IN_NB; ++u)
            if (u < cnt) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        const float4 mb = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < IN_NB; ++u)
            if (u < cnt) {
                const float dx = v[u].x - mb.x, dy = v[u].y - mb.y, dz = v[u].z - mb.z, dw = v[u].w - mb.w;
                q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
            }
        chan_merge(a[0], nb, mb.x, q.x);
        chan_merge(a[1], nb, mb.y, q.y);
        chan_merge(a[2], nb, mb.z, q.z);
        chan_merge(a[3], nb, mb.w, q.w);
    }
    float* r = red[pl][cq];
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = a[j].mean; r[4 + j] = a[j].m2; }
    r[8] = a[0].n;
    __syncthreads();
    if (pl < 4 && live) {                          // thread (pl = j, cq) merges channel c + j over the 16 pixel lanes
        const int j = pl;
        ChanAcc t{0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l < 16; ++l) chan_merge(t, red[l][cq][8], red[l][cq][j], red[l][cq][4 + j]);
        float* o = part + (((size_t)b * nchunks + chunk) * C + c + j) * 2;
        o[0] = t.mean;
        o[1] = t.m2;
    }
}

// scalar-lane variant for channel counts that are not a multiple of 4 (chunked two-pass)
template <typename T>
__global__ void __launch_bounds__(256) k_instnorm_partial_c1(const T* __restrict__ x, float* __restrict__ part,
                                                             int HW, int C, int nchunks) {
    __shared__ float red[256];
    __shared__ float mu_s[64];
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int c = blockIdx.z * 64 + (threadIdx.x & 63);
    const int pl = threadIdx.x >> 6;
    const int p0 = chunk * IN_CHUNK_C1;
    const int p1 = p0 + IN_CHUNK_C1 < HW ? p0 + IN_CHUNK_C1 : HW;
    const T* xb = x + (size_t)b * HW * C;
    float acc = 0.f;
    if (c < C)
        for (int p = p0 + pl; p < p1; p += 4) acc += ld1(xb + (size_t)p * C + c);
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
            float d = ld1(xb + (size_t)p * C + c) - mu;
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

// 16 lanes per (b, c): lane j folds chunks j, j+16, ... in order, then a fixed xor tree merges the 16 lanes
// (the 4-workgroup serial recurrence this replaces took 100 us inside the training step)
__global__ void __launch_bounds__(64) k_instnorm_merge(const float* __restrict__ part, float* __restrict__ mean,
                                                       float* __restrict__ var, int HW, int C, int nchunks, int chunk_px,
                                                       int n) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 4), j = threadIdx.x & 15;
    const int ii = i < n ? i : n - 1;              // every lane takes part in the shuffles
    const int c = ii % C, b = ii / C;
    ChanAcc a{0.f, 0.f, 0.f};
    for (int k = j; k < nchunks; k += 16) {
        const float2 rec = *(const float2*)(part + (((size_t)b * nchunks + k) * C + c) * 2);
        const int p0 = k * chunk_px;
        chan_merge(a, (float)((p0 + chunk_px < HW ? p0 + chunk_px : HW) - p0), rec.x, rec.y);
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        const float nb = __shfl_xor(a.n, off, 64), mb = __shfl_xor(a.mean, off, 64), m2b = __shfl_xor(a.m2, off, 64);
        // both partners must compute the same bits: merge (lower lane's record, upper lane's record) in that order
        ChanAcc lo = (j & off) ? ChanAcc{nb, mb, m2b} : a;
        const ChanAcc hi = (j & off) ? a : ChanAcc{nb, mb, m2b};
        chan_merge(lo, hi.n, hi.mean, hi.m2);
        a = lo;
    }
    if (j == 0 && i < n) {
        mean[i] = a.mean;
        var[i] = a.m2 / (float)HW;
    }
}

static int in_chunk_px(int C) { return (C & 3) == 0 ? IN_CHUNK : IN_CHUNK_C1; }

extern "C" size_t dasr_instnorm_stats_workspace(int B, int HW, int C) {
    if (B <= 0 || HW <= 0 || C <= 0) return 0;
    const int ch = in_chunk_px(C);
    size_t nchunks = ((size_t)HW + ch - 1) / ch;
    return sizeof(float) * 2 * (size_t)B * nchunks * C;
}

template <typename T>
static int instnorm_stats_impl(const T* x, float* mean, float* var, void* workspace, size_t workspace_bytes, int B, int HW,
                               int C, void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(mean); DASR_CHECK_PTR(var); DASR_CHECK_PTR(workspace);
    DASR_CHECK_SHAPE(B > 0 && HW > 0 && C > 0);
    if (workspace_bytes < dasr_instnorm_stats_workspace(B, HW, C)) return DASR_E_WORKSPACE;
    const int ch = in_chunk_px(C);
    int nchunks = (HW + ch - 1) / ch;
    float* part = (float*)workspace;
    if ((C & 3) == 0) {
        DASR_LAUNCH((k_instnorm_partial<T>), dim3(nchunks, B, dasr_cdiv(C, 64)), dim3(256), 0, stream, x, part, HW, C,
                    nchunks);
    } else {
        DASR_LAUNCH((k_instnorm_partial_c1<T>), dim3(nchunks, B, dasr_cdiv(C, 64)), dim3(256), 0, stream, x, part, HW, C,
                    nchunks);
    }
    int n = B * C;
    DASR_LAUNCH(k_instnorm_merge, dim3(dasr_cdiv(n, 4)), dim3(64), 0, stream, (const float*)part, mean, var, HW, C, nchunks,
                ch, n);
    DASR_RETURN_LAUNCH_STATUS();
}

extern "C" int dasr_instnorm_stats(const float* x, float* mean, float* var, void* workspace, size_t workspace_bytes,
                                   int B, int HW, int C, void* stream) {
    return instnorm_stats_impl<float>(x, mean, var, workspace, workspace_bytes, B, HW, C, stream);
}
extern "C" int dasr_instnorm_stats_bf16(const unsigned short* x, float* mean, float* var, void* workspace,
                                        size_t workspace_bytes, int B, int HW, int C, void* stream) {
    return instnorm_stats_impl<bf16_t>((const bf16_t*)x, mean, var, workspace, workspace_bytes, B, HW, C, stream);
}
