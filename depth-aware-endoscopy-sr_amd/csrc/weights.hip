// weights.hip — weight_norm (dim 0) + packing of PyTorch-layout kernels into HWIO, and its backward.
// Reference: torch.nn.utils.weight_norm call sites sftmd_arch.py:741,851 (w = g*v/||v||, norm over all
// dims but 0; for ConvTranspose2d dim 0 is the in-channel axis).  Tiny tensors: one workgroup per norm
// group, wave shuffles + LDS for the reduction.
#include "dasr_common.h"
#include "bf16.h"

__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats of LDS */) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();  // protect `red` against the previous use
    if (lane == 0) red[wv] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// element (n, r) of a norm group: n = dim-0 index, r = remaining flat index in [0, R)
// conv:        v[o][i][kh][kw]   n=o, r = i*KK + t      -> w[(t*I + i)*O + o]
// transposed:  v[i][o][kh][kw]   n=i, r = o*KK + t      -> w[(t*I + i)*O + o]
__device__ __forceinline__ size_t hwio_index(int n, int r, int ldo, int o_off, int I, int KK, int transposed) {
    int a = r / KK, t = r % KK;
    int o = transposed ? a : n;
    int i = transposed ? n : a;
    return ((size_t)t * I + i) * ldo + o_off + o;
}

__device__ __forceinline__ float block_max_256(float v, float* red /* >= 4 floats of LDS */) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// One norm group n (a workgroup).  TW = element type of the packed kernel: float, or bf16_t for the bf16 trunk convolutions
// (fp32 master weights, the normalisation is done in fp32 and rounded once).  amax (optional, dasr_common.h amax buffer of
// the packed kernel): this group's max |w| goes to part o_off + n, the part count ldo is written by group 0 of offset 0.
// plain: only the HWIO half is written (a bias vector seen as an [O][1][1][1] kernel).
template <typename TW>
__device__ __forceinline__ void weight_pack_group(const float* __restrict__ v, const float* __restrict__ g, TW* __restrict__ w,
                                                  float* __restrict__ inv_norm, float* __restrict__ amax, int n, int O, int I,
                                                  int KK, int transposed, int ldo, int o_off, int plain, float* red) {
    int R = (transposed ? O : I) * KK;
    const float* vn = v + (size_t)n * R;
    float scale = 1.f;
    if (g) {
        float ss = 0.f;
        for (int r = threadIdx.x; r < R; r += 256) ss += vn[r] * vn[r];
        ss = block_sum_256(ss, red);
        float inv = 1.0f / sqrtf(ss);
        if (threadIdx.x == 0 && inv_norm) inv_norm[n] = inv;
        scale = g[n] * inv;
    }
    // second half of the packed buffer: the same kernel with each tap transposed, [tap][ldo][I]
    TW* wT = w + (size_t)KK * I * ldo;
    float m = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) {
        const float val = vn[r] * scale;
        m = fmaxf(m, fabsf(val));
        st1(w + hwio_index(n, r, ldo, o_off, I, KK, transposed), val);
        if (plain) continue;
        int a = r / KK, t = r % KK;
        int o = transposed ? a : n, i = transposed ? n : a;
        st1(wT + ((size_t)t * ldo + o_off + o) * I + i, val);
    }
    if (amax) {
        m = block_max_256(m, red);
        if (threadIdx.x == 0) {
            amax[1 + o_off + n] = m;
            if (o_off + n == 0) ((int*)amax)[0] = ldo;
        }
    }
}

template <typename TW>
__global__ void __launch_bounds__(256) k_weight_pack_fwd(const float* __restrict__ v, const float* __restrict__ g,
                                                         TW* __restrict__ w, float* __restrict__ inv_norm, int O,
                                                         int I, int KK, int transposed, int ldo, int o_off) {
    __shared__ float red[4];
    weight_pack_group<TW>(v, g, w, inv_norm, nullptr, blockIdx.x, O, I, KK, transposed, ldo, o_off, 0, red);
}

// Every kernel of a network in ONE launch: a table of jobs in device memory (dasr.h: dasr_pack_job), one workgroup per norm
// group of every job (wg_begin = the job's first workgroup; the table is sorted by it).
__global__ void __launch_bounds__(256) k_weight_pack_multi(const dasr_pack_job* __restrict__ jobs, int njobs) {
    __shared__ float red[4];
    int lo = 0, hi = njobs - 1;                  // the last job whose wg_begin <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].wg_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const dasr_pack_job J = jobs[lo];
    const int n = blockIdx.x - J.wg_begin;
    if (J.bf16) weight_pack_group<bf16_t>(J.v, J.g, (bf16_t*)J.w, J.inv_norm, J.amax, n, J.O, J.I, J.KK, J.transposed, J.ldo, J.o_off, J.plain, red);
    else        weight_pack_group<float>(J.v, J.g, (float*)J.w, J.inv_norm, J.amax, n, J.O, J.I, J.KK, J.transposed, J.ldo, J.o_off, J.plain, red);
}

// dv = (g/||v||) * (dw - v * <dw,v>/||v||^2) ; dg = <dw,v>/||v||
// One workgroup per norm group.  NR = register slots per thread for the group's gathered dw (R <= 256 * NR): all loads of the
// strided gather are issued before the first use and kept for the second pass (one latency instead of 2 * R / 256 of them);
// NR = 0: any R, two passes over memory.
template <int NR>
__global__ void __launch_bounds__(256) k_weight_pack_bwd(const float* __restrict__ dw, const float* __restrict__ v,
                                                         const float* __restrict__ g,
                                                         const float* __restrict__ inv_norm, float* __restrict__ dv,
                                                         float* __restrict__ dg, int O, int I, int KK, int transposed, int ldo,
                                                         int o_off) {
    __shared__ float red[4];
    int n = blockIdx.x;
    int R = (transposed ? O : I) * KK;
    const float* vn = v + (size_t)n * R;
    float* dvn = dv + (size_t)n * R;
    if (NR > 0) {
        float d[NR > 0 ? NR : 1], vv[NR > 0 ? NR : 1];
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int r = threadIdx.x + 256 * u;
            d[u] = r < R ? dw[hwio_index(n, r, ldo, o_off, I, KK, transposed)] : 0.f;
            vv[u] = r < R && g ? vn[r] : 0.f;
        }
        if (!g) {
#pragma unroll
            for (int u = 0; u < NR; ++u)
                if (threadIdx.x + 256 * u < R) dvn[threadIdx.x + 256 * u] = d[u];
            return;
        }
        float dot = 0.f;
#pragma unroll
        for (int u = 0; u < NR; ++u) dot += d[u] * vv[u];
        dot = block_sum_256(dot, red);
        const float inv = inv_norm[n], gn = g[n];
        if (threadIdx.x == 0) dg[n] = dot * inv;
        const float c1 = gn * inv, c2 = gn * dot * inv * inv * inv;
#pragma unroll
        for (int u = 0; u < NR; ++u)
            if (threadIdx.x + 256 * u < R) dvn[threadIdx.x + 256 * u] = c1 * d[u] - c2 * vv[u];
        return;
    }
    if (!g) {
        for (int r = threadIdx.x; r < R; r += 256) dvn[r] = dw[hwio_index(n, r, ldo, o_off, I, KK, transposed)];
        return;
    }
    float dot = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) dot += dw[hwio_index(n, r, ldo, o_off, I, KK, transposed)] * vn[r];
    dot = block_sum_256(dot, red);
    float inv = inv_norm[n];
    float gn = g[n];
    if (threadIdx.x == 0) dg[n] = dot * inv;
    float c1 = gn * inv, c2 = gn * dot * inv * inv * inv;
    for (int r = threadIdx.x; r < R; r += 256) dvn[r] = c1 * dw[hwio_index(n, r, ldo, o_off, I, KK, transposed)] - c2 * vn[r];
}

extern "C" int dasr_weight_pack_fwd(const float* v, const float* g, float* w, float* inv_norm, int O, int I, int KH,
                                    int KW, int transposed, int ldo, int o_off, void* stream) {
    DASR_CHECK_PTR(v); DASR_CHECK_PTR(w);
    if (g) DASR_CHECK_PTR(inv_norm);
    DASR_CHECK_SHAPE(O > 0 && I > 0 && KH > 0 && KW > 0 && o_off >= 0 && ldo >= o_off + O);
    int groups = transposed ? I : O;
    DASR_LAUNCH((k_weight_pack_fwd<float>), dim3(groups), dim3(256), 0, stream, v, g, w, inv_norm, O, I, KH * KW, transposed,
                ldo, o_off);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_weight_pack_fwd_bf16(const float* v, const float* g, unsigned short* w, float* inv_norm, int O, int I,
                                         int KH, int KW, int transposed, int ldo, int o_off, void* stream) {
    DASR_CHECK_PTR(v); DASR_CHECK_PTR(w);
    if (g) DASR_CHECK_PTR(inv_norm);
    DASR_CHECK_SHAPE(O > 0 && I > 0 && KH > 0 && KW > 0 && o_off >= 0 && ldo >= o_off + O);
    int groups = transposed ? I : O;
    DASR_LAUNCH((k_weight_pack_fwd<bf16_t>), dim3(groups), dim3(256), 0, stream, v, g, (bf16_t*)w, inv_norm, O, I, KH * KW,
                transposed, ldo, o_off);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_weight_pack_bwd(const float* dw, const float* v, const float* g, const float* inv_norm, float* dv,
                                    float* dg, int O, int I, int KH, int KW, int transposed, int ldo, int o_off,
                                    void* stream) {
    DASR_CHECK_PTR(dw); DASR_CHECK_PTR(v); DASR_CHECK_PTR(dv);
    if (g) { DASR_CHECK_PTR(inv_norm); DASR_CHECK_PTR(dg); }
    DASR_CHECK_SHAPE(O > 0 && I > 0 && KH > 0 && KW > 0 && o_off >= 0 && ldo >= o_off + O);
    int groups = transposed ? I : O;
    const int R = (transposed ? O : I) * KH * KW;
    if (R <= 256 * 5)       DASR_LAUNCH((k_weight_pack_bwd<5>), dim3(groups), dim3(256), 0, stream, dw, v, g, inv_norm, dv, dg, O, I,
                                        KH * KW, transposed, ldo, o_off);      // (up to 128 channels x 3x3 = 1152)
    else if (R <= 256 * 11) DASR_LAUNCH((k_weight_pack_bwd<11>), dim3(groups), dim3(256), 0, stream, dw, v, g, inv_norm, dv, dg, O, I,
                                        KH * KW, transposed, ldo, o_off);      // (32 channels x 9x9 = 2592)
    else                    DASR_LAUNCH((k_weight_pack_bwd<0>), dim3(groups), dim3(256), 0, stream, dw, v, g, inv_norm, dv, dg, O, I,
                                        KH * KW, transposed, ldo, o_off);
    DASR_RETURN_LAUNCH_STATUS();
}

// jobs_host: the same table in host memory (validated here; the launch reads jobs_device)
extern "C" int dasr_weight_pack_multi(const dasr_pack_job* jobs_host, const dasr_pack_job* jobs_device, int njobs, void* stream) {
    DASR_CHECK_PTR(jobs_host); DASR_CHECK_PTR(jobs_device);
    DASR_CHECK_SHAPE(njobs > 0);
    long long total = 0;
    for (int j = 0; j < njobs; ++j) {
        const dasr_pack_job& J = jobs_host[j];
        DASR_CHECK_PTR(J.v); DASR_CHECK_PTR(J.w);
        if (J.g) DASR_CHECK_PTR(J.inv_norm);
        DASR_CHECK_SHAPE(J.O > 0 && J.I > 0 && J.KK > 0 && J.o_off >= 0 && J.ldo >= J.o_off + J.O);
        DASR_CHECK_SHAPE(J.wg_begin == total);
        DASR_CHECK_SHAPE(!J.amax || (J.ldo <= DASR_AMAX_MAX_PARTS && !J.transposed));
        total += J.transposed ? J.I : J.O;
        DASR_CHECK_SHAPE(total < (1ll << 30));
    }
    DASR_LAUNCH(k_weight_pack_multi, dim3((unsigned)total), dim3(256), 0, stream, jobs_device, njobs);
    DASR_RETURN_LAUNCH_STATUS();
}
