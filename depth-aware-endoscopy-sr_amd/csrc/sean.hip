// sean.hip — the Depth-Guided Block's dynamic convolution + DFN modulation, forward and backward.
//
// Reference per SEAN call (normalization.py:56,59,80-89) plus the ReLU / residual of
// Depth_Residual_Block_Mask.forward (sftmd_arch.py:826-834):
//     xhat   = IN(IN(t))                       -> one per-(b,c) scale/shift (dasr_double_in_scale)
//     gamma1 = conv3x3(style_map; W_gamma_s)   -> 3x3 conv of the K-channel MASK with the per-sample
//     beta1  = conv3x3(style_map; W_beta_s)       kernel D[b] built by dasr_dynk_fwd (no 256-ch style map)
//     out    = xhat*(1 + a_g*gamma1 + (1-a_g)*gamma2) + a_b*beta1 + (1-a_b)*beta2   (+residual) (ReLU)
//
// HBM-bound: algorithmic traffic per pixel is 3 reads + 1 write of C floats + K mask floats
// (1064 B/px at C=64, K=10).  Layout: activations NHWC, mask NCHW as delivered.
//
// Forward kernel: one workgroup = TILE_H x TILE_W pixels x 64 channels.  The mask tile with its
// 1-pixel halo and the sample's dynamic kernels D[b] (2*9*K*64 floats = 46 KB at K=10) are staged in
// LDS once per workgroup; each thread owns one channel lane and walks the tile's pixels, so t / gb2 /
// out accesses are 256-byte coalesced rows and the LDS reads of D are conflict-free (lane = channel).
#include "dasr_common.h"
#include "bf16.h"
#include "conv_kernels.h"

#define SEAN_TH 8
#define SEAN_TW 32
#define SEAN_MAXK 16

struct SeanGeom {
    int B, H, W, C, K;
};

// LDS carve (floats): D tile [2][9][K][64] | mask tile [K][(TH+2)][(TW+2)]
__device__ __forceinline__ int sean_lds_D_floats(int K) { return 2 * 9 * K * 64; }
__device__ __forceinline__ int sean_lds_M_floats(int K) { return K * (SEAN_TH + 2) * (SEAN_TW + 2); }

__device__ __forceinline__ void sean_stage(const SeanGeom& g, const float* __restrict__ mask,
                                           const float* __restrict__ D, float* sD, float* sM, int b, int c0, int y0,
                                           int x0) {
    int K = g.K;
    // dynamic kernels of this sample, channel slice [c0, c0+64)
    int nD = 2 * 9 * K * 64;
    for (int i = threadIdx.x; i < nD; i += blockDim.x) {
        int cl = i & 63, r = i >> 6;  // r = (s*9+tap)*K + k
        int c = c0 + cl;
        sD[i] = c < g.C ? D[((size_t)b * 18 * K + r) * g.C + c] : 0.f;
    }
    // mask tile with halo, zero outside the image (zero padding of the reference's 3x3 convs)
    int MW = SEAN_TW + 2, MH = SEAN_TH + 2;
    int nM = K * MH * MW;
    for (int i = threadIdx.x; i < nM; i += blockDim.x) {
        int xx = i % MW, yy = (i / MW) % MH, k = i / (MW * MH);
        int gy = y0 + yy - 1, gx = x0 + xx - 1;
        float v = 0.f;
        if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) v = mask[(((size_t)b * K + k) * g.H + gy) * g.W + gx];
        sM[i] = v;
    }
}

// gamma1/beta1 (without bias) at tile-local pixel (ly, lx) for channel lane cl
__device__ __forceinline__ void sean_dynconv(const float* sD, const float* sM, int K, int ly, int lx, int cl,
                                             float& g1, float& b1) {
    int MW = SEAN_TW + 2, MH = SEAN_TH + 2;
    float ag = 0.f, ab = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        int yy = ly + tap / 3, xx = lx + tap % 3;
        for (int k = 0; k < K; ++k) {
            float m = sM[(k * MH + yy) * MW + xx];  // wave-uniform address: LDS broadcast
            if (m != 0.f) {                          // one-hot masks: 9 of 9*K terms are live
                ag = fmaf(m, sD[((0 * 9 + tap) * K + k) * 64 + cl], ag);
                ab = fmaf(m, sD[((1 * 9 + tap) * K + k) * 64 + cl], ab);
            }
        }
    }
    g1 = ag;
    b1 = ab;
}

// T = storage type of the activation tensors (t, gb2, residual, out, and their gradients): float, or bf16_t on the
// mixed-precision path; everything per-(b,c), D, the biases and all arithmetic stay fp32.
template <typename T>
__global__ void __launch_bounds__(256) k_sean_fwd(SeanGeom g, const T* __restrict__ t,
                                                  const float* __restrict__ mean, const float* __restrict__ var,
                                                  const T* __restrict__ gb2, const float* __restrict__ mask,
                                                  const float* __restrict__ D, const float* __restrict__ bias_g,
                                                  const float* __restrict__ bias_b, const float* __restrict__ alpha_g,
                                                  const float* __restrict__ alpha_b, const T* __restrict__ residual,
                                                  T* __restrict__ out, int relu, float eps,
                                                  const int* __restrict__ onehot_flag) {
    if (onehot_flag && *onehot_flag == 0) return;   // one-hot masks: k_sean_fwd_onehot does the work
    DASR_DYN_SMEM(smem);
    float* sD = (float*)smem;
    float* sM = sD + sean_lds_D_floats(g.K);
    int tiles_x = (g.W + SEAN_TW - 1) / SEAN_TW;
    int x0 = (blockIdx.x % tiles_x) * SEAN_TW, y0 = (blockIdx.x / tiles_x) * SEAN_TH;
    int b = blockIdx.y, c0 = blockIdx.z * 64;
    sean_stage(g, mask, D, sD, sM, b, c0, y0, x0);
    __syncthreads();
    int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    int c = c0 + cl;
    if (c >= g.C) return;
    float a_g = alpha_g[0], a_b = alpha_b[0];
    float mu = mean[(size_t)b * g.C + c];
    float s = dasr_double_in_scale(var[(size_t)b * g.C + c], eps);
    float bg = bias_g[c], bb = bias_b[c];
    for (int lp = pl; lp < SEAN_TH * SEAN_TW; lp += 4) {
        int ly = lp / SEAN_TW, lx = lp % SEAN_TW;
        int y = y0 + ly, x = x0 + lx;
        if (y >= g.H || x >= g.W) continue;
        size_t p = ((size_t)b * g.H + y) * g.W + x;
        float g1, b1;
        sean_dynconv(sD, sM, g.K, ly, lx, cl, g1, b1);
        g1 += bg;
        b1 += bb;
        float g2 = ld1(gb2 + p * 2 * g.C + c), b2 = ld1(gb2 + p * 2 * g.C + g.C + c);
        float gam = a_g * g1 + (1.f - a_g) * g2;
        float bet = a_b * b1 + (1.f - a_b) * b2;
        float xh = (ld1(t + p * g.C + c) - mu) * s;
        float o = xh * (1.f + gam) + bet;
        if (residual) o += ld1(residual + p * g.C + c);
        if (relu) o = o > 0.f ? o : 0.f;
        st1(out + p * g.C + c, o);
    }
}

// ---------------------------------------------------------------------------------------- backward
// Pass A (per tile): g0 = dout*relu'(out); dgb2, dres; dxhat -> dt (temporarily); per-(b,c) sums
//   S1 = sum dxhat, S2 = sum dxhat*(t-mean); dbias_s; dalpha; dD (LDS accumulation, then one float
//   atomic per (tap,k,c) per workgroup).
// Pass B (elementwise): dt = s*(dxhat - S1/N) + s'(var)*(2/N)*(t-mean)*S2.
// workspace: S [B][C][2] floats.
template <typename T>
__global__ void __launch_bounds__(256) k_sean_bwd_a(SeanGeom g, const T* __restrict__ dout,
                                                    const T* __restrict__ out, const T* __restrict__ t,
                                                    const float* __restrict__ mean, const float* __restrict__ var,
                                                    const T* __restrict__ gb2, const float* __restrict__ mask,
                                                    const float* __restrict__ D, const float* __restrict__ bias_g,
                                                    const float* __restrict__ bias_b,
                                                    const float* __restrict__ alpha_g,
                                                    const float* __restrict__ alpha_b, T* __restrict__ dt,
                                                    T* __restrict__ dgb2, float* __restrict__ dD,
                                                    float* __restrict__ dbias_g, float* __restrict__ dbias_b,
                                                    float* __restrict__ dalpha_g, float* __restrict__ dalpha_b,
                                                    T* __restrict__ dres, float* __restrict__ S, int relu,
                                                    float eps, const int* __restrict__ onehot_flag) {
    if (onehot_flag && *onehot_flag == 0) return;
    DASR_DYN_SMEM(smem);
    float* sD = (float*)smem;
    float* sM = sD + sean_lds_D_floats(g.K);
    float* sdD = sM + sean_lds_M_floats(g.K);       // [2][9][K][64] accumulators
    float* sred = sdD + sean_lds_D_floats(g.K);     // [6][256] reduction scratch
    int K = g.K;
    int tiles_x = (g.W + SEAN_TW - 1) / SEAN_TW;
    int x0 = (blockIdx.x % tiles_x) * SEAN_TW, y0 = (blockIdx.x / tiles_x) * SEAN_TH;
    int b = blockIdx.y, c0 = blockIdx.z * 64;
    sean_stage(g, mask, D, sD, sM, b, c0, y0, x0);
    for (int i = threadIdx.x; i < 2 * 9 * K * 64; i += blockDim.x) sdD[i] = 0.f;
    __syncthreads();
    int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    int c = c0 + cl;
    bool live = c < g.C;
    float a_g = alpha_g[0], a_b = alpha_b[0];
    float mu = live ? mean[(size_t)b * g.C + c] : 0.f;
    float s = live ? dasr_double_in_scale(var[(size_t)b * g.C + c], eps) : 0.f;
    float bg = live ? bias_g[c] : 0.f, bb = live ? bias_b[c] : 0.f;
    float S1 = 0.f, S2 = 0.f, dag = 0.f, dab = 0.f, dbg = 0.f, dbb = 0.f;
    int MW = SEAN_TW + 2, MH = SEAN_TH + 2;
    if (live)
        for (int lp = pl; lp < SEAN_TH * SEAN_TW; lp += 4) {
            int ly = lp / SEAN_TW, lx = lp % SEAN_TW;
            int y = y0 + ly, x = x0 + lx;
            if (y >= g.H || x >= g.W) continue;
            size_t p = ((size_t)b * g.H + y) * g.W + x;
            float g0 = ld1(dout + p * g.C + c);
            if (relu && !(ld1(out + p * g.C + c) > 0.f)) g0 = 0.f;
            if (dres) st1(dres + p * g.C + c, g0);
            float g1, b1;
            sean_dynconv(sD, sM, K, ly, lx, cl, g1, b1);
            g1 += bg;
            b1 += bb;
            float g2 = ld1(gb2 + p * 2 * g.C + c), b2 = ld1(gb2 + p * 2 * g.C + g.C + c);
            float gam = a_g * g1 + (1.f - a_g) * g2;
            float xc = ld1(t + p * g.C + c) - mu;
            float xh = xc * s;
            float dgam = g0 * xh, dbet = g0;
            st1(dgb2 + p * 2 * g.C + c, (1.f - a_g) * dgam);
            st1(dgb2 + p * 2 * g.C + g.C + c, (1.f - a_b) * dbet);
            dag = fmaf(dgam, g1 - g2, dag);
            dab = fmaf(dbet, b1 - b2, dab);
            float dg1 = a_g * dgam, db1 = a_b * dbet;
            dbg += dg1;
            dbb += db1;
            for (int tap = 0; tap < 9; ++tap) {
                int yy = ly + tap / 3, xx = lx + tap % 3;
                for (int k = 0; k < K; ++k) {
                    float m = sM[(k * MH + yy) * MW + xx];
                    if (m != 0.f) {
                        atomicAdd(&sdD[((0 * 9 + tap) * K + k) * 64 + cl], dg1 * m);
                        atomicAdd(&sdD[((1 * 9 + tap) * K + k) * 64 + cl], db1 * m);
                    }
                }
            }
            float dxh = g0 * (1.f + gam);
            st1(dt + p * g.C + c, dxh);
            S1 += dxh;
            S2 = fmaf(dxh, xc, S2);
        }
    // reduce the six per-thread partials over the 4 pixel lanes
    sred[0 * 256 + threadIdx.x] = S1;
    sred[1 * 256 + threadIdx.x] = S2;
    sred[2 * 256 + threadIdx.x] = dbg;
    sred[3 * 256 + threadIdx.x] = dbb;
    sred[4 * 256 + threadIdx.x] = dag;
    sred[5 * 256 + threadIdx.x] = dab;
    __syncthreads();
    if (pl == 0) {
        float r[6];
        for (int q = 0; q < 6; ++q)
            r[q] = sred[q * 256 + cl] + sred[q * 256 + 64 + cl] + sred[q * 256 + 128 + cl] + sred[q * 256 + 192 + cl];
        if (live) {
            atomicAdd(&S[((size_t)b * g.C + c) * 2 + 0], r[0]);
            atomicAdd(&S[((size_t)b * g.C + c) * 2 + 1], r[1]);
            atomicAdd(&dbias_g[c], r[2]);
            atomicAdd(&dbias_b[c], r[3]);
        }
        float ra = r[4], rb = r[5];  // dalpha: also reduce over the 64 channel lanes (dead lanes hold 0)
        for (int off = 32; off > 0; off >>= 1) {
            ra += __shfl_down(ra, off, 64);
            rb += __shfl_down(rb, off, 64);
        }
        if (cl == 0) {
            atomicAdd(&dalpha_g[0], ra);
            atomicAdd(&dalpha_b[0], rb);
        }
    }
    // flush the dynamic-kernel gradient of this tile
    for (int i = threadIdx.x; i < 2 * 9 * K * 64; i += blockDim.x) {
        int cl2 = i & 63, r = i >> 6;
        float v = sdD[i];
        if (c0 + cl2 < g.C && v != 0.f) atomicAdd(&dD[((size_t)b * 18 * K + r) * g.C + c0 + cl2], v);
    }
}

// dt = s*(dxhat - S1/N) + s'(var)*(2/N)*(t-mean)*S2; float4 per lane when C % 4 == 0 (per-(b,c) constants are
// recomputed from mean/var/S: 4 channels x 4 small loads, all L1/L2 hits)
template <typename T>
__global__ void __launch_bounds__(256) k_sean_bwd_b(const T* __restrict__ t, const float* __restrict__ mean,
                                                    const float* __restrict__ var, const float* __restrict__ S,
                                                    T* __restrict__ dt, int HW, int C, size_t n, float eps,
                                                    float* __restrict__ amax) {
    __shared__ float s_part[16];
    const float invN = 1.0f / (float)HW;
    float om = 0.f;
    if ((C & 3) == 0) {
        const size_t n4 = n >> 2;
        const int C4 = C >> 2;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            const int c = 4 * (int)(i % C4);
            const size_t b = i / ((size_t)C4 * HW);
            const size_t bc = b * C + c;
            const float4 tv = ld4(t + 4 * i);
            float4 dv = ld4(dt + 4 * i);
            const float4 mu = *(const float4*)(mean + bc), vr = *(const float4*)(var + bc);
            const float4 s01 = *(const float4*)(S + 2 * bc), s23 = *(const float4*)(S + 2 * bc + 4);
            dv.x = dasr_double_in_scale(vr.x, eps) * (dv.x - s01.x * invN) +
                   dasr_double_in_dscale(vr.x, eps) * 2.f * invN * (tv.x - mu.x) * s01.y;
            dv.y = dasr_double_in_scale(vr.y, eps) * (dv.y - s01.z * invN) +
                   dasr_double_in_dscale(vr.y, eps) * 2.f * invN * (tv.y - mu.y) * s01.w;
            dv.z = dasr_double_in_scale(vr.z, eps) * (dv.z - s23.x * invN) +
                   dasr_double_in_dscale(vr.z, eps) * 2.f * invN * (tv.z - mu.z) * s23.y;
            dv.w = dasr_double_in_scale(vr.w, eps) * (dv.w - s23.z * invN) +
                   dasr_double_in_dscale(vr.w, eps) * 2.f * invN * (tv.w - mu.w) * s23.w;
            st4(dt + 4 * i, dv);
            om = dasr_amax4(om, dv);
        }
        if (amax) dasr_amax_commit(amax, om, s_part, dasr_flat_wg(), dasr_flat_nwg());
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        size_t b = i / ((size_t)C * HW);
        size_t bc = b * C + c;
        float v = var[bc];
        float s = dasr_double_in_scale(v, eps), ds = dasr_double_in_dscale(v, eps);
        float xc = ld1(t + i) - mean[bc];
        const float o = s * (ld1(dt + i) - S[bc * 2] * invN) + ds * 2.f * invN * xc * S[bc * 2 + 1];
        st1(dt + i, o);
        om = dasr_amax1(om, o);
    }
    if (amax) dasr_amax_commit(amax, om, s_part, dasr_flat_wg(), dasr_flat_nwg());
}


// Same as the float4 branch above for C4 = C/4 dividing 256: blockIdx.y = sample, a thread keeps ONE channel quad
// and walks the pixels, so the per-(b,c) factors (two rsqrt-style scales and their derivative: 16 sqrt/div per float4
// in the kernel above, plus two 64-bit divisions for the index) are computed once per thread.
template <typename T>
__global__ void __launch_bounds__(256) k_sean_bwd_b_rows(const T* __restrict__ t, const float* __restrict__ mean,
                                                         const float* __restrict__ var, const float* __restrict__ S,
                                                         T* __restrict__ dt, int HW, int C, float eps,
                                                         float* __restrict__ amax) {
    __shared__ float s_part[16];
    const int C4 = C >> 2, q = threadIdx.x % C4, pl = threadIdx.x / C4, npl = 256 / C4;
    const int b = blockIdx.y, c = 4 * q;
    const float invN = 1.0f / (float)HW;
    const size_t bc = (size_t)b * C + c;
    const float4 mu = *(const float4*)(mean + bc), vr = *(const float4*)(var + bc);
    const float4 s01 = *(const float4*)(S + 2 * bc), s23 = *(const float4*)(S + 2 * bc + 4);
    const float4 A = make_float4(dasr_double_in_scale(vr.x, eps), dasr_double_in_scale(vr.y, eps),
                                 dasr_double_in_scale(vr.z, eps), dasr_double_in_scale(vr.w, eps));
    const float4 Bm = make_float4(s01.x * invN, s01.z * invN, s23.x * invN, s23.z * invN);
    const float4 D2 = make_float4(dasr_double_in_dscale(vr.x, eps) * 2.f * invN, dasr_double_in_dscale(vr.y, eps) * 2.f * invN,
                                  dasr_double_in_dscale(vr.z, eps) * 2.f * invN, dasr_double_in_dscale(vr.w, eps) * 2.f * invN);
    const float4 S1 = make_float4(s01.y, s01.w, s23.y, s23.w);
    const T* tb = t + (size_t)b * HW * C + c;
    T* db = dt + (size_t)b * HW * C + c;
    float om = 0.f;
    for (int p = blockIdx.x * npl + pl; p < HW; p += gridDim.x * npl) {
        const float4 tv = ld4(tb + (size_t)p * C);
        float4 dv = ld4(db + (size_t)p * C);
        // same expression order as the general kernel: s*(dv - S0/N) + (ds*2/N)*(t - mu)*S1
        dv.x = A.x * (dv.x - Bm.x) + D2.x * (tv.x - mu.x) * S1.x;
        dv.y = A.y * (dv.y - Bm.y) + D2.y * (tv.y - mu.y) * S1.y;
        dv.z = A.z * (dv.z - Bm.z) + D2.z * (tv.z - mu.z) * S1.z;
        dv.w = A.w * (dv.w - Bm.w) + D2.w * (tv.w - mu.w) * S1.w;
        st4(db + (size_t)p * C, dv);
        om = dasr_amax4(om, dv);
    }
    if (amax) dasr_amax_commit(amax, om, s_part, dasr_flat_wg(), dasr_flat_nwg());
}

// =====================================================================================================
// One-hot fast path.
//
// The reference's masks are one-hot by construction (getDepthMask, LQGTker_Depth_dataset.py:204-225) and
// stay binary through the 'nearest' resize (normalization.py:59).  dasr_mask_compress turns the K float
// planes into one byte per pixel (region index, K = "no region") and raises a flag if any pixel is not
// one-hot; the kernels below run when the flag is clear, the general kernels above when it is set — the
// decision is taken ON THE DEVICE (both kernels are launched, one exits at its first instruction), so the
// host never synchronises.
//
// With a region index the 3x3 dynamic convolution is a GATHER: gamma1[p][c] = bias + sum_tap D[tap][r(p+tap)][c],
// 9 LDS row reads instead of 9*K multiply-adds.  Lane layout: 16 lanes x float4 = the 64 channels of one
// pixel, 4 pixels per wave-instruction (1 KiB contiguous global accesses); every ds_read_b128 lane group
// covers all 64 banks exactly once whatever rows the four pixels pick (rows are 64-dword multiples).
// =====================================================================================================
#define SF_TH 8
#define SF_TW 32

__global__ void __launch_bounds__(256) k_mask_compress(const float* __restrict__ mask, unsigned char* __restrict__ region,
                                                       int* __restrict__ flag, int K, int HW, size_t n) {
    int bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t b = i / HW, p = i % HW;
        int ones = 0, other = 0, idx = K;
        for (int k = 0; k < K; ++k) {
            float m = mask[(b * K + k) * HW + p];
            if (m == 1.f) { ++ones; idx = k; }
            else if (m != 0.f) ++other;
        }
        if (other != 0 || ones > 1) bad = 1;
        region[i] = (unsigned char)idx;
    }
    if (bad) atomicAdd(flag, 1);
}

extern "C" int dasr_mask_compress(const float* mask, unsigned char* region, int* flag, int B, int K, int H, int W,
                                  void* stream) {
    DASR_CHECK_PTR(mask); DASR_CHECK_PTR(region); DASR_CHECK_PTR(flag);
    DASR_CHECK_SHAPE(B > 0 && K > 0 && H > 0 && W > 0);
    if (K > SEAN_MAXK) return DASR_E_UNSUPPORTED;
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    size_t n = (size_t)B * H * W;
    DASR_LAUNCH(k_mask_compress, dim3(dasr_ew_grid(n)), dim3(256), 0, stream, mask, region, flag, K, H * W, n);
    DASR_RETURN_LAUNCH_STATUS();
}

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Stage D[b] (channel slice c0..c0+63) as [18][K+1][64] with a zero row K ("no region": outside the image
// or a pixel no mask claims); float4 copies, 16 threads per 64-channel row.
// With bias_g / bias_b given, the conv biases are folded into the tap-0 rows (bias + row, the first addition of the
// gather's fixed summation order, so the result is bit-identical to adding the bias first).
template <int NB = 8>
__device__ __forceinline__ void sean_stage_D(const SeanGeom& g, const float* __restrict__ D, float* sD, int b, int c0,
                                             const float* __restrict__ bias_g = nullptr,
                                             const float* __restrict__ bias_b = nullptr) {
    const int K1 = g.K + 1, rows = 18 * K1;
    const int q = threadIdx.x & 15, rstep = blockDim.x >> 4;
    const bool inC = c0 + 4 * q + 3 < g.C;
    // batches of NB rows per thread: the loads of a batch are all issued before its first LDS write (one L2 round
    // trip per batch instead of one per row - a "load; wait; write" loop is a chain of dependent round trips)
    for (int r0 = threadIdx.x >> 4; r0 < rows; r0 += NB * rstep) {
        float4 v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int r = r0 + u * rstep;
            const int st = r / K1, k = r - st * K1;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rows && k < g.K && inC) v[u] = *(const float4*)(D + (((size_t)b * 18 + st) * g.K + k) * g.C + c0 + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int r = r0 + u * rstep;
            if (r >= rows) continue;
            const int st = r / K1;
            if (bias_g && inC && (st == 0 || st == 9)) {
                const float4 bv = *(const float4*)((st == 0 ? bias_g : bias_b) + c0 + 4 * q);
                v[u] = make_float4(bv.x + v[u].x, bv.y + v[u].y, bv.z + v[u].z, bv.w + v[u].w);
            }
            *(float4*)(sD + r * 64 + 4 * q) = v[u];
        }
    }
}
// region tile with a 1-pixel halo; TH rows x SF_TW columns; two bytes per thread and pass, loads before stores
__device__ __forceinline__ void sean_stage_R(const SeanGeom& g, const unsigned char* __restrict__ region,
                                             unsigned char* sR, int b, int y0, int x0, int TH) {
    const int n = (TH + 2) * (SF_TW + 2);
    for (int i0 = threadIdx.x; i0 < n; i0 += 2 * blockDim.x) {
        unsigned char v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + u * blockDim.x;
            const int gy = y0 + i / (SF_TW + 2) - 1, gx = x0 + i % (SF_TW + 2) - 1;
            v[u] = (unsigned char)g.K;
            if (i < n && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) v[u] = region[((size_t)b * g.H + gy) * g.W + gx];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < n) sR[i] = v[u];
        }
    }
}
// gamma1 / beta1 of tile-local pixel (ly, lx): 9 row gathers from sD each (same summation order as the general
// kernel: bias, then taps 0..8).  The scheduling barrier keeps the 9 gamma rows and the 9 beta rows from being
// in registers at the same time (the fwd kernel must stay within 168 VGPRs for 3 waves per SIMD).
__device__ __forceinline__ void sean_gather(const float* sD, const unsigned char* sR, int K1, int ly, int lx, int cq,
                                            float4 bg, float4 bb, float4& g1, float4& b1) {
    int k[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) k[tap] = sR[(ly + tap / 3) * (SF_TW + 2) + lx + tap % 3];
    g1 = bg;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) g1 = f4add(g1, *(const float4*)(sD + ((0 * 9 + tap) * K1 + k[tap]) * 64 + 4 * cq));
    DASR_SCHED_BARRIER();
    b1 = bb;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) b1 = f4add(b1, *(const float4*)(sD + ((1 * 9 + tap) * K1 + k[tap]) * 64 + 4 * cq));
}
// Interior pixels: when the nine region bytes of a pixel's 3x3 neighbourhood are equal (most pixels: depth regions are
// blobs), gamma1 / beta1 are one row of a per-region table S[s][k] = bias + D[s][0][k] + ... + D[s][8][k] (summed in the
// gather's own order: bit-identical) - two LDS reads instead of eighteen and no adds.  Taken when EVERY lane of the wave
// is interior (a vote), so the wave does not run both paths.
__device__ __forceinline__ void sean_gather_interior(const float* sD, const float* sS, const unsigned char* sR, int K,
                                                     int K1, int ly, int lx, int cq, float4 bg, float4 bb, float4& g1,
                                                     float4& b1) {
    int k[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) k[tap] = sR[(ly + tap / 3) * (SF_TW + 2) + lx + tap % 3];
    bool same = true;
#pragma unroll
    for (int tap = 1; tap < 9; ++tap) same = same && k[tap] == k[0];
    if (__all(same ? 1 : 0)) {             // (a pixel outside the image sees nine "no region" bytes: clamped, result unused)
        const int kk = k[0] < K ? k[0] : K - 1;
        g1 = *(const float4*)(sS + (0 * K + kk) * 64 + 4 * cq);
        b1 = *(const float4*)(sS + (1 * K + kk) * 64 + 4 * cq);
        return;
    }
    g1 = bg;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) g1 = f4add(g1, *(const float4*)(sD + ((0 * 9 + tap) * K1 + k[tap]) * 64 + 4 * cq));
    DASR_SCHED_BARRIER();
    b1 = bb;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) b1 = f4add(b1, *(const float4*)(sD + ((1 * 9 + tap) * K1 + k[tap]) * 64 + 4 * cq));
}
// the same with the biases already folded into the tap-0 rows (sean_stage_D with bias pointers)
__device__ __forceinline__ void sean_gather_folded(const float* sD, const unsigned char* sR, int K1, int ly, int lx,
                                                   int cq, float4& g1, float4& b1) {
    int k[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) k[tap] = sR[(ly + tap / 3) * (SF_TW + 2) + lx + tap % 3];
    g1 = *(const float4*)(sD + ((0 * 9 + 0) * K1 + k[0]) * 64 + 4 * cq);
#pragma unroll
    for (int tap = 1; tap < 9; ++tap) g1 = f4add(g1, *(const float4*)(sD + ((0 * 9 + tap) * K1 + k[tap]) * 64 + 4 * cq));
    DASR_SCHED_BARRIER();
    b1 = *(const float4*)(sD + ((1 * 9 + 0) * K1 + k[0]) * 64 + 4 * cq);
#pragma unroll
    for (int tap = 1; tap < 9; ++tap) b1 = f4add(b1, *(const float4*)(sD + ((1 * 9 + tap) * K1 + k[tap]) * 64 + 4 * cq));
    DASR_SCHED_BARRIER();
}

// Forward.  Workgroup = 256 threads, tile = SF_TH rows x 32 columns; wave w owns rows w, w+4, ...  A lane holds 4
// channels of one pixel (float4), a wave-instruction moves 4 x 256 B contiguous.
//
// The streaming loop is written so that the compiler can keep TWO groups of loads in flight per wave: loads are
// unconditional (coordinates clamped into the image, only the store is predicated), RELU / HAS_RES are template
// parameters, and every load is "scalar row pointer + 32-bit per-lane byte offset" (SGPR-base form, no 64-bit
// address registers; a row is W*2C*4 bytes, far below 4 GiB).  With predicated loads the join of the
// skipped-consume path forced `s_waitcnt vmcnt(0)` at the loop head (a pending load into a register about to be
// overwritten), which serialised the two buffers.  The first group of a tile is requested before the tile's D /
// region staging (and, for the first tile, before the one-hot flag is even known), so the staging latency and the
// two barriers are covered by loads in flight.
struct SeanTile { int b, y0, x0; };

// Residual variant: one more operand in flight per pixel group.  Under the 168-register budget of three workgroups per CU it
// spilled (8.4 MB of scratch writes per launch at B = 16); with 256 registers and TWO workgroups per CU (the launch then deals
// its tiles over 512 slots) it does not: 144.5 -> 138 us at B = 32, 72.1 -> 70.9 us at B = 16 (A/B on one MI355X, round 3).
#ifndef DASR_SEAN_RES_OCC
#define DASR_SEAN_RES_OCC 2
#endif
template <bool RELU, bool HAS_RES, typename TA, bool AMAX = false>
__global__ void __launch_bounds__(256, HAS_RES ? DASR_SEAN_RES_OCC : 3) k_sean_fwd_onehot(SeanGeom g, const TA* __restrict__ t,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ var,
                                                            const TA* __restrict__ gb2,
                                                            const unsigned char* __restrict__ region,
                                                            const int* __restrict__ flag, const float* __restrict__ D,
                                                            const float* __restrict__ bias_g,
                                                            const float* __restrict__ bias_b,
                                                            const float* __restrict__ alpha_g,
                                                            const float* __restrict__ alpha_b,
                                                            const TA* __restrict__ residual,
                                                            TA* __restrict__ out, float eps, int tiles_per_wg,
                                                            float* __restrict__ amax) {
    DASR_DYN_SMEM(smem);
    constexpr int TH = SF_TH, NG = TH;                         // NG groups of 8 pixels per wave and tile (TH/4 rows x 4)
    const int K1 = g.K + 1;
    float* sD = (float*)smem;                                  // [18][K+1][64]
    unsigned char* sR = (unsigned char*)(sD + 18 * K1 * 64);   // [(TH+2)*(TW+2)]
    const int tiles_x = (g.W + SF_TW - 1) / SF_TW, tiles_y = (g.H + TH - 1) / TH;
    const int tiles_per_sample = tiles_x * tiles_y;
    const int c0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wv = DASR_UNIFORM((int)(threadIdx.x >> 6));
    const int cq = lane & 15, ps = lane >> 4;
    const bool live = c0 + 4 * cq < g.C;
    const int c = live ? c0 + 4 * cq : 0;                      // dead lanes (C % 64 != 0) shadow channel 0, never store
    // Persistent workgroup: a contiguous chunk of (sample, tile) pairs; D[b] is restaged only when the sample
    // changes, so its 50 KB L2->LDS copy is paid about once per workgroup instead of once per tile.
    // tiles_per_wg packs (base, rem): the first `rem` workgroups take base + 1 consecutive tiles, the others base, so
    // that ALL 768 resident slots are used and every CU ends up with the same number of tiles (640 equal workgroups
    // left CUs with 4 or 6 tiles at B = 16)
    const int base = tiles_per_wg >> 16, rem = tiles_per_wg & 0xffff;
    const int first = blockIdx.x * base + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
    int last = first + base + ((int)blockIdx.x < rem ? 1 : 0);
    if (last > g.B * tiles_per_sample) last = g.B * tiles_per_sample;
    if (first >= last) {
        if (AMAX) dasr_amax_commit_idle(amax, dasr_flat_wg(), dasr_flat_nwg());
        return;
    }
    auto tile_of = [&](int tt) {
        const int tile = tt % tiles_per_sample;
        return SeanTile{tt / tiles_per_sample, (tile / tiles_x) * TH, (tile % tiles_x) * SF_TW};
    };
    struct Buf { float4 tv[2], g2[2], b2[2], rv[2]; };
    auto issue = [&](const SeanTile& T, int grp, Buf& f) {
        const int y = imin(T.y0 + wv + 4 * (grp >> 2), g.H - 1);                // wave-uniform
        const size_t row = ((size_t)T.b * g.H + y) * g.W;                       // scalar
        const char* trow = (const char*)(t + row * g.C);
        const char* grow = (const char*)(gb2 + row * 2 * g.C);
        const char* brow = grow + (size_t)g.C * sizeof(TA);                      // beta half of the (gamma2 | beta2) pair
        const char* rrow = HAS_RES ? (const char*)(residual + row * g.C) : nullptr;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned x = (unsigned)imin(T.x0 + 8 * (grp & 3) + 4 * u + ps, g.W - 1);
            const unsigned ot = (x * (unsigned)g.C + (unsigned)c) * (unsigned)sizeof(TA);
            const unsigned og = (x * 2u * (unsigned)g.C + (unsigned)c) * (unsigned)sizeof(TA);
            f.tv[u] = ld4_nt((const TA*)(trow + ot));                            // streamed once: keep D / halos cached
            f.g2[u] = ld4_nt((const TA*)(grow + og));
            f.b2[u] = ld4_nt((const TA*)(brow + og));
            if (HAS_RES) f.rv[u] = ld4((const TA*)(rrow + ot));
        }
    };
    Buf fa, fb;
    SeanTile T = tile_of(first);
    issue(T, 0, fa);
    if (flag && *flag != 0) return;   // masks are not one-hot: the general kernel does the work
    const float a_g = alpha_g[0], a_b = alpha_b[0];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int cur_b = -1;
    float4 mu = zero4, sc = zero4;
    float om = 0.f;                                        // (AMAX) running max |out| of this lane
    for (int tt = first; tt < last; ++tt) {
        const int b = T.b, y0 = T.y0, x0 = T.x0;
        __syncthreads();                                   // previous tile is done with sR (and sD)
        if (b != cur_b) {
            sean_stage_D<HAS_RES ? 2 : 4>(g, D, sD, b, c0, bias_g, bias_b);
            cur_b = b;
            if (live) {
                mu = *(const float4*)(mean + (size_t)b * g.C + c);
                const float4 vr = *(const float4*)(var + (size_t)b * g.C + c);
                sc = make_float4(dasr_double_in_scale(vr.x, eps), dasr_double_in_scale(vr.y, eps),
                                 dasr_double_in_scale(vr.z, eps), dasr_double_in_scale(vr.w, eps));
            }
        }
        sean_stage_R(g, region, sR, b, y0, x0, TH);
        __syncthreads();
        auto consume = [&](int grp, const Buf& f) {
            const int ly = wv + 4 * (grp >> 2), y = y0 + ly;
            const size_t row = ((size_t)b * g.H + imin(y, g.H - 1)) * g.W;
            char* orow = (char*)(out + row * g.C);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int lx = 8 * (grp & 3) + 4 * u + ps, x = x0 + lx;
                float4 g1, b1;
                sean_gather_folded(sD, sR, K1, ly, lx, cq, g1, b1);
                float4 o;
                o.x = (f.tv[u].x - mu.x) * sc.x * (1.f + a_g * g1.x + (1.f - a_g) * f.g2[u].x) + a_b * b1.x + (1.f - a_b) * f.b2[u].x;
                o.y = (f.tv[u].y - mu.y) * sc.y * (1.f + a_g * g1.y + (1.f - a_g) * f.g2[u].y) + a_b * b1.y + (1.f - a_b) * f.b2[u].y;
                o.z = (f.tv[u].z - mu.z) * sc.z * (1.f + a_g * g1.z + (1.f - a_g) * f.g2[u].z) + a_b * b1.z + (1.f - a_b) * f.b2[u].z;
                o.w = (f.tv[u].w - mu.w) * sc.w * (1.f + a_g * g1.w + (1.f - a_g) * f.g2[u].w) + a_b * b1.w + (1.f - a_b) * f.b2[u].w;
                if (HAS_RES) { o.x += f.rv[u].x; o.y += f.rv[u].y; o.z += f.rv[u].z; o.w += f.rv[u].w; }
                if (RELU) {
                    o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f;
                    o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
                }
                if (AMAX) om = dasr_amax4(om, o);     // (clamped / shadow lanes hold duplicates of stored values)
                if (live && y < g.H && x < g.W) {
                    TA* dst = (TA*)(orow + ((unsigned)x * (unsigned)g.C + (unsigned)c) * (unsigned)sizeof(TA));
                    st4_nt(dst, o);       // written once, read by the next kernel from HBM: do not displace the inputs
                }
            }
        };
        // Two statically named register buffers: group n+1 is requested before group n is consumed, so a wave keeps
        // 6-16 KiB of HBM reads in flight while it gathers and modulates.  fa already holds group 0 of this tile.
#pragma unroll 1
        for (int grp = 0; grp < NG - 2; grp += 2) {
            issue(T, grp + 1, fb);
            consume(grp, fa);
            issue(T, grp + 2, fa);
            consume(grp + 1, fb);
        }
        issue(T, NG - 1, fb);
        consume(NG - 2, fa);
        if (tt + 1 < last) {
            T = tile_of(tt + 1);
            issue(T, 0, fa);
        }
        consume(NG - 1, fb);
    }
    if (AMAX) dasr_amax_commit(amax, om, sD, dasr_flat_wg(), dasr_flat_nwg());     // (sD: dead after the last tile)
}

// ---- forward, one-hot, bf16 activations with EIGHT channels (16 bytes) per lane --------------------------------------
// With 4 bf16 channels per lane a wave instruction moves 512 bytes, half of what the fp32 instantiation has in flight per
// instruction - the bf16 forward sat at 0.57-0.66 of the HBM peak against 0.75-0.8.  Here a lane owns 8 channels of one pixel
// (one 16-byte load per operand, one 16-byte store), a wave instruction covers 8 pixels x 64 channels = 1 KB, and the
// structure is otherwise that of k_sean_fwd_onehot (groups of 8 pixels, two register buffers, persistent tiles).  The gather
// reads the pixel's nine region bytes once and two float4 per tap and table (channels 8 c8 .. + 3 and + 4 .. + 7): lanes of
// DIFFERENT pixels now share a 16-lane LDS pass, a 2-way conflict only where the two pixels lie in different regions.
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 ld8_nt(const bf16_t* p) {
    const s16x8 r = __builtin_nontemporal_load((const s16x8*)p);
    bf16x8 v;
    __builtin_memcpy(&v, &r, 16);
    return v;
}
__device__ __forceinline__ bf16x8 ld8(const bf16_t* p) { return *(const bf16x8*)p; }
template <bool RELU, bool HAS_RES>
__global__ void __launch_bounds__(256, HAS_RES ? DASR_SEAN_RES_OCC : 3) k_sean_fwd_onehot_w8(SeanGeom g, const bf16_t* __restrict__ t,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ var,
                                                            const bf16_t* __restrict__ gb2,
                                                            const unsigned char* __restrict__ region,
                                                            const int* __restrict__ flag, const float* __restrict__ D,
                                                            const float* __restrict__ bias_g,
                                                            const float* __restrict__ bias_b,
                                                            const float* __restrict__ alpha_g,
                                                            const float* __restrict__ alpha_b,
                                                            const bf16_t* __restrict__ residual,
                                                            bf16_t* __restrict__ out, float eps, int tiles_per_wg) {
    DASR_DYN_SMEM(smem);
    constexpr int TH = SF_TH, NG = TH;                         // NG groups of 8 pixels per wave and tile (TH/4 rows x 4)
    const int K1 = g.K + 1;
    float* sD = (float*)smem;                                  // [18][K+1][64]
    unsigned char* sR = (unsigned char*)(sD + 18 * K1 * 64);   // [(TH+2)*(TW+2)]
    const int tiles_x = (g.W + SF_TW - 1) / SF_TW, tiles_y = (g.H + TH - 1) / TH;
    const int tiles_per_sample = tiles_x * tiles_y;
    const int c0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wv = DASR_UNIFORM((int)(threadIdx.x >> 6));
    const int c8 = lane & 7, ps = lane >> 3;
    const bool live = c0 + 8 * c8 < g.C;
    const int c = live ? c0 + 8 * c8 : 0;                      // dead lanes (C % 64 != 0) shadow channel 0, never store
    const int base = tiles_per_wg >> 16, rem = tiles_per_wg & 0xffff;          // (as k_sean_fwd_onehot)
    const int first = blockIdx.x * base + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
    int last = first + base + ((int)blockIdx.x < rem ? 1 : 0);
    if (last > g.B * tiles_per_sample) last = g.B * tiles_per_sample;
    if (first >= last) return;
    auto tile_of = [&](int tt) {
        const int tile = tt % tiles_per_sample;
        return SeanTile{tt / tiles_per_sample, (tile / tiles_x) * TH, (tile % tiles_x) * SF_TW};
    };
    struct Buf { bf16x8 tv, g2, b2, rv; };
    auto issue = [&](const SeanTile& T, int grp, Buf& f) {
        const int y = imin(T.y0 + wv + 4 * (grp >> 2), g.H - 1);                // wave-uniform
        const size_t row = ((size_t)T.b * g.H + y) * g.W;                       // scalar
        const char* trow = (const char*)(t + row * g.C);
        const char* grow = (const char*)(gb2 + row * 2 * g.C);
        const char* brow = grow + (size_t)g.C * sizeof(bf16_t);                  // beta half of the (gamma2 | beta2) pair
        const char* rrow = HAS_RES ? (const char*)(residual + row * g.C) : nullptr;
        const unsigned x = (unsigned)imin(T.x0 + 8 * (grp & 3) + ps, g.W - 1);
        const unsigned ot = (x * (unsigned)g.C + (unsigned)c) * 2u;
        const unsigned og = (x * 2u * (unsigned)g.C + (unsigned)c) * 2u;
        f.tv = ld8_nt((const bf16_t*)(trow + ot));                               // streamed once: keep D / halos cached
        f.g2 = ld8_nt((const bf16_t*)(grow + og));
        f.b2 = ld8_nt((const bf16_t*)(brow + og));
        if (HAS_RES) f.rv = ld8((const bf16_t*)(rrow + ot));
    };
    Buf fa, fb;
    SeanTile T = tile_of(first);
    issue(T, 0, fa);
    if (flag && *flag != 0) return;   // masks are not one-hot: the general kernel does the work
    const float a_g = alpha_g[0], a_b = alpha_b[0];
    int cur_b = -1;
    float mu[8], sc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu[j] = 0.f; sc[j] = 0.f; }
    for (int tt = first; tt < last; ++tt) {
        const int b = T.b, y0 = T.y0, x0 = T.x0;
        __syncthreads();                                   // previous tile is done with sR (and sD)
        if (b != cur_b) {
            sean_stage_D<HAS_RES ? 2 : 4>(g, D, sD, b, c0, bias_g, bias_b);
            cur_b = b;
            if (live) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    mu[j] = mean[(size_t)b * g.C + c + j];
                    sc[j] = dasr_double_in_scale(var[(size_t)b * g.C + c + j], eps);
                }
            }
        }
        sean_stage_R(g, region, sR, b, y0, x0, TH);
        __syncthreads();
        auto consume = [&](int grp, const Buf& f) {
            const int ly = wv + 4 * (grp >> 2), y = y0 + ly;
            const size_t row = ((size_t)b * g.H + imin(y, g.H - 1)) * g.W;
            char* orow = (char*)(out + row * g.C);
            const int lx = 8 * (grp & 3) + ps, x = x0 + lx;
            int k[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) k[tap] = sR[(ly + tap / 3) * (SF_TW + 2) + lx + tap % 3];
            bf16x8 o8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cq = 2 * c8 + h;
                float4 g1 = *(const float4*)(sD + ((0 * 9 + 0) * K1 + k[0]) * 64 + 4 * cq);
#pragma unroll
                for (int tap = 1; tap < 9; ++tap) g1 = f4add(g1, *(const float4*)(sD + ((0 * 9 + tap) * K1 + k[tap]) * 64 + 4 * cq));
                float4 b1 = *(const float4*)(sD + ((1 * 9 + 0) * K1 + k[0]) * 64 + 4 * cq);
#pragma unroll
                for (int tap = 1; tap < 9; ++tap) b1 = f4add(b1, *(const float4*)(sD + ((1 * 9 + tap) * K1 + k[tap]) * 64 + 4 * cq));
                const float g1v[4] = {g1.x, g1.y, g1.z, g1.w}, b1v[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * h + j;
                    float o = (dasr_bf2f(f.tv[e]) - mu[e]) * sc[e] * (1.f + a_g * g1v[j] + (1.f - a_g) * dasr_bf2f(f.g2[e])) +
                              a_b * b1v[j] + (1.f - a_b) * dasr_bf2f(f.b2[e]);
                    if (HAS_RES) o += dasr_bf2f(f.rv[e]);
                    if (RELU) o = o > 0.f ? o : 0.f;
                    o8[e] = dasr_f2bf(o);
                }
            }
            if (live && y < g.H && x < g.W) {
                s16x8 r;
                __builtin_memcpy(&r, &o8, 16);
                __builtin_nontemporal_store(r, (s16x8*)(orow + ((unsigned)x * (unsigned)g.C + (unsigned)c) * 2u));
            }
        };
#pragma unroll 1
        for (int grp = 0; grp < NG - 2; grp += 2) {
            issue(T, grp + 1, fb);
            consume(grp, fa);
            issue(T, grp + 2, fa);
            consume(grp + 1, fb);
        }
        issue(T, NG - 1, fb);
        consume(NG - 2, fa);
        if (tt + 1 < last) {
            T = tile_of(tt + 1);
            issue(T, 0, fa);
        }
        consume(NG - 1, fb);
    }
}

// ---- backward, pass A, one-hot -------------------------------------------------------------------------
// The dynamic-kernel gradient  dD[s][tap][k][c] = sum_p [r(p+tap) == k] * G_s[p][c],  G = (a_g*dgamma, a_b*dbeta),
// is a segmented reduction by region.  LDS float atomics are far too slow for it on gfx950 (measured ~3 cycles
// per LANE), so it runs on the matrix cores: per tap, dD_tap = O_tap^T . G with O_tap the pixels x regions one-hot
// matrix (built on the fly from the region bytes), M = region (<= 16), N = 16 channels, K = pixels.
//
// Workgroup = 512 threads (8 waves), tile = SB_TH<T> rows x 32 columns, persistent over the tiles of ONE sample.
//   phase 1 (all waves, elementwise): everything per pixel (dgb2, dres, dxhat -> dt, the per-channel sums) and
//            G -> LDS tile sG [pixels][128 ch] in bf16.  All of a tile's 4-pixel steps are in flight together
//            (unconditional clamped loads into registers) and the NEXT tile's loads are issued before the matrix phase,
//            so HBM latency hides under it and under the barriers.
//   phase 2 (matrix cores, v_mfma_f32_16x16x32_bf16: K = 32 pixels per instruction): wave w accumulates tap w for all
//            128 channels (8 N-tiles) and N-tile w of tap 8: 9 accumulators of 4 VGPRs, resident for the whole kernel.
//            The one-hot operand is exact in bf16; the G operand is read with ds_read_b64_tr_b16 (a lane needs 8
//            consecutive PIXELS of one channel).  bf16 activations: G is rounded to bf16 once (its consumers, the
//            gamma_s / beta_s gradients, are sums over ~10^5 pixels).  fp32 activations: G is split into THREE bf16
//            pieces (hi, mid, lo: 3 x 8 = 24 mantissa bits, each residual exact), three MFMAs per step - every product
//            with a one-hot factor is exact, the accumulation is fp32, so the result is an fp32 sum in a different order;
//            5x fewer matrix-pipe cycles than the v_mfma_f32_16x16x4_f32 version this replaces (80 of its 300 us).
// At the end each workgroup writes its dD as a slab; k_sean_dD_reduce sums the slabs in a fixed order.
// The two phases of consecutive tiles OVERLAP: sG / sR are double / triple buffered, one barrier per tile, and inside an
// iteration waves 0-3 run [phase 1 of tile i, phase 2 of tile i-1] while waves 4-7 (the second wave of each SIMD) run
// them in the opposite order - the VALU-heavy elementwise phase of one wave shares its SIMD with the MFMA / LDS phase of
// the other instead of both waves doing the same thing at the same time (phases back to back: 8.9 us per 64 pixels at
// fp32, of which each phase is about a third).
#ifndef SB_DEEP_PREFETCH
#define SB_DEEP_PREFETCH 0      // 1: bf16 loads two tiles ahead over two register sets (measured: no change - the kernel is
                                // instruction-bound, not bound by bytes in flight)
#endif
#ifndef SB_STEP_ISSUE
#define SB_STEP_ISSUE 0      // 1: the next tile's loads of step u right after step u is consumed (measured: no better)
#endif
template <typename T> struct SeanBwdCfg;
template <> struct SeanBwdCfg<bf16_t> { static constexpr int NP = 1; };   // bf16 pieces of G
template <> struct SeanBwdCfg<float>  { static constexpr int NP = 3; };   // three exact pieces
// tile rows: as many as the two G buffers leave room for beside the D table (160 KB of LDS)
__host__ __device__ constexpr int sean_bwd_lds_bytes(int K, int TH, int NP) {
    return 4 * (18 * (K + 1) * 64 + 8 * 18 * 16 + 2 * K * 64) + 2 * (2 * NP * TH * 32 * 160) + 3 * (((TH + 2) * 34 + 15) / 16 * 16) + 16;
}
#define SB16_GST 160           // sG pixel stride in bf16 elements (320 B: the 4 pixel rows of a transposed read fall in
                               // four disjoint 64-byte bank ranges)
// Eight consecutive region bytes starting at an arbitrary (unaligned) LDS address, as two dwords: three aligned
// ds_read_b32 and two v_alignbyte_b32.  (Written as eight byte reads, the compiler merges them into ONE ds_read_b64 of
// unknown alignment, which does not return the bytes at the misaligned address on this hardware: the first GPU run of
// this kernel had an O(1) error in dD that the CPU emulator could not show.)
__device__ __forceinline__ void lds_ld8_unaligned(const unsigned char* p, unsigned& lo, unsigned& hi) {
#if DASR_DEVICE_BUILD
    const unsigned addr = (unsigned)(size_t)p, sh = addr & 3u;
    const unsigned* q = (const unsigned*)(p - sh);
    const unsigned w0 = q[0], w1 = q[1], w2 = q[2];
    lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
    hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
#else
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
#endif
}

template <typename T, int NST> struct SeanBwdLoads;
template <int NST> struct SeanBwdLoads<bf16_t, NST> { bf16x4 g0[NST], ov[NST], tv[NST], g2[NST], b2[NST]; };
template <int NST> struct SeanBwdLoads<float, NST> { float4 g0[NST], ov[NST], tv[NST], g2[NST], b2[NST]; };
__device__ __forceinline__ float4 sean_f4(bf16x4 v) {
    return make_float4(dasr_bf2f(v[0]), dasr_bf2f(v[1]), dasr_bf2f(v[2]), dasr_bf2f(v[3]));
}
__device__ __forceinline__ float4 sean_f4(float4 v) { return v; }
__device__ __forceinline__ void sean_ldraw(const bf16_t* p, bf16x4& v) { v = *(const bf16x4*)p; }
__device__ __forceinline__ void sean_ldraw(const float* p, float4& v) { v = *(const float4*)p; }

template <typename T, int SB_TH, bool AMAX = false>
__global__ void __launch_bounds__(512) k_sean_bwd_a_onehot(
    SeanGeom g, const T* __restrict__ dout, const T* __restrict__ out, const T* __restrict__ t,
    const float* __restrict__ mean, const float* __restrict__ var, const T* __restrict__ gb2,
    const unsigned char* __restrict__ region, const int* __restrict__ flag, const float* __restrict__ D,
    const float* __restrict__ bias_g, const float* __restrict__ bias_b, const float* __restrict__ alpha_g,
    const float* __restrict__ alpha_b, T* __restrict__ dt, T* __restrict__ dgb2, float* __restrict__ dD_slabs,
    float* __restrict__ dbias_g, float* __restrict__ dbias_b, float* __restrict__ dalpha_g,
    float* __restrict__ dalpha_b, T* __restrict__ dres, float* __restrict__ S, int relu, float eps, int ntiles,
    float* __restrict__ dgb2_amax) {
    if (flag && *flag != 0) return;
    DASR_DYN_SMEM(smem);
    constexpr int NP = SeanBwdCfg<T>::NP;
    constexpr int WPR = 8 / SB_TH, NST = SB_TH;                 // waves per tile row, 4-pixel steps per wave and tile
    constexpr int PLANE = SB_TH * SF_TW * SB16_GST;             // one bf16 plane of G
    constexpr int NR = (SB_TH + 2) * (SF_TW + 2), NRP = (NR + 15) / 16 * 16;   // region bytes of a tile with halo
    const int K1 = g.K + 1;
    float* sD = (float*)smem;                                   // [18][K+1][64]
    bf16_t* sG = (bf16_t*)(sD + 18 * K1 * 64);                  // [2][NP][SB_TH*SF_TW][SB16_GST]: gamma part | beta part
    float* sred = (float*)(sG + 2 * NP * PLANE);                // [8 waves][18][16] reduction scratch
    float* sS = sred + 8 * 18 * 16;                             // [2][K][64]: bias + all nine taps of ONE region (interior pixels)
    unsigned char* sR = (unsigned char*)(sS + 2 * g.K * 64);    // [3][NRP]
    const int b = blockIdx.y, c0 = blockIdx.z * 64;
    const int tiles_x = (g.W + SF_TW - 1) / SF_TW;
    // the wave id decides the order of the phases and guards MFMAs: keep it in an SGPR (an MFMA ignores EXEC)
    const int lane = threadIdx.x & 63, wv = DASR_UNIFORM((int)(threadIdx.x >> 6));
    const int cq = lane & 15, ps = lane >> 4;
    const bool live = c0 + 4 * cq < g.C;
    const int c = live ? c0 + 4 * cq : 0;                       // dead lanes shadow channel 0 and never store
    const float a_g = alpha_g[0], a_b = alpha_b[0];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 mu = zero4, sc = zero4, bg = zero4, bb = zero4;
    if (live) {
        mu = *(const float4*)(mean + (size_t)b * g.C + c);
        const float4 vr = *(const float4*)(var + (size_t)b * g.C + c);
        sc = make_float4(dasr_double_in_scale(vr.x, eps), dasr_double_in_scale(vr.y, eps),
                         dasr_double_in_scale(vr.z, eps), dasr_double_in_scale(vr.w, eps));
        bg = *(const float4*)(bias_g + c);
        bb = *(const float4*)(bias_b + c);
    }
    float4 S1 = zero4, S2 = zero4, dbg = zero4, dbb = zero4;
    float dag = 0.f, dab = 0.f;
    float gm = 0.f;                                             // (AMAX) running max |dgb2| of this lane
    f32x4 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;
    sean_stage_D(g, D, sD, b, c0);
    __syncthreads();
    // per-region table of the interior fast path (sean_gather_interior): the gather's own summation order
    for (int e = threadIdx.x; e < 2 * g.K * 16; e += blockDim.x) {
        const int q = e & 15, k = (e >> 4) % g.K, sgb = e / (16 * g.K);
        float4 v = zero4;
        if (c0 + 4 * q + 3 < g.C) v = *(const float4*)((sgb == 0 ? bias_g : bias_b) + c0 + 4 * q);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) v = f4add(v, *(const float4*)(sD + ((sgb * 9 + tap) * K1 + k) * 64 + 4 * q));
        *(float4*)(sS + (sgb * g.K + k) * 64 + 4 * q) = v;
    }
    const int mydy = wv / 3, mydx = wv % 3;     // tap of this wave (taps 0..7); tap 8 = (2,2) is shared
    const int ly = wv / WPR, lxw = (SF_TW / WPR) * (wv % WPR);   // phase 1: this wave's tile row and first column
    // unconditional loads of the tile's 4-pixel steps (coordinates clamped into the image)
    auto issue_step = [&](int tile, int u, SeanBwdLoads<T, NST>& f) {
        const int x0 = (tile % tiles_x) * SF_TW, y0 = (tile / tiles_x) * SB_TH;
        const int y = imin(y0 + ly, g.H - 1);
        const size_t row = ((size_t)b * g.H + y) * g.W;
        const int x = imin(x0 + lxw + 4 * u + ps, g.W - 1);
        const size_t p = row + x;
        sean_ldraw(dout + p * g.C + c, f.g0[u]);
        sean_ldraw(out + p * g.C + c, f.ov[u]);
        sean_ldraw(t + p * g.C + c, f.tv[u]);
        sean_ldraw(gb2 + p * 2 * g.C + c, f.g2[u]);
        sean_ldraw(gb2 + p * 2 * g.C + g.C + c, f.b2[u]);
    };
    auto issue = [&](int tile, SeanBwdLoads<T, NST>& f) {
#pragma unroll
        for (int u = 0; u < NST; ++u) issue_step(tile, u, f);
    };
    auto f4 = [](auto v) { return sean_f4(v); };
    // the region bytes of a tile with halo (K = "no region" outside the image): one byte per thread of waves 0-3
    auto issue_r = [&](int tile) {
        const int x0 = (tile % tiles_x) * SF_TW, y0 = (tile / tiles_x) * SB_TH;
        const int i = (int)threadIdx.x < NR ? (int)threadIdx.x : 0;
        const int gy = y0 + i / (SF_TW + 2) - 1, gx = x0 + i % (SF_TW + 2) - 1;
        unsigned char v = (unsigned char)g.K;
        if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) v = region[((size_t)b * g.H + gy) * g.W + gx];
        return v;
    };
    // ---- phase 1 of one tile: per-pixel work, G -> sGb; region bytes of the tile in sRb
    // (the registers of step u are free once it is consumed: the NEXT tile's loads of that step are issued right there)
    auto phase1 = [&](int tile, int next_tile, SeanBwdLoads<T, NST>& cur, bf16_t* sGb, const unsigned char* sRb) {
        const int x0 = (tile % tiles_x) * SF_TW, y0 = (tile / tiles_x) * SB_TH;
        const int y = y0 + ly;
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int lx = lxw + 4 * u + ps, x = x0 + lx;
            float4 G1 = zero4, B1 = zero4;
            // (outside the per-pixel condition: every lane of the wave takes part in the vote; the reads are LDS only)
            float4 g1, b1;
            sean_gather_interior(sD, sS, sRb, g.K, K1, ly, lx, cq, bg, bb, g1, b1);
            if (live && y < g.H && x < g.W) {
                const size_t p = ((size_t)b * g.H + y) * g.W + x;
                float4 g0 = f4(cur.g0[u]);
                if (relu) {
                    const float4 ov = f4(cur.ov[u]);
                    g0.x = ov.x > 0.f ? g0.x : 0.f; g0.y = ov.y > 0.f ? g0.y : 0.f;
                    g0.z = ov.z > 0.f ? g0.z : 0.f; g0.w = ov.w > 0.f ? g0.w : 0.f;
                }
                if (dres) st4(dres + p * g.C + c, g0);
                const float4 tv = f4(cur.tv[u]), g2 = f4(cur.g2[u]), b2 = f4(cur.b2[u]);
                const float4 xc = make_float4(tv.x - mu.x, tv.y - mu.y, tv.z - mu.z, tv.w - mu.w);
                const float4 xh = make_float4(xc.x * sc.x, xc.y * sc.y, xc.z * sc.z, xc.w * sc.w);
                const float4 dgam = make_float4(g0.x * xh.x, g0.y * xh.y, g0.z * xh.z, g0.w * xh.w);
                const float4 dg2 = make_float4((1.f - a_g) * dgam.x, (1.f - a_g) * dgam.y, (1.f - a_g) * dgam.z, (1.f - a_g) * dgam.w);
                const float4 db2 = make_float4((1.f - a_b) * g0.x, (1.f - a_b) * g0.y, (1.f - a_b) * g0.z, (1.f - a_b) * g0.w);
                st4(dgb2 + p * 2 * g.C + c, dg2);
                st4(dgb2 + p * 2 * g.C + g.C + c, db2);
                if (AMAX) gm = dasr_amax4(dasr_amax4(gm, dg2), db2);
                dag += dgam.x * (g1.x - g2.x) + dgam.y * (g1.y - g2.y) + dgam.z * (g1.z - g2.z) +
                       dgam.w * (g1.w - g2.w);
                dab += g0.x * (b1.x - b2.x) + g0.y * (b1.y - b2.y) + g0.z * (b1.z - b2.z) + g0.w * (b1.w - b2.w);
                G1 = make_float4(a_g * dgam.x, a_g * dgam.y, a_g * dgam.z, a_g * dgam.w);
                B1 = make_float4(a_b * g0.x, a_b * g0.y, a_b * g0.z, a_b * g0.w);
                dbg = f4add(dbg, G1);
                dbb = f4add(dbb, B1);
                float4 dxh;
                dxh.x = g0.x * (1.f + a_g * g1.x + (1.f - a_g) * g2.x);
                dxh.y = g0.y * (1.f + a_g * g1.y + (1.f - a_g) * g2.y);
                dxh.z = g0.z * (1.f + a_g * g1.z + (1.f - a_g) * g2.z);
                dxh.w = g0.w * (1.f + a_g * g1.w + (1.f - a_g) * g2.w);
                st4(dt + p * g.C + c, dxh);
                S1 = f4add(S1, dxh);
                S2.x = fmaf(dxh.x, xc.x, S2.x); S2.y = fmaf(dxh.y, xc.y, S2.y);
                S2.z = fmaf(dxh.z, xc.z, S2.z); S2.w = fmaf(dxh.w, xc.w, S2.w);
            }
#if SB_STEP_ISSUE
            issue_step(next_tile, u, cur);
#endif
            bf16_t* gp = sGb + (ly * SF_TW + lx) * SB16_GST + 4 * cq;
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {  // bf16 pieces of G: value, then the exact residuals (fp32 activations)
                st4(gp + pc * PLANE, G1);      // channels 0..63: gamma part
                st4(gp + pc * PLANE + 64, B1); // channels 64..127: beta part
                if (pc + 1 < NP) {
                    G1 = make_float4(G1.x - round_to<bf16_t>(G1.x), G1.y - round_to<bf16_t>(G1.y),
                                     G1.z - round_to<bf16_t>(G1.z), G1.w - round_to<bf16_t>(G1.w));
                    B1 = make_float4(B1.x - round_to<bf16_t>(B1.x), B1.y - round_to<bf16_t>(B1.y),
                                     B1.z - round_to<bf16_t>(B1.z), B1.w - round_to<bf16_t>(B1.w));
                }
            }
            DASR_SCHED_BARRIER();              // one step's gather rows at a time (256-VGPR budget)
        }
    };
    // ---- phase 2 of one tile: one-hot(region) x G, 32 pixels per MFMA (one tile row per K step).  Called under
    // wave-uniform conditions only (EXEC all ones: ds_read_b64_tr_b16 and the MFMAs need it).
    auto phase2 = [&](const bf16_t* sGb, const unsigned char* sRb) {
        const int i16 = lane & 15, kg = lane >> 4;            // A: region row i16, pixels 8*kg .. 8*kg+7 of the step
        const int tq = (lane & 15) >> 2, tp = lane & 3;       // transposed-read role inside the 16-lane group
#pragma unroll
        for (int s = 0; s < SB_TH; ++s) {
            unsigned my_lo, my_hi, t8_lo, t8_hi;
            lds_ld8_unaligned(sRb + (s + mydy) * (SF_TW + 2) + 8 * kg + mydx, my_lo, my_hi);
            lds_ld8_unaligned(sRb + (s + 2) * (SF_TW + 2) + 8 * kg + 2, t8_lo, t8_hi);
            bf16x8 a_my, a_8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a_my[j] = dasr_f2bf((int)((my_lo >> (8 * j)) & 0xffu) == i16 ? 1.f : 0.f);
                a_my[4 + j] = dasr_f2bf((int)((my_hi >> (8 * j)) & 0xffu) == i16 ? 1.f : 0.f);
                a_8[j] = dasr_f2bf((int)((t8_lo >> (8 * j)) & 0xffu) == i16 ? 1.f : 0.f);
                a_8[4 + j] = dasr_f2bf((int)((t8_hi >> (8 * j)) & 0xffu) == i16 ? 1.f : 0.f);
            }
            const bf16_t* gq0 = sGb + (s * SF_TW + 8 * kg + tq) * SB16_GST + 4 * tp;
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                const bf16_t* gq = gq0 + pc * PLANE;
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) {
                    const bf16x4 lo = lds_read_tr16(gq + 16 * nt), hi = lds_read_tr16(gq + 16 * nt + 4 * SB16_GST);
                    bf16x8 bv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bv[e] = lo[e]; bv[4 + e] = hi[e]; }
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_my, bv, acc[nt], 0, 0, 0);
                }
                // tap 8, N-tile `wv`: its own operand reads and an UNCONDITIONAL MFMA.  (An MFMA ignores EXEC: written
                // as `if (nt == wv) acc[8] = mfma(...)` inside the loop above, hipcc guards it with s_and_saveexec on the
                // per-lane compare and no branch, so every wave executed all eight of them - wrong on the GPU, right on
                // the CPU emulator.  Never put an MFMA under a condition that lives in a VGPR.)
                {
                    const bf16x4 lo = lds_read_tr16(gq + 16 * wv), hi = lds_read_tr16(gq + 16 * wv + 4 * SB16_GST);
                    bf16x8 bv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bv[e] = lo[e]; bv[4 + e] = hi[e]; }
                    acc[8] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_8, bv, acc[8], 0, 0, 0);
                }
            }
        }
    };
    // This workgroup's tiles: blockIdx.x + i * gridDim.x, i = 0 .. n-1.  Iteration i (0 .. n): phase 1 of tile i into
    // buffer i & 1 and phase 2 of tile i-1 from buffer (i-1) & 1, in the order the wave's half dictates; the region bytes
    // of tile i+1 are written to sR[(i+1) % 3] (waves 0-3 only: their loads and the wait on them stay out of the other
    // waves' instruction stream) and become visible at the next barrier.
    const int n = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    auto tile_of = [&](int i) { return (int)blockIdx.x + (i < n ? i : n - 1) * (int)gridDim.x; };   // clamped
    // Order of the two phases inside an iteration.  fp32 (three G planes: the matrix phase is as long as the elementwise
    // one): waves 4-7 - the second wave of each SIMD - run phase 2 first, so the VALU phase of one wave shares the SIMD with
    // the MFMA / LDS phase of the other (253 -> 237 us).  bf16 (one plane, latency-bound): the same order everywhere, so
    // that every wave issues its next loads as early as possible (flipped: 1075-1090 us, not flipped 972).
#ifndef SB_FLIP_MODE
#define SB_FLIP_MODE (sizeof(T) == 2 ? 2 : 0)
#endif
    const bool second_half = SB_FLIP_MODE == 0 ? wv >= 4 : SB_FLIP_MODE == 1 ? (wv & 1) != 0 : false;
    const bool stager = SB_FLIP_MODE == 0 ? wv < 4 : true;
    // Operand prefetch distance: bf16 keeps TWO register sets (20 VGPRs each at two rows per tile) and loads two tiles
    // ahead - the kernel is latency-bound there (bytes in flight per CU / round-trip time); fp32 has no registers left
    // for a second set (242 of 256).  The tile loop is unrolled by two so that the sets have static names.
    constexpr bool DEEP = sizeof(T) == 2 && SB_DEEP_PREFETCH;
    constexpr int DIST = DEEP ? 2 : 1;
    SeanBwdLoads<T, NST> setA, setB;
    unsigned char rnext = (unsigned char)g.K;
    if (n > 0) {
        issue(tile_of(0), setA);
        if (DEEP) issue(tile_of(1), setB);
        if (stager) {
            rnext = issue_r(tile_of(0));
            if ((int)threadIdx.x < NR) sR[threadIdx.x] = rnext;
            rnext = issue_r(tile_of(1));
        }
        auto iteration = [&](int i, SeanBwdLoads<T, NST>& cur) {
            __syncthreads();
            if (stager) {
                if ((int)threadIdx.x < NR) sR[((i + 1) % 3) * NRP + threadIdx.x] = rnext;
                rnext = issue_r(tile_of(i + 2));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if ((h == 0) != second_half) {
                    if (i < n) {
                        phase1(tile_of(i), tile_of(i + DIST), cur, sG + (i & 1) * NP * PLANE, sR + (i % 3) * NRP);
#if !SB_STEP_ISSUE
                        issue(tile_of(i + DIST), cur);   // in flight during the other phase, the barrier (and, bf16, a whole tile)
#endif
                    }
                } else if (i >= 1) {
                    phase2(sG + ((i - 1) & 1) * NP * PLANE, sR + ((i - 1) % 3) * NRP);
                }
            }
        };
        for (int i = 0; i <= n; i += 2) {
            iteration(i, setA);
            if (i + 1 <= n) iteration(i + 1, DEEP ? setB : setA);
        }
    }
    // (here, where every lane of every wave is still on the same path; sred is written only below)
    if (AMAX) dasr_amax_commit(dgb2_amax, gm, sred, dasr_flat_wg(), dasr_flat_nwg());
    // ---- per-channel sums: reduce over the 4 pixel sub-lanes (lanes l, l+16, l+32, l+48), then over the 8 waves
    float vals[18] = {S1.x, S1.y, S1.z, S1.w, S2.x, S2.y, S2.z, S2.w, dbg.x, dbg.y, dbg.z, dbg.w,
                      dbb.x, dbb.y, dbb.z, dbb.w, dag, dab};
#pragma unroll
    for (int q = 0; q < 18; ++q) {
        vals[q] += __shfl_xor(vals[q], 16, 64);
        vals[q] += __shfl_xor(vals[q], 32, 64);
    }
    __syncthreads();
    if (ps == 0) {
#pragma unroll
        for (int q = 0; q < 18; ++q) sred[(wv * 18 + q) * 16 + cq] = vals[q];
    }
    __syncthreads();
    if (wv == 0 && ps == 0) {
        float r[18];
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            r[q] = 0.f;
            for (int w = 0; w < 8; ++w) r[q] += sred[(w * 18 + q) * 16 + cq];
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                atomicAdd(&S[((size_t)b * g.C + c + j) * 2 + 0], r[j]);
                atomicAdd(&S[((size_t)b * g.C + c + j) * 2 + 1], r[4 + j]);
                atomicAdd(&dbias_g[c + j], r[8 + j]);
                atomicAdd(&dbias_b[c + j], r[12 + j]);
            }
        }
        float ra = live ? r[16] : 0.f, rb = live ? r[17] : 0.f;
        for (int off = 8; off > 0; off >>= 1) {
            ra += __shfl_xor(ra, off, 64);
            rb += __shfl_xor(rb, off, 64);
        }
        if (cq == 0) {
            atomicAdd(dalpha_g, ra);
            atomicAdd(dalpha_b, rb);
        }
    }
    // ---- slab [b][blockIdx.x][18][K][C]: D fragment of the 16x16 MFMA: column = lane&15 (channel within the N-tile),
    // row = 4*(lane>>4) + reg (region index)
    float* slab = dD_slabs + ((size_t)b * gridDim.x + blockIdx.x) * 18 * g.K * g.C;
    {
        const int j = lane & 15;
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) {
            const int tap = nt < 8 ? wv : 8;
            const int ntile = nt < 8 ? nt : wv;            // 0..7: gamma channels 0..63 then beta channels 0..63
            const int s = ntile >> 2, cc = c0 + 16 * (ntile & 3) + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * (lane >> 4) + r;
                if (k < g.K && cc < g.C) slab[((size_t)(s * 9 + tap) * g.K + k) * g.C + cc] = acc[nt][r];
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_sean_dD_reduce(const float* __restrict__ slabs, const int* __restrict__ flag,
                                                        float* __restrict__ dD, int per_sample, int nslab, size_t n) {
    if (flag && *flag != 0) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t b = i / per_sample, e = i % per_sample;
        const float* p = slabs + b * (size_t)nslab * per_sample + e;
        float acc = 0.f;
        for (int s = 0; s < nslab; ++s) acc += p[(size_t)s * per_sample];
        dD[i] = acc;
    }
}

// ------------------------------------------------------------------------------------------ host side
static int sean_bwd_blocks_per_sample(int B, int H, int W) {
    int ntiles = ((W + SF_TW - 1) / SF_TW) * ((H + 4 - 1) / 4);
    int n = 256 / B;
    if (n < 1) n = 1;
    if (n > ntiles) n = ntiles;
    return n;
}

template <typename T>
static int sean_fwd_impl(const T* t, const float* mean, const float* var, const T* gb2, const float* mask,
                         const unsigned char* region, const int* onehot_flag, const float* D, const float* bias_g,
                         const float* bias_b, const float* alpha_g, const float* alpha_b, const T* residual,
                         T* out, int relu, int B, int H, int W, int C, int K, float eps, void* stream, float* out_amax = nullptr) {
    DASR_CHECK_PTR(t); DASR_CHECK_PTR(mean); DASR_CHECK_PTR(var); DASR_CHECK_PTR(gb2); DASR_CHECK_PTR(mask);
    DASR_CHECK_PTR(D); DASR_CHECK_PTR(bias_g); DASR_CHECK_PTR(bias_b); DASR_CHECK_PTR(alpha_g); DASR_CHECK_PTR(alpha_b);
    DASR_CHECK_PTR(out);
    if (region == nullptr && onehot_flag != nullptr) return DASR_E_NULL;
    DASR_CHECK_SHAPE(B > 0 && H > 0 && W > 0 && C > 0 && K > 0);
    if (K > SEAN_MAXK) return DASR_E_UNSUPPORTED;
    SeanGeom g{B, H, W, C, K};
    const bool fast = region != nullptr && (C % 4) == 0;
    const bool fast_only = fast && onehot_flag == nullptr;   // the caller vouches for one-hot masks
    // max |out|: kept by the gather kernel itself when it is the only one launched (at most 768 workgroups); otherwise a
    // pass over `out` afterwards (which of the two kernels works is decided on the device)
    const bool fused_amax = sizeof(T) == 4 && out_amax != nullptr && fast_only;
    if (fast) {
        int tiles = B * ((W + SF_TW - 1) / SF_TW) * ((H + SF_TH - 1) / SF_TH);
        int slices = (int)dasr_cdiv(C, 64);
        int nwg = (residual ? 256 * DASR_SEAN_RES_OCC : 768) / slices;   // 256 CUs x 3 (2) resident workgroups (51 KB of LDS each)
        if (nwg < 1) nwg = 1;
        if (nwg > tiles) nwg = tiles;
        const int per = ((tiles / nwg) << 16) | (tiles % nwg);      // (base, rem), see the kernel; rem < nwg <= 768
        size_t lds = sizeof(float) * (size_t)(18 * (K + 1) * 64) + (SF_TH + 2) * (SF_TW + 2);
        // bf16 activations: eight channels per lane (C % 8 == 0); dasr_set_conv_bf16_impl(+ 2048): the 4-channel form (A/B, tests)
        const bool w8 = sizeof(T) == 2 && (C % 8) == 0 && (dasr_get_conv_bf16_impl() & 2048) == 0;
#define SEAN_FWD_GO(RELU, RES)                                                                                     \
    do {                                                                                                           \
        if (w8)                                                                                                    \
            DASR_LAUNCH((k_sean_fwd_onehot_w8<RELU, RES>), dim3(nwg, slices), dim3(256), lds, stream, g, (const bf16_t*)t, \
                        mean, var, (const bf16_t*)gb2, region, onehot_flag, D, bias_g, bias_b, alpha_g, alpha_b,      \
                        (const bf16_t*)residual, (bf16_t*)out, eps, per);                                            \
        else if (fused_amax)                                                                                            \
            DASR_LAUNCH((k_sean_fwd_onehot<RELU, RES, T, sizeof(T) == 4>), dim3(nwg, slices), dim3(256), lds, stream, g, t, \
                        mean, var, gb2, region, onehot_flag, D, bias_g, bias_b, alpha_g, alpha_b, residual, out, eps, per, out_amax); \
        else                                                                                                       \
            DASR_LAUNCH((k_sean_fwd_onehot<RELU, RES, T>), dim3(nwg, slices), dim3(256), lds, stream, g, t,        \
                        mean, var, gb2, region, onehot_flag, D, bias_g, bias_b, alpha_g, alpha_b, residual, out, eps, per, (float*)nullptr); \
    } while (0)
        if (relu) { if (residual) SEAN_FWD_GO(true, true); else SEAN_FWD_GO(true, false); }
        else      { if (residual) SEAN_FWD_GO(false, true); else SEAN_FWD_GO(false, false); }
#undef SEAN_FWD_GO
    }
    if (!fast_only) {
        int tiles = ((W + SEAN_TW - 1) / SEAN_TW) * ((H + SEAN_TH - 1) / SEAN_TH);
        size_t lds = sizeof(float) * (size_t)(2 * 9 * K * 64 + K * (SEAN_TH + 2) * (SEAN_TW + 2));
        DASR_LAUNCH((k_sean_fwd<T>), dim3(tiles, B, dasr_cdiv(C, 64)), dim3(256), lds, stream, g, t, mean, var, gb2, mask, D,
                    bias_g, bias_b, alpha_g, alpha_b, residual, out, relu, eps,
                    fast ? onehot_flag : (const int*)nullptr);
    }
    if (sizeof(T) == 4 && out_amax && !fused_amax)
        return absmax_raise((const float*)out, (size_t)B * H * W * C, out_amax, stream);
    DASR_RETURN_LAUNCH_STATUS();
}

extern "C" int dasr_sean_fwd(const float* t, const float* mean, const float* var, const float* gb2, const float* mask,
                             const unsigned char* region, const int* onehot_flag, const float* D, const float* bias_g,
                             const float* bias_b, const float* alpha_g, const float* alpha_b, const float* residual,
                             float* out, float* out_amax, int relu, int B, int H, int W, int C, int K, float eps, void* stream) {
    return sean_fwd_impl<float>(t, mean, var, gb2, mask, region, onehot_flag, D, bias_g, bias_b, alpha_g, alpha_b, residual,
                                out, relu, B, H, W, C, K, eps, stream, out_amax);
}
extern "C" int dasr_sean_fwd_bf16(const unsigned short* t, const float* mean, const float* var, const unsigned short* gb2,
                                  const float* mask, const unsigned char* region, const int* onehot_flag, const float* D,
                                  const float* bias_g, const float* bias_b, const float* alpha_g, const float* alpha_b,
                                  const unsigned short* residual, unsigned short* out, int relu, int B, int H, int W, int C,
                                  int K, float eps, void* stream) {
    return sean_fwd_impl<bf16_t>((const bf16_t*)t, mean, var, (const bf16_t*)gb2, mask, region, onehot_flag, D, bias_g,
                                 bias_b, alpha_g, alpha_b, (const bf16_t*)residual, (bf16_t*)out, relu, B, H, W, C, K, eps,
                                 stream);
}

__global__ void __launch_bounds__(256) k_sean_bwd_zero(float* __restrict__ S, int nS, float* __restrict__ dbg,
                                                       float* __restrict__ dbb, int C, float* __restrict__ dag,
                                                       float* __restrict__ dab) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nS) S[i] = 0.f;
    else if (i < nS + C) dbg[i - nS] = 0.f;
    else if (i < nS + 2 * C) dbb[i - nS - C] = 0.f;
    else if (i == nS + 2 * C) dag[0] = 0.f;
    else if (i == nS + 2 * C + 1) dab[0] = 0.f;
}

// LDS bytes of the general (soft-mask) backward kernel: D tables + mask tile + dD accumulators + reduction scratch
static size_t sean_bwd_general_lds(int K) {
    return sizeof(float) * (size_t)(2 * (2 * 9 * K * 64) + K * (SEAN_TH + 2) * (SEAN_TW + 2) + 6 * 256);
}
// Largest region count the soft-mask kernels take forward AND backward (the backward keeps two [2][9][K][64] tables in
// the CU's 160 KB of LDS: K <= 14); one-hot masks go up to SEAN_MAXK = 16.
extern "C" int dasr_sean_soft_mask_max_regions(void) {
    int k = SEAN_MAXK;
    while (k > 1 && sean_bwd_general_lds(k) > 160 * 1024) --k;
    return k;
}

extern "C" size_t dasr_sean_bwd_workspace(int B, int H, int W, int C, int K) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0) return 0;
    size_t S = 2 * (size_t)B * C;
    size_t slabs = (size_t)B * sean_bwd_blocks_per_sample(B, H, W) * 18 * K * C;
    return sizeof(float) * (S + slabs);
}

template <typename T>
static int sean_bwd_impl(const T* dout, const T* out, const T* t, const float* mean, const float* var,
                         const T* gb2, const float* mask, const unsigned char* region, const int* onehot_flag,
                         const float* D, const float* bias_g, const float* bias_b, const float* alpha_g,
                         const float* alpha_b, T* dt, T* dgb2, float* dD, float* dbias_g, float* dbias_b,
                         float* dalpha_g, float* dalpha_b, T* dres, void* workspace, size_t workspace_bytes,
                         int relu, int B, int H, int W, int C, int K, float eps, void* stream, float* dt_amax = nullptr,
                         float* dgb2_amax = nullptr) {
    DASR_CHECK_PTR(dout); DASR_CHECK_PTR(out); DASR_CHECK_PTR(t); DASR_CHECK_PTR(mean); DASR_CHECK_PTR(var);
    DASR_CHECK_PTR(gb2); DASR_CHECK_PTR(mask); DASR_CHECK_PTR(D); DASR_CHECK_PTR(bias_g); DASR_CHECK_PTR(bias_b);
    DASR_CHECK_PTR(alpha_g); DASR_CHECK_PTR(alpha_b); DASR_CHECK_PTR(dt); DASR_CHECK_PTR(dgb2); DASR_CHECK_PTR(dD);
    DASR_CHECK_PTR(dbias_g); DASR_CHECK_PTR(dbias_b); DASR_CHECK_PTR(dalpha_g); DASR_CHECK_PTR(dalpha_b);
    DASR_CHECK_PTR(workspace);
    if (region == nullptr && onehot_flag != nullptr) return DASR_E_NULL;
    DASR_CHECK_SHAPE(B > 0 && H > 0 && W > 0 && C > 0 && K > 0);
    if (K > SEAN_MAXK) return DASR_E_UNSUPPORTED;
    if (workspace_bytes < dasr_sean_bwd_workspace(B, H, W, C, K)) return DASR_E_WORKSPACE;
    SeanGeom g{B, H, W, C, K};
    hipStream_t st = (hipStream_t)stream;
    float* S = (float*)workspace;
    float* slabs = S + 2 * (size_t)B * C;
    const bool fast = region != nullptr && (C % 4) == 0;
    const bool fast_only = fast && onehot_flag == nullptr;
    if (!fast_only && K > dasr_sean_soft_mask_max_regions()) return DASR_E_UNSUPPORTED;
    // the accumulators the kernels add into with atomics: one launch instead of five memsets; dD is only accumulated
    // into by the general kernel (the one-hot path overwrites all of it in k_sean_dD_reduce)
    (void)st;
    DASR_LAUNCH(k_sean_bwd_zero, dim3(dasr_cdiv((size_t)2 * B * C + 2 * C + 2, 256)), dim3(256), 0, stream, S, 2 * B * C,
                dbias_g, dbias_b, C, dalpha_g, dalpha_b);
    if (!fast_only) {
        hipError_t e = hipMemsetAsync(dD, 0, sizeof(float) * (size_t)B * 18 * K * C, st);
        if (e != hipSuccess) return (int)e;
    }
    if (fast) {
        int nblk = sean_bwd_blocks_per_sample(B, H, W);
        constexpr int NP = SeanBwdCfg<T>::NP;
        // tile rows (bf16 / fp32 with its three G planes): as many as the two G buffers leave room for beside the D table
#ifndef SB_TH_BF16
#define SB_TH_BF16 2
#endif
        constexpr int TH_BIG = sizeof(T) == 2 ? SB_TH_BF16 : 1, TH_SMALL = sizeof(T) == 2 ? 2 : 1;
        const bool big = sean_bwd_lds_bytes(K, TH_BIG, NP) <= 160 * 1024;
        const int TH = big ? TH_BIG : TH_SMALL;
        int ntiles = ((W + SF_TW - 1) / SF_TW) * ((H + TH - 1) / TH);
        size_t lds = (size_t)sean_bwd_lds_bytes(K, TH, NP);
        if (lds > 160 * 1024) return DASR_E_UNSUPPORTED;
        constexpr bool F32 = sizeof(T) == 4;
        if (F32 && dgb2_amax && fast_only && big)
            DASR_LAUNCH((k_sean_bwd_a_onehot<T, TH_BIG, F32>), dim3(nblk, B, dasr_cdiv(C, 64)), dim3(512), lds, stream, g, dout, out, t,
                        mean, var, gb2, region, onehot_flag, D, bias_g, bias_b, alpha_g, alpha_b, dt, dgb2, slabs, dbias_g,
                        dbias_b, dalpha_g, dalpha_b, dres, S, relu, eps, ntiles, dgb2_amax);
        else if (F32 && dgb2_amax && fast_only)
            DASR_LAUNCH((k_sean_bwd_a_onehot<T, TH_SMALL, F32>), dim3(nblk, B, dasr_cdiv(C, 64)), dim3(512), lds, stream, g, dout, out, t,
                        mean, var, gb2, region, onehot_flag, D, bias_g, bias_b, alpha_g, alpha_b, dt, dgb2, slabs, dbias_g,
                        dbias_b, dalpha_g, dalpha_b, dres, S, relu, eps, ntiles, dgb2_amax);
        else if (big)
            DASR_LAUNCH((k_sean_bwd_a_onehot<T, TH_BIG>), dim3(nblk, B, dasr_cdiv(C, 64)), dim3(512), lds, stream, g, dout, out, t,
                        mean, var, gb2, region, onehot_flag, D, bias_g, bias_b, alpha_g, alpha_b, dt, dgb2, slabs, dbias_g,
                        dbias_b, dalpha_g, dalpha_b, dres, S, relu, eps, ntiles, (float*)nullptr);
        else
            DASR_LAUNCH((k_sean_bwd_a_onehot<T, TH_SMALL>), dim3(nblk, B, dasr_cdiv(C, 64)), dim3(512), lds, stream, g, dout, out, t,
                        mean, var, gb2, region, onehot_flag, D, bias_g, bias_b, alpha_g, alpha_b, dt, dgb2, slabs, dbias_g,
                        dbias_b, dalpha_g, dalpha_b, dres, S, relu, eps, ntiles, (float*)nullptr);
        size_t n = (size_t)B * 18 * K * C;
        DASR_LAUNCH(k_sean_dD_reduce, dim3(dasr_ew_grid(n)), dim3(256), 0, stream, (const float*)slabs, onehot_flag, dD,
                    18 * K * C, nblk, n);
    }
    if (!fast_only) {
        int tiles = ((W + SEAN_TW - 1) / SEAN_TW) * ((H + SEAN_TH - 1) / SEAN_TH);
        size_t lds = sean_bwd_general_lds(K);
        DASR_LAUNCH((k_sean_bwd_a<T>), dim3(tiles, B, dasr_cdiv(C, 64)), dim3(256), lds, stream, g, dout, out, t, mean, var,
                    gb2, mask, D, bias_g, bias_b, alpha_g, alpha_b, dt, dgb2, dD, dbias_g, dbias_b, dalpha_g, dalpha_b,
                    dres, S, relu, eps, fast ? onehot_flag : (const int*)nullptr);
    }
    if (sizeof(T) == 4 && dgb2_amax && !fast_only) {       // (which kernel wrote dgb2 was decided on the device)
        int rc = absmax_raise((const float*)dgb2, (size_t)B * H * W * 2 * C, dgb2_amax, stream);
        if (rc) return rc;
    }
    size_t n = (size_t)B * H * W * C;
    if ((C & 3) == 0 && (256 % (C / 4)) == 0 && B <= 65535) {
        const int npl = 256 / (C / 4);
        unsigned gx = dasr_cdiv((size_t)H * W, npl * 8);     // eight pixels per thread
        if (gx < 1) gx = 1;
        if (dt_amax && (size_t)gx * B > DASR_AMAX_MAX_PARTS) gx = DASR_AMAX_MAX_PARTS / B;   // (one partial maximum per workgroup)
        DASR_LAUNCH((k_sean_bwd_b_rows<T>), dim3(gx, B), dim3(256), 0, stream, t, mean, var, S, dt, H * W, C, eps,
                    sizeof(T) == 4 ? dt_amax : (float*)nullptr);
    } else {
        DASR_LAUNCH((k_sean_bwd_b<T>), dim3(dasr_ew_grid((C & 3) == 0 ? n / 4 : n)), dim3(256), 0, stream, t, mean, var, S,
                    dt, H * W, C, n, eps, sizeof(T) == 4 ? dt_amax : (float*)nullptr);
    }
    DASR_RETURN_LAUNCH_STATUS();
}

extern "C" int dasr_sean_bwd(const float* dout, const float* out, const float* t, const float* mean, const float* var,
                             const float* gb2, const float* mask, const unsigned char* region, const int* onehot_flag,
                             const float* D, const float* bias_g, const float* bias_b, const float* alpha_g,
                             const float* alpha_b, float* dt, float* dgb2, float* dD, float* dbias_g, float* dbias_b,
                             float* dalpha_g, float* dalpha_b, float* dres, float* dt_amax, float* dgb2_amax,
                             void* workspace, size_t workspace_bytes,
                             int relu, int B, int H, int W, int C, int K, float eps, void* stream) {
    return sean_bwd_impl<float>(dout, out, t, mean, var, gb2, mask, region, onehot_flag, D, bias_g, bias_b, alpha_g, alpha_b,
                                dt, dgb2, dD, dbias_g, dbias_b, dalpha_g, dalpha_b, dres, workspace, workspace_bytes, relu, B,
                                H, W, C, K, eps, stream, dt_amax, dgb2_amax);
}
extern "C" int dasr_sean_bwd_bf16(const unsigned short* dout, const unsigned short* out, const unsigned short* t,
                                  const float* mean, const float* var, const unsigned short* gb2, const float* mask,
                                  const unsigned char* region, const int* onehot_flag, const float* D, const float* bias_g,
                                  const float* bias_b, const float* alpha_g, const float* alpha_b, unsigned short* dt,
                                  unsigned short* dgb2, float* dD, float* dbias_g, float* dbias_b, float* dalpha_g,
                                  float* dalpha_b, unsigned short* dres, void* workspace, size_t workspace_bytes, int relu,
                                  int B, int H, int W, int C, int K, float eps, void* stream) {
    return sean_bwd_impl<bf16_t>((const bf16_t*)dout, (const bf16_t*)out, (const bf16_t*)t, mean, var, (const bf16_t*)gb2,
                                 mask, region, onehot_flag, D, bias_g, bias_b, alpha_g, alpha_b, (bf16_t*)dt, (bf16_t*)dgb2, dD,
                                 dbias_g, dbias_b, dalpha_g, dalpha_b, (bf16_t*)dres, workspace, workspace_bytes, relu, B, H, W,
                                 C, K, eps, stream);
}
