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

__global__ void __launch_bounds__(256) k_sean_fwd(SeanGeom g, const float* __restrict__ t,
                                                  const float* __restrict__ mean, const float* __restrict__ var,
                                                  const float* __restrict__ gb2, const float* __restrict__ mask,
                                                  const float* __restrict__ D, const float* __restrict__ bias_g,
                                                  const float* __restrict__ bias_b, const float* __restrict__ alpha_g,
                                                  const float* __restrict__ alpha_b, const float* __restrict__ residual,
                                                  float* __restrict__ out, int relu, float eps) {
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
        float g2 = gb2[p * 2 * g.C + c], b2 = gb2[p * 2 * g.C + g.C + c];
        float gam = a_g * g1 + (1.f - a_g) * g2;
        float bet = a_b * b1 + (1.f - a_b) * b2;
        float xh = (t[p * g.C + c] - mu) * s;
        float o = xh * (1.f + gam) + bet;
        if (residual) o += residual[p * g.C + c];
        if (relu) o = o > 0.f ? o : 0.f;
        out[p * g.C + c] = o;
    }
}

extern "C" int dasr_sean_fwd(const float* t, const float* mean, const float* var, const float* gb2, const float* mask,
                             const float* D, const float* bias_g, const float* bias_b, const float* alpha_g,
                             const float* alpha_b, const float* residual, float* out, int relu, int B, int H, int W,
                             int C, int K, float eps, void* stream) {
    DASR_CHECK_PTR(t); DASR_CHECK_PTR(mean); DASR_CHECK_PTR(var); DASR_CHECK_PTR(gb2); DASR_CHECK_PTR(mask);
    DASR_CHECK_PTR(D); DASR_CHECK_PTR(bias_g); DASR_CHECK_PTR(bias_b); DASR_CHECK_PTR(alpha_g); DASR_CHECK_PTR(alpha_b);
    DASR_CHECK_PTR(out);
    DASR_CHECK_SHAPE(B > 0 && H > 0 && W > 0 && C > 0 && K > 0);
    if (K > SEAN_MAXK) return DASR_E_UNSUPPORTED;
    SeanGeom g{B, H, W, C, K};
    int tiles = ((W + SEAN_TW - 1) / SEAN_TW) * ((H + SEAN_TH - 1) / SEAN_TH);
    size_t lds = sizeof(float) * (size_t)(2 * 9 * K * 64 + K * (SEAN_TH + 2) * (SEAN_TW + 2));
    DASR_LAUNCH(k_sean_fwd, dim3(tiles, B, dasr_cdiv(C, 64)), dim3(256), lds, stream, g, t, mean, var, gb2, mask, D,
                bias_g, bias_b, alpha_g, alpha_b, residual, out, relu, eps);
    DASR_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------- backward
// Pass A (per tile): g0 = dout*relu'(out); dgb2, dres; dxhat -> dt (temporarily); per-(b,c) sums
//   S1 = sum dxhat, S2 = sum dxhat*(t-mean); dbias_s; dalpha; dD (LDS accumulation, then one float
//   atomic per (tap,k,c) per workgroup).
// Pass B (elementwise): dt = s*(dxhat - S1/N) + s'(var)*(2/N)*(t-mean)*S2.
// workspace: S [B][C][2] floats.
__global__ void __launch_bounds__(256) k_sean_bwd_a(SeanGeom g, const float* __restrict__ dout,
                                                    const float* __restrict__ out, const float* __restrict__ t,
                                                    const float* __restrict__ mean, const float* __restrict__ var,
                                                    const float* __restrict__ gb2, const float* __restrict__ mask,
                                                    const float* __restrict__ D, const float* __restrict__ bias_g,
                                                    const float* __restrict__ bias_b,
                                                    const float* __restrict__ alpha_g,
                                                    const float* __restrict__ alpha_b, float* __restrict__ dt,
                                                    float* __restrict__ dgb2, float* __restrict__ dD,
                                                    float* __restrict__ dbias_g, float* __restrict__ dbias_b,
                                                    float* __restrict__ dalpha_g, float* __restrict__ dalpha_b,
                                                    float* __restrict__ dres, float* __restrict__ S, int relu,
                                                    float eps) {
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
            float g0 = dout[p * g.C + c];
            if (relu && !(out[p * g.C + c] > 0.f)) g0 = 0.f;
            if (dres) dres[p * g.C + c] = g0;
            float g1, b1;
            sean_dynconv(sD, sM, K, ly, lx, cl, g1, b1);
            g1 += bg;
            b1 += bb;
            float g2 = gb2[p * 2 * g.C + c], b2 = gb2[p * 2 * g.C + g.C + c];
            float gam = a_g * g1 + (1.f - a_g) * g2;
            float xc = t[p * g.C + c] - mu;
            float xh = xc * s;
            float dgam = g0 * xh, dbet = g0;
            dgb2[p * 2 * g.C + c] = (1.f - a_g) * dgam;
            dgb2[p * 2 * g.C + g.C + c] = (1.f - a_b) * dbet;
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
            dt[p * g.C + c] = dxh;
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

__global__ void __launch_bounds__(256) k_sean_bwd_b(const float* __restrict__ t, const float* __restrict__ mean,
                                                    const float* __restrict__ var, const float* __restrict__ S,
                                                    float* __restrict__ dt, int HW, int C, size_t n, float eps) {
    float invN = 1.0f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        size_t b = i / ((size_t)C * HW);
        size_t bc = b * C + c;
        float v = var[bc];
        float s = dasr_double_in_scale(v, eps), ds = dasr_double_in_dscale(v, eps);
        float xc = t[i] - mean[bc];
        dt[i] = s * (dt[i] - S[bc * 2] * invN) + ds * 2.f * invN * xc * S[bc * 2 + 1];
    }
}

extern "C" size_t dasr_sean_bwd_workspace(int B, int H, int W, int C, int K) {
    (void)H; (void)W; (void)K;
    return sizeof(float) * 2 * (size_t)B * C;
}

extern "C" int dasr_sean_bwd(const float* dout, const float* out, const float* t, const float* mean, const float* var,
                             const float* gb2, const float* mask, const float* D, const float* bias_g,
                             const float* bias_b, const float* alpha_g, const float* alpha_b, float* dt, float* dgb2,
                             float* dD, float* dbias_g, float* dbias_b, float* dalpha_g, float* dalpha_b, float* dres,
                             void* workspace, size_t workspace_bytes, int relu, int B, int H, int W, int C, int K,
                             float eps, void* stream) {
    DASR_CHECK_PTR(dout); DASR_CHECK_PTR(out); DASR_CHECK_PTR(t); DASR_CHECK_PTR(mean); DASR_CHECK_PTR(var);
    DASR_CHECK_PTR(gb2); DASR_CHECK_PTR(mask); DASR_CHECK_PTR(D); DASR_CHECK_PTR(bias_g); DASR_CHECK_PTR(bias_b);
    DASR_CHECK_PTR(alpha_g); DASR_CHECK_PTR(alpha_b); DASR_CHECK_PTR(dt); DASR_CHECK_PTR(dgb2); DASR_CHECK_PTR(dD);
    DASR_CHECK_PTR(dbias_g); DASR_CHECK_PTR(dbias_b); DASR_CHECK_PTR(dalpha_g); DASR_CHECK_PTR(dalpha_b);
    DASR_CHECK_PTR(workspace);
    DASR_CHECK_SHAPE(B > 0 && H > 0 && W > 0 && C > 0 && K > 0);
    if (K > SEAN_MAXK) return DASR_E_UNSUPPORTED;
    if (workspace_bytes < dasr_sean_bwd_workspace(B, H, W, C, K)) return DASR_E_WORKSPACE;
    SeanGeom g{B, H, W, C, K};
    hipStream_t st = (hipStream_t)stream;
    float* S = (float*)workspace;
    hipError_t e;
    if ((e = hipMemsetAsync(S, 0, sizeof(float) * 2 * (size_t)B * C, st)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(dD, 0, sizeof(float) * (size_t)B * 18 * K * C, st)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(dbias_g, 0, sizeof(float) * (size_t)C, st)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(dbias_b, 0, sizeof(float) * (size_t)C, st)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(dalpha_g, 0, sizeof(float), st)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(dalpha_b, 0, sizeof(float), st)) != hipSuccess) return (int)e;
    int tiles = ((W + SEAN_TW - 1) / SEAN_TW) * ((H + SEAN_TH - 1) / SEAN_TH);
    size_t lds = sizeof(float) * (size_t)(2 * (2 * 9 * K * 64) + K * (SEAN_TH + 2) * (SEAN_TW + 2) + 6 * 256);
    DASR_LAUNCH(k_sean_bwd_a, dim3(tiles, B, dasr_cdiv(C, 64)), dim3(256), lds, stream, g, dout, out, t, mean, var, gb2,
                mask, D, bias_g, bias_b, alpha_g, alpha_b, dt, dgb2, dD, dbias_g, dbias_b, dalpha_g, dalpha_b, dres, S, relu, eps);
    size_t n = (size_t)B * H * W * C;
    DASR_LAUNCH(k_sean_bwd_b, dim3(dasr_ew_grid(n)), dim3(256), 0, stream, t, mean, var, S, dt, H * W, C, n, eps);
    DASR_RETURN_LAUNCH_STATUS();
}
