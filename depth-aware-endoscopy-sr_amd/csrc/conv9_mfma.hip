// conv9_mfma.hip — the generator's output convolution: 9x9, pad 4, Cin = 32 -> Cout = 3 at HR resolution
// (reference: conv_output, sftmd_arch.py:910,948), forward / dgrad / wgrad on the fp32 matrix cores.
//
// With only 3 output channels a plain implicit GEMM (N = Cout) would use 3/32 of an MFMA tile.  Instead
// one kernel axis is folded into N:
//   forward:  P[q][(kw,co)] = sum_{kh,ci} x[q.y+kh-4, q.x][ci] * w[kh][kw][ci][co]        M = pixels q, N = 27, K = 288
//             y[p][co]      = sum_kw P[(p.y, p.x+kw-4)][(kw,co)]                            (9-term shift-add via LDS)
//   dgrad:    dx[q][ci]     = sum_kh sum_{k'} dyrow(q.y-kh+4)[3*(q.x-4) + k'] * Wd[kh][k'][ci]   M = pixels, N = 32, K = 9*28
//             with k' = 3*(8-kw) + co running over the 27 consecutive floats of the 3-channel dy row
//   wgrad:    dW[kh][(8-kw)*3+co][ci] = sum_q x[q][ci] * dyrow(q.y-kh+4)[3*q.x + n']      M = ci, N = 27, K = pixels
// so every MFMA runs at 27/32 (fwd, wgrad) or 27/28 (dgrad) useful width.  dy has 3 channels, so its
// haloed tile is tiny (14 KB) and consecutive lanes read it at a 3-float stride (conflict-free: gcd(3,32)=1).
#include "dasr_common.h"
#include "bf16.h"
#include "conv_kernels.h"

#define C9_TH 8           // tile rows
#define C9_TQ 64          // tile columns (forward: includes the 8-column halo; 56 outputs per tile)
#define C9_CK 8            // channels per K chunk: 63 KB of LDS per workgroup -> two workgroups per CU
#define C9_CKP 12          // LDS pixel stride in floats (conflict-free ds_read_b128 at a 48-byte stride)
#define C9_PST 28         // P row stride (27 used)
#define C9_DYW ((C9_TQ + 8) * 3 + 8)   // dy tile row stride in floats: 72 px * 3 ch + pad

// TX (template parameter of the kernels) = storage type of the Cin-channel activation (x, dx, mask_src): float, or
// bf16_t on the mixed-precision path.  The 3-channel image side (y, dy), the kernel and the arithmetic (exact-fp32
// MFMA) are fp32 in both: this convolution produces the network's output.
struct Conv9Args {
    const void* x;       // fwd/wgrad: [B,H,W,Cin] (TX) ; dgrad: unused
    const float* w;      // HWIO [9][9][Cin][Cout]
    const float* bias;   // fwd
    const float* dy;     // dgrad/wgrad: [B,H,W,Cout]
    void* out;           // fwd: y [B,H,W,Cout] (float); dgrad: dx [B,H,W,Cin] (TX); wgrad: slabs (float)
    int B, H, W, Cin, Cout;
    int accumulate, P, ntiles;
    const void* mask_src;    // dgrad: see ConvMfmaArgs::mask_src (conv_mfma.hip) (TX)
    int mask_act, unps_r;
};

// ------------------------------------------------------------------------------------------ forward
template <typename TX>
__global__ void __launch_bounds__(256) k_conv9x9_fwd_mfma(Conv9Args a) {
    DASR_DYN_SMEM(smem);
    float* sIn = (float*)smem;                                    // [16][64][CKP]  (later: P [8][64][PST])
    float* sW = sIn + (C9_TH + 8) * C9_TQ * C9_CKP;               // [9][32][CKP]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int TWO = C9_TQ - 8;
    const int tiles_x = (a.W + TWO - 1) / TWO;
    const int x0 = (blockIdx.x % tiles_x) * TWO, y0 = (blockIdx.x / tiles_x) * C9_TH, b = blockIdx.y;
    const int NN = 9 * a.Cout;   // <= 27

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int c0 = 0; c0 < a.Cin; c0 += C9_CK) {
        // all global loads of the chunk are issued before the barrier and the first LDS write ("load; wait; write"
        // loops are chains of dependent round trips: 17 per chunk here)
        constexpr int NIN9 = (C9_TH + 8) * C9_TQ * (C9_CK / 4) / 256, NW9 = 9 * 32 * C9_CK / 256;
        float4 vin[NIN9];
        float vw[NW9];
#pragma unroll
        for (int u = 0; u < NIN9; ++u) {
            const int idx = tid + 256 * u;
            const int q4 = idx % (C9_CK / 4), pix = idx / (C9_CK / 4);
            const int gy = y0 - 4 + pix / C9_TQ, gx = x0 - 4 + pix % C9_TQ;
            vin[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                vin[u] = ld4((const TX*)a.x + (((size_t)b * a.H + gy) * a.W + gx) * a.Cin + c0 + 4 * q4);
        }
#pragma unroll
        for (int u = 0; u < NW9; ++u) {
            const int idx = tid + 256 * u;
            const int k = idx % C9_CK, n = (idx / C9_CK) % 32, kh = idx / (C9_CK * 32);
            vw[u] = 0.f;
            if (n < NN) {
                const int kw = n / a.Cout, co = n % a.Cout;
                vw[u] = a.w[(((size_t)kh * 9 + kw) * a.Cin + c0 + k) * a.Cout + co];
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NIN9; ++u) {
            const int idx = tid + 256 * u;
            *(float4*)(sIn + (idx / (C9_CK / 4)) * C9_CKP + 4 * (idx % (C9_CK / 4))) = vin[u];
        }
#pragma unroll
        for (int u = 0; u < NW9; ++u) {
            const int idx = tid + 256 * u;
            sW[((idx / (C9_CK * 32)) * 32 + (idx / C9_CK) % 32) * C9_CKP + idx % C9_CK] = vw[u];
        }
        __syncthreads();
#pragma unroll
        for (int kh = 0; kh < 9; ++kh) {
#pragma unroll
            for (int q = 0; q < C9_CK / 8; ++q) {
                const float4 Bf = *(const float4*)(sW + (kh * 32 + li) * C9_CKP + 8 * q + 4 * lh);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int r = 2 * wv + (t >> 1), half = t & 1;
                    const float4 A = *(const float4*)(sIn + ((r + kh) * C9_TQ + 32 * half + li) * C9_CKP + 8 * q + 4 * lh);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, Bf.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, Bf.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, Bf.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, Bf.w, acc[t], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();
    float* sP = sIn;   // [8][64][PST]
    if (li < C9_PST) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = 2 * wv + (t >> 1), half = t & 1;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                int qx = 32 * half + (g & 3) + 8 * (g >> 2) + 4 * lh;
                sP[(r * C9_TQ + qx) * C9_PST + li] = acc[t][g];
            }
        }
    }
    __syncthreads();
    const int nout = C9_TH * TWO * a.Cout;
    for (int idx = tid; idx < nout; idx += 256) {
        int co = idx % a.Cout, ox = (idx / a.Cout) % TWO, r = idx / (a.Cout * TWO);
        int gy = y0 + r, gx = x0 + ox;
        if (gy >= a.H || gx >= a.W) continue;
        float v = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int kw = 0; kw < 9; ++kw) v += sP[(r * C9_TQ + ox + kw) * C9_PST + kw * a.Cout + co];
        ((float*)a.out)[(((size_t)b * a.H + gy) * a.W + gx) * a.Cout + co] = v;
    }
}

// ------------------------------------------------------------------------------------------ dgrad
// tile: 8 rows x 64 columns of dx pixels x 32 input channels (blockIdx.z selects the 32-channel slice)
#define C9_NDY (((C9_TH + 8) * C9_DYW + 255) / 256)
__device__ __forceinline__ void c9_load_dy(const Conv9Args& a, float (&v)[C9_NDY], int b, int y0, int x0, int tid) {
    // rows y0-4 .. y0+11, columns x0-4 .. x0+67, Cout channels interleaved; zero outside the image
    const int rowf = (C9_TQ + 8) * a.Cout;
#pragma unroll
    for (int u = 0; u < C9_NDY; ++u) {
        const int idx = tid + 256 * u;
        const int f = idx % C9_DYW, ry = idx / C9_DYW;
        v[u] = 0.f;
        if (idx < (C9_TH + 8) * C9_DYW && f < rowf) {
            const int gy = y0 - 4 + ry, gx = x0 - 4 + f / a.Cout, co = f % a.Cout;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                v[u] = a.dy[(((size_t)b * a.H + gy) * a.W + gx) * a.Cout + co];
        }
    }
}
__device__ __forceinline__ void c9_store_dy(float* sDy, const float (&v)[C9_NDY], int tid) {
#pragma unroll
    for (int u = 0; u < C9_NDY; ++u) {
        const int idx = tid + 256 * u;
        if (idx < (C9_TH + 8) * C9_DYW) sDy[idx] = v[u];
    }
}

template <bool UNMASK, typename TX>
__global__ void __launch_bounds__(256) k_conv9x9_dgrad_mfma(Conv9Args a) {
    DASR_DYN_SMEM(smem);
    float* sDy = (float*)smem;                       // [16][DYW]
    float* sW = sDy + (C9_TH + 8) * C9_DYW;          // [9][28][32]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int tiles_x = (a.W + C9_TQ - 1) / C9_TQ;
    const int x0 = (blockIdx.x % tiles_x) * C9_TQ, y0 = (blockIdx.x / tiles_x) * C9_TH, b = blockIdx.y;
    const int n0 = blockIdx.z * 32;
    const int KK = 9 * a.Cout;   // 27 live k' per kh
    {
        float vdy[C9_NDY];
        c9_load_dy(a, vdy, b, y0, x0, tid);
        constexpr int NWD = (9 * 28 * 32 + 255) / 256;       // 32 (the last one half used)
        float vw[NWD];
#pragma unroll
        for (int u = 0; u < NWD; ++u) {
            const int idx = tid + 256 * u;
            const int ci = idx & 31, kp = (idx >> 5) % 28, kh = idx / (28 * 32);
            vw[u] = 0.f;
            if (idx < 9 * 28 * 32 && kp < KK) {
                const int kw = 8 - kp / a.Cout, co = kp % a.Cout;
                vw[u] = a.w[(((size_t)kh * 9 + kw) * a.Cin + n0 + ci) * a.Cout + co];
            }
        }
        c9_store_dy(sDy, vdy, tid);
#pragma unroll
        for (int u = 0; u < NWD; ++u) {
            const int idx = tid + 256 * u;
            if (idx < 9 * 28 * 32) sW[idx] = vw[u];
        }
    }
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll 1
    for (int kh = 0; kh < 9; ++kh) {
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const float bv = sW[(kh * 28 + 2 * s + lh) * 32 + li];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = 2 * wv + (t >> 1), half = t & 1;
                const float av = sDy[(r - kh + 8) * C9_DYW + a.Cout * (32 * half + li) + 2 * s + lh];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int gy = y0 + 2 * wv + (t >> 1), half = t & 1;
        if (gy >= a.H) continue;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int gx = x0 + 32 * half + (g & 3) + 8 * (g >> 2) + 4 * lh;
            if (gx >= a.W) continue;
            size_t o = (((size_t)b * a.H + gy) * a.W + gx) * a.Cin + n0 + li;
            float v = acc[t][g];
            if (UNMASK) {
                v *= dasr_act_grad_from_out(ld1((const TX*)a.mask_src + o), a.mask_act);
                if (a.unps_r > 1) {
                    const int ur = a.unps_r;
                    o = ((((size_t)b * (a.H / ur) + gy / ur) * (a.W / ur) + gx / ur) * a.Cin + n0 + li) * (ur * ur) +
                        (gy % ur) * ur + (gx % ur);
                }
            }
            if (a.accumulate) v += ld1((const TX*)a.out + o);
            st1((TX*)a.out + o, v);
        }
    }
}

// ------------------------------------------------------------------------------------------ wgrad
// Workgroup walks a strip of 8 x 64 pixel tiles; wave w owns tile rows 2w, 2w+1 and keeps nine
// [32 ci x 32 n'] accumulators (one per kh).  Each wave writes its own slab [9][32][32];
// k_conv9_wgrad_reduce sums slabs in a fixed order and un-folds n' = (8-kw)*3 + co.
template <typename TX>
__global__ void __launch_bounds__(256, 2) k_conv9x9_wgrad_mfma(Conv9Args a) {
    DASR_DYN_SMEM(smem);
    float* sX = (float*)smem;                         // [8][64][32]
    float* sDy = sX + C9_TH * C9_TQ * 32;             // [16][DYW]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int tiles_x = (a.W + C9_TQ - 1) / C9_TQ, tiles_y = (a.H + C9_TH - 1) / C9_TH;
    const int ci0 = blockIdx.x * 32;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int tile = blockIdx.y; tile < a.ntiles; tile += a.P) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int x0 = tx * C9_TQ, y0 = ty * C9_TH;
        {   // x tile in four batches of four float4 per thread (144 accumulator registers are live), dy in one
            float vdy[C9_NDY];
            float4 vx[4];
            auto ldx = [&](int u) {
                const int idx = tid + 256 * u;
                const int c4 = idx & 7, pix = idx >> 3;
                const int gy = y0 + pix / C9_TQ, gx = x0 + pix % C9_TQ;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gy < a.H && gx < a.W)
                    v = ld4((const TX*)a.x + (((size_t)b * a.H + gy) * a.W + gx) * a.Cin + ci0 + 4 * c4);
                return v;
            };
            auto stx = [&](int u, float4 v) {
                const int idx = tid + 256 * u;
                *(float4*)(sX + (idx >> 3) * 32 + 4 * (idx & 7)) = v;
            };
#pragma unroll
            for (int u = 0; u < 4; ++u) vx[u] = ldx(u);
            c9_load_dy(a, vdy, b, y0, x0, tid);
            __syncthreads();                         // every wave is done with the previous tile
#pragma unroll
            for (int u = 0; u < 4; ++u) stx(u, vx[u]);
            c9_store_dy(sDy, vdy, tid);
#pragma unroll
            for (int bt = 1; bt < 4; ++bt) {
#pragma unroll
                for (int u = 0; u < 4; ++u) vx[u] = ldx(4 * bt + u);
#pragma unroll
                for (int u = 0; u < 4; ++u) stx(4 * bt + u, vx[u]);
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int s = 0; s < 2 * C9_TQ / 2; ++s) {       // this wave's 2 rows x 64 columns, two pixels per step
            const int r = 2 * wv + s / (C9_TQ / 2), qx = 2 * (s % (C9_TQ / 2)) + lh;
            const float av = sX[(r * C9_TQ + qx) * 32 + li];
            const float* dyq = sDy + (r + 8) * C9_DYW + a.Cout * qx + li;
            float bvv[9];
#pragma unroll
            for (int kh = 0; kh < 9; ++kh) bvv[kh] = dyq[-kh * C9_DYW];
#pragma unroll
            for (int kh = 0; kh < 9; ++kh) acc[kh] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bvv[kh], acc[kh], 0, 0, 0);
        }
    }
    float* slab = (float*)a.out + ((size_t)(blockIdx.y * 4 + wv) * gridDim.x + blockIdx.x) * (9 * 32 * 32);
#pragma unroll
    for (int kh = 0; kh < 9; ++kh)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            int ci = (g & 3) + 8 * (g >> 2) + 4 * lh;
            slab[(kh * 32 + ci) * 32 + li] = acc[kh][g];
        }
}

// dw[kh][kw][ci][co] = sum over slabs of slab[cig][kh][ci%32][(8-kw)*Cout+co]; the slab range is split over
// blockIdx.y (partial sums meet in the zeroed dw through float atomics)
__global__ void __launch_bounds__(256) k_conv9_wgrad_reduce(const float* __restrict__ slabs, float* __restrict__ dw,
                                                            int Cin, int Cout, int nslabs, int cgroups, int per_y) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n = 81 * Cin * Cout;
    if (i >= n) return;
    int co = i % Cout, ci = (i / Cout) % Cin, kw = (i / (Cout * Cin)) % 9, kh = i / (Cout * Cin * 9);
    int np = (8 - kw) * Cout + co;
    const float* p = slabs + (size_t)(ci / 32) * (9 * 32 * 32) + (kh * 32 + (ci & 31)) * 32 + np;
    const int s0 = blockIdx.y * per_y;
    const int s1 = s0 + per_y < nslabs ? s0 + per_y : nslabs;
    float acc = 0.f;
    for (int s = s0; s < s1; ++s) acc += p[(size_t)s * cgroups * (9 * 32 * 32)];
    atomicAdd(&dw[i], acc);
}

// ------------------------------------------------------------------------------------------ host side
bool conv9_mfma_supported(const ConvGeom& g) {
    return g.KH == 9 && g.KW == 9 && g.stride == 1 && g.pad == 4 && !g.transposed && (g.Cin % 32) == 0 &&
           g.Cout >= 1 && g.Cout <= 3 && g.H == g.Ho && g.W == g.Wo;
}
template <typename TX>
static int conv9_fwd_impl(const ConvGeom& g, const TX* x, const float* w, const float* bias, float* y, void* stream) {
    Conv9Args a{x, w, bias, nullptr, y, g.B, g.H, g.W, g.Cin, g.Cout, 0, 0, 0, nullptr, 0, 1};
    int TWO = C9_TQ - 8;
    int tiles = ((g.W + TWO - 1) / TWO) * ((g.H + C9_TH - 1) / C9_TH);
    size_t lds = sizeof(float) * (size_t)((C9_TH + 8) * C9_TQ * C9_CKP + 9 * 32 * C9_CKP);
    DASR_LAUNCH((k_conv9x9_fwd_mfma<TX>), dim3(tiles, g.B), dim3(256), lds, stream, a);
    DASR_RETURN_LAUNCH_STATUS();
}
int conv9_mfma_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, void* stream) {
    return conv9_fwd_impl<float>(g, x, w, bias, y, stream);
}
template <typename TX>
static int conv9_dgrad_impl(const ConvGeom& g, const float* dconv, const float* w, TX* dx, int accumulate,
                            const TX* mask_src, int mask_act, int unps_r, void* stream) {
    Conv9Args a{nullptr, w, nullptr, dconv, dx, g.B, g.H, g.W, g.Cin, g.Cout, accumulate, 0, 0,
                mask_src, mask_act, unps_r < 1 ? 1 : unps_r};
    int tiles = ((g.W + C9_TQ - 1) / C9_TQ) * ((g.H + C9_TH - 1) / C9_TH);
    size_t lds = sizeof(float) * (size_t)((C9_TH + 8) * C9_DYW + 9 * 28 * 32);
    if (mask_src) {
        DASR_LAUNCH((k_conv9x9_dgrad_mfma<true, TX>), dim3(tiles, g.B, g.Cin / 32), dim3(256), lds, stream, a);
    } else {
        DASR_LAUNCH((k_conv9x9_dgrad_mfma<false, TX>), dim3(tiles, g.B, g.Cin / 32), dim3(256), lds, stream, a);
    }
    DASR_RETURN_LAUNCH_STATUS();
}
int conv9_mfma_dgrad(const ConvGeom& g, const float* dconv, const float* w, float* dx, int accumulate,
                     const float* mask_src, int mask_act, int unps_r, void* stream) {
    return conv9_dgrad_impl<float>(g, dconv, w, dx, accumulate, mask_src, mask_act, unps_r, stream);
}
static void conv9_wgrad_plan(const ConvGeom& g, int& ntiles, int& P) {
    ntiles = g.B * ((g.H + C9_TH - 1) / C9_TH) * ((g.W + C9_TQ - 1) / C9_TQ);
    P = 512 / (g.Cin / 32);
    if (P > ntiles) P = ntiles;
    if (P < 1) P = 1;
}
size_t conv9_mfma_wgrad_workspace(const ConvGeom& g) {
    int ntiles, P;
    conv9_wgrad_plan(g, ntiles, P);
    return sizeof(float) * (size_t)P * 4 * (g.Cin / 32) * 9 * 32 * 32;
}
template <typename TX>
static int conv9_wgrad_impl(const ConvGeom& g, const TX* x, const float* dconv, float* dw, void* workspace, void* stream) {
    int ntiles, P;
    conv9_wgrad_plan(g, ntiles, P);
    Conv9Args a{x, nullptr, nullptr, dconv, workspace, g.B, g.H, g.W, g.Cin, g.Cout, 0, P, ntiles, nullptr, 0, 1};
    size_t lds = sizeof(float) * (size_t)(C9_TH * C9_TQ * 32 + (C9_TH + 8) * C9_DYW);
    DASR_LAUNCH((k_conv9x9_wgrad_mfma<TX>), dim3(g.Cin / 32, P), dim3(256), lds, stream, a);
    int n = 81 * g.Cin * g.Cout;
    hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * n, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    const int nslabs = P * 4, ysplit = nslabs >= 64 ? 32 : 1;
    DASR_LAUNCH(k_conv9_wgrad_reduce, dim3(dasr_cdiv(n, 256), ysplit), dim3(256), 0, stream, (const float*)workspace, dw,
                g.Cin, g.Cout, nslabs, g.Cin / 32, (nslabs + ysplit - 1) / ysplit);
    DASR_RETURN_LAUNCH_STATUS();
}
int conv9_mfma_wgrad(const ConvGeom& g, const float* x, const float* dconv, float* dw, void* workspace, void* stream) {
    return conv9_wgrad_impl<float>(g, x, dconv, dw, workspace, stream);
}
