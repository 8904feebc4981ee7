// conv9_split.hip — the generator's 9x9 output convolution (conv_output, sftmd_arch.py:910,948: 32 -> 3 channels at HR
// resolution) of the FP32 path on the fp16 matrix cores, at fp32 accuracy: the "split fp16 x 2" scheme of conv_split_bf16.hip
// (two fp16 pieces per operand scaled by a power of two taken from the tensor's max |.|, three MFMA products per term, fp32
// accumulation) on the folding of conv9_mfma.hip / conv9_bf16_mfma.hip (one kernel axis goes into N: an MFMA tile is 27/32
// full).  The exact-fp32 MFMA kernels it replaces ran at 79 / 108 / 95 TF (forward / dgrad / wgrad: 11 ms of the 77 ms x8
// step); here the matrix work is 3/16 of theirs.
//   forward:  P[q][(kw,co)] = sum_{kh,ci} x[q.y+kh-4, q.x][ci] * w[kh][kw][ci][co]        M = pixels q, N = 27, K = 9*Cin
//             y[p][co]      = bias[co] + sum_kw P[(p.y, p.x+kw-4)][(kw,co)]                 (9-term shift-add via LDS, fp32)
//   dgrad:    dx[q][ci]     = sum_kh sum_k' E[q.y-kh+4][q.x][k'] * Wd[kh][k'][ci]            M = pixels, N = 32 ci, K = 9*32
//   wgrad:    dW[kh][k'][ci] = sum_q x[q][ci] * E[q.y-kh+4][q.x][k']                         M = ci, N = 27, K = pixels
// with k' = Cout*(8-kw) + co and E[row][px][k'] = the 27 consecutive values of the 3-channel dy row starting at pixel px-4.
// Everything that is an MFMA operand lives in LDS as TWO fp16 images (value and what the first rounding left), so tiles are
// half the size of the bf16 kernels': the forward walks 16-channel chunks of an 8 x 56 pixel tile, dgrad / wgrad 8 x 32 tiles.
// All three kernels are persistent (one 512-thread workgroup per CU, kernel images staged once) and hold the next tile's
// global loads in registers while the current one is multiplied.
#include "bf16.h"
#include "conv_kernels.h"

typedef _Float16 h16_t;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

#define S9_TH 8                       // tile rows

struct Conv9SplitArgs {
    const float* x;      // fwd / wgrad: [B,H,W,Cin]
    const float* w;      // HWIO [9][9][Cin][Cout]
    const float* bias;   // fwd
    const float* dy;     // dgrad / wgrad: [B,H,W,Cout]
    void* out;           // fwd: y; dgrad: dx; wgrad: slabs
    const float* xmax;   // amax buffers (dasr_common.h): x (fwd / wgrad), dy (dgrad / wgrad) ...
    const float* dmax;
    const float* wmax;   // ... and the kernel (fwd / dgrad)
    int B, H, W, Cin, Cout;
    int accumulate, P, ntiles;
};

// ---- scales: the maximum of an amax buffer's parts, by the whole workgroup (phase 1 before a barrier, phase 2 after it)
__device__ __forceinline__ float s9_amax_share(const float* buf) {
    float m = 0.f;
    if (buf) {
        int n;
        memcpy(&n, buf, 4);
        for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, buf[1 + i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}
__host__ __device__ static inline int s9_scale_exp(float m) {      // 2^k puts m in [2^14, 2^15); k clamped to [-60, 60]
    unsigned u;
    memcpy(&u, &m, 4);
    int k = 14 - ((int)((u >> 23) & 0xffu) - 127);
    return k < -60 ? -60 : (k > 60 ? 60 : k);
}
__host__ __device__ static inline float s9_pow2(int k) {
    const unsigned u = (unsigned)(127 + k) << 23;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// a, b: the two amax buffers of the launch; returns their scale exponents through s_red ([32] floats of LDS)
__device__ __forceinline__ void s9_scales(const float* a, const float* b, float* s_red, int& ka, int& kb) {
    const float ma = s9_amax_share(a), mb = s9_amax_share(b);
    if ((threadIdx.x & 63) == 0) {
        s_red[threadIdx.x >> 6] = ma;
        s_red[16 + (threadIdx.x >> 6)] = mb;
    }
    __syncthreads();
    float xa = s_red[0], xb = s_red[16];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { xa = fmaxf(xa, s_red[w]); xb = fmaxf(xb, s_red[16 + w]); }
    ka = s9_scale_exp(xa);
    kb = s9_scale_exp(xb);
}
// v s = h0 + h1 to 22-23 bits
__device__ __forceinline__ void s9_split(float v, float s, h16_t& h0, h16_t& h1) {
    v *= s;
    h0 = (h16_t)v;
    h1 = (h16_t)(v - (float)h0);
}
__device__ __forceinline__ f32x16 s9_mma(const h16x8& a0, const h16x8& a1, const h16x8& b0, const h16x8& b1, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c, 0, 0, 0);       // smallest terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
    return c;
}

// ------------------------------------------------------------------------------------------ forward
// 512 threads, tile = 8 rows x 56 columns of y (input tile 16 x 64 pixels), wave w owns tile row w (two 32-pixel M-tiles).
// K loop: 16-channel chunks; per chunk 9 kh x 2 M-tiles x 3 products = 54 MFMAs per wave.  LDS: two input images
// [16*64 px][24] fp16 (48-byte pixel stride: conflict-free ds_read_b128), the kernel [Cin/16][9 kh][32 n][24] x 2, the P
// image [8][64][28] fp32 aliasing the input.
#define S9F_TQ 64
#define S9F_ST 24
#define S9F_PST 28
__global__ void __launch_bounds__(512) k_conv9_fwd_split(Conv9SplitArgs a) {
    DASR_DYN_SMEM(smem);
    constexpr int NPX = (S9_TH + 8) * S9F_TQ;                      // 1024 staged pixels
    h16_t* sIn0 = (h16_t*)smem;                                    // [NPX][S9F_ST]
    h16_t* sIn1 = sIn0 + NPX * S9F_ST;
    const int NCK = a.Cin / 16;
    h16_t* sW0 = sIn1 + NPX * S9F_ST;                              // [NCK][9][32][S9F_ST]
    h16_t* sW1 = sW0 + NCK * 9 * 32 * S9F_ST;
    float* s_red = (float*)(sW1 + NCK * 9 * 32 * S9F_ST);          // [32]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int TWO = S9F_TQ - 8;
    const int tiles_x = (a.W + TWO - 1) / TWO, tiles_y = (a.H + S9_TH - 1) / S9_TH;
    const int total = tiles_x * tiles_y * a.B;
    const int NN = 9 * a.Cout;   // <= 27
    int kx, kw_;
    s9_scales(a.xmax, a.wmax, s_red, kx, kw_);
    const float sx = s9_pow2(kx), sw = s9_pow2(kw_), inv = s9_pow2(-(kx + kw_));
    // kernel images, once: sW[c][kh][n = kw*Cout+co][ci] (zero rows for n >= 9*Cout)
    for (int c = 0; c < NCK; ++c) {
        float wq[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int e = tid + 512 * u, ci = e & 15, n = (e >> 4) & 31, kh = e >> 9;
            const int nn = n < NN ? n : 0, kw = nn / a.Cout, co = nn % a.Cout;
            const float v = a.w[(((size_t)kh * 9 + kw) * a.Cin + 16 * c + ci) * a.Cout + co];
            wq[u] = n < NN ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int e = tid + 512 * u, ci = e & 15, n = (e >> 4) & 31, kh = e >> 9;
            h16_t h0, h1;
            s9_split(wq[u], sw, h0, h1);
            sW0[((c * 9 + kh) * 32 + n) * S9F_ST + ci] = h0;
            sW1[((c * 9 + kh) * 32 + n) * S9F_ST + ci] = h1;
        }
    }
    constexpr int NPC = NPX * 4 / 512;                             // 8 pieces of 16 bytes (4 channels) per thread and chunk
    u32x4_t vin[NPC];
    // loads of (tile, 16-channel chunk c0): unconditional (a tile index past the end re-reads the last tile)
    auto fetch = [&](int tile, int c0) {
        const int b = tile / (tiles_x * tiles_y), tt = tile - b * (tiles_x * tiles_y);
        const int x0 = (tt % tiles_x) * TWO, y0 = (tt / tiles_x) * S9_TH;
        const BufRsrc rx = dasr_make_rsrc(a.x + (size_t)b * a.H * a.W * a.Cin, (size_t)a.H * a.W * a.Cin * sizeof(float));
#pragma unroll
        for (int u = 0; u < NPC; ++u) {
            const int idx = tid + 512 * u, pix = idx >> 2, q4 = idx & 3;
            const int gy = y0 - 4 + pix / S9F_TQ, gx = x0 - 4 + pix % S9F_TQ;
            const unsigned off = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                                     ? (unsigned)(((gy * a.W + gx) * a.Cin + c0 + 4 * q4) * (int)sizeof(float)) : DASR_OOB;
            vin[u] = dasr_buffer_load16(rx, off);
        }
    };
    // Which tile a (workgroup, trip) slot works on.  a.P == 0: slot s = tile s.  a.P == 1 (256 workgroups, dealt to the eight
    // XCDs round robin): the 32 workgroups of an XCD take a 4 x 8 block of tiles per trip, so that the halo rows and columns a
    // tile shares with its neighbours (16 x 64 input pixels per 8 x 56 outputs: 2.3 x the bytes) are re-read from that XCD's L2
    // while they are hot instead of from another XCD's tile in flight at the same time.
    const int nbx = (tiles_x + 7) / 8, nby = (tiles_y + 3) / 4;
    const int nslots = a.P == 1 ? ((a.B * nbx * nby + 7) / 8) * 256 : total;
    auto slot_tile = [&](int s) {
        if (a.P != 1) return s;
        const int xcd = s & 7, j = (s >> 3) & 31, g = (s >> 8) * 8 + xcd;
        const int bimg = g / (nbx * nby), r = g - bimg * (nbx * nby);
        const int ty = (r / nbx) * 4 + (j >> 3), tx = (r % nbx) * 8 + (j & 7);
        return (bimg < a.B && ty < tiles_y && tx < tiles_x) ? (bimg * tiles_y + ty) * tiles_x + tx : -1;
    };
    auto next_slot = [&](int s) {                     // the next slot of this workgroup that holds a tile (or nslots)
        for (s += gridDim.x; s < nslots && slot_tile(s) < 0; s += gridDim.x) {}
        return s;
    };
    int slot = (int)blockIdx.x < nslots && slot_tile(blockIdx.x) >= 0 ? (int)blockIdx.x : next_slot(blockIdx.x);
    if (slot < nslots) fetch(slot_tile(slot), 0);
    for (; slot < nslots;) {
        const int tile = slot_tile(slot);
        const int nslot = next_slot(slot);
        const int b = tile / (tiles_x * tiles_y), tt = tile - b * (tiles_x * tiles_y);
        const int x0 = (tt % tiles_x) * TWO, y0 = (tt / tiles_x) * S9_TH;
        f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        for (int c = 0; c < NCK; ++c) {
            __syncthreads();                               // every wave is done with the previous chunk / tile (incl. its P image)
#pragma unroll
            for (int u = 0; u < NPC; ++u) {
                const int idx = tid + 512 * u;
                float f[4];
                memcpy(f, &vin[u], 16);
                h16x4 p0, p1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h16_t h0, h1;
                    s9_split(f[j], sx, h0, h1);
                    p0[j] = h0; p1[j] = h1;
                }
                *(h16x4*)(sIn0 + (idx >> 2) * S9F_ST + 4 * (idx & 3)) = p0;
                *(h16x4*)(sIn1 + (idx >> 2) * S9F_ST + 4 * (idx & 3)) = p1;
            }
            __syncthreads();
            {                                              // next chunk of this tile, or the first chunk of the next tile
                const bool lastc = c + 1 == NCK;
                const int nt = nslot < nslots ? slot_tile(nslot) : tile;
                fetch(lastc ? nt : tile, lastc ? 0 : 16 * (c + 1));
            }
            const int wo = c * 9 * 32 * S9F_ST;
#pragma unroll
            for (int kh = 0; kh < 9; ++kh) {
                const h16x8 B0 = *(const h16x8*)(sW0 + wo + (kh * 32 + li) * S9F_ST + 8 * lh);
                const h16x8 B1 = *(const h16x8*)(sW1 + wo + (kh * 32 + li) * S9F_ST + 8 * lh);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int o = ((wv + kh) * S9F_TQ + 32 * m + li) * S9F_ST + 8 * lh;
                    const h16x8 A0 = *(const h16x8*)(sIn0 + o), A1 = *(const h16x8*)(sIn1 + o);
                    acc[m] = s9_mma(A0, A1, B0, B1, acc[m]);
                }
            }
        }
        __syncthreads();
        float* sP = (float*)smem;   // [8][64][PST]
        if (li < S9F_PST) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int qx = 32 * m + (g & 3) + 8 * (g >> 2) + 4 * lh;
                    sP[(wv * S9F_TQ + qx) * S9F_PST + li] = acc[m][g];
                }
        }
        __syncthreads();
        const int nout = S9_TH * TWO * a.Cout;
        float* y = (float*)a.out;
        for (int idx = tid; idx < nout; idx += 512) {
            const int co = idx % a.Cout, ox = (idx / a.Cout) % TWO, r = idx / (a.Cout * TWO);
            const int gy = y0 + r, gx = x0 + ox;
            if (gy >= a.H || gx >= a.W) continue;
            float v = 0.f;
#pragma unroll
            for (int kw = 0; kw < 9; ++kw) v += sP[(r * S9F_TQ + ox + kw) * S9F_PST + kw * a.Cout + co];
            y[(((size_t)b * a.H + gy) * a.W + gx) * a.Cout + co] = fmaf(v, inv, a.bias ? a.bias[co] : 0.f);
        }
        slot = nslot;
    }
}

// ------------------------------------------------------------------------------------------ the E image (dgrad / wgrad)
// tile = 8 rows x 32 columns.  sDy: fp32 dy tile, rows y0-4 .. y0+11, columns x0-4 .. x0+35, Cout channels interleaved (zero
// outside the image).  E[row][px][k'] (two fp16 images, EST elements per pixel) = sd * sDy[row][Cout*px + k'], 0 for k' >= 9 Cout.
#define S9_TQ 32
#define S9_DYW ((S9_TQ + 8) * 3 + 8)      // fp32 dy tile row stride: 40 px * 3 ch + pad = 128
template <int EST>
__device__ __forceinline__ void s9_build_E(const Conv9SplitArgs& a, const float* sDy, h16_t* sE0, h16_t* sE1, float sd, int tid) {
    const int KK = 9 * a.Cout;
    for (int pc = tid; pc < (S9_TH + 8) * S9_TQ * 4; pc += 512) {         // 16-byte pieces: (row, px, 8 k')
        const int p8 = pc & 3, px = (pc >> 2) % S9_TQ, row = pc / (4 * S9_TQ);
        const float* src = sDy + row * S9_DYW + a.Cout * px + 8 * p8;
        h16x8 o0, o1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h16_t h0, h1;
            s9_split(8 * p8 + j < KK ? src[j] : 0.f, sd, h0, h1);
            o0[j] = h0; o1[j] = h1;
        }
        *(h16x8*)(sE0 + (row * S9_TQ + px) * EST + 8 * p8) = o0;
        *(h16x8*)(sE1 + (row * S9_TQ + px) * EST + 8 * p8) = o1;
    }
}
// the dy tile of a tile: loads into registers / registers into LDS (the next tile's loads stay in flight in between)
constexpr int S9_NDYE = (S9_TH + 8) * S9_DYW, S9_NDI = (S9_NDYE + 511) / 512;
__device__ __forceinline__ void s9_fetch_dy(const Conv9SplitArgs& a, int b, int y0, int x0, int tid, float (&vdy)[S9_NDI]) {
    const int rowf = (S9_TQ + 8) * a.Cout;
#pragma unroll
    for (int u = 0; u < S9_NDI; ++u) {
        const int idx = tid + 512 * u;
        const int f = idx % S9_DYW, ry = idx / S9_DYW;
        const int gy = y0 - 4 + ry, gx = x0 - 4 + f / a.Cout, co = f % a.Cout;
        const bool ok = idx < S9_NDYE && f < rowf && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        // (clamped address, unconditional load: a predicated prefetch would make hipcc drain vmcnt(0))
        const size_t o = ok ? (((size_t)b * a.H + gy) * a.W + gx) * a.Cout + co : 0;
        const float v = a.dy[o];
        vdy[u] = ok ? v : 0.f;
    }
}

// ------------------------------------------------------------------------------------------ dgrad
// tile: 8 rows x 32 columns of dx pixels x 32 input channels (blockIdx.y selects the 32-channel slice); wave w owns tile row w
// (one 32-pixel M-tile): 9 kh x 2 K-halves x 3 products = 54 MFMAs per wave and tile.
__global__ void __launch_bounds__(512) k_conv9_dgrad_split(Conv9SplitArgs a) {
    DASR_DYN_SMEM(smem);
    constexpr int EST = 40;                                        // 80-byte pixel stride (conflict-free ds_read_b128)
    constexpr int NE = (S9_TH + 8) * S9_TQ * EST;
    h16_t* sE0 = (h16_t*)smem;                                     // [16][32][EST]
    h16_t* sE1 = sE0 + NE;
    h16_t* sW0 = sE1 + NE;                                         // [9 kh][32 ci][EST]: Wd[kh][k'][ci], k' contiguous
    h16_t* sW1 = sW0 + 9 * 32 * EST;
    float* sDy = (float*)(sW1 + 9 * 32 * EST);                     // [16][DYW]
    float* s_red = sDy + S9_NDYE;                                  // [32]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int tiles_x = (a.W + S9_TQ - 1) / S9_TQ, tiles_y = (a.H + S9_TH - 1) / S9_TH;
    const int total = tiles_x * tiles_y * a.B;
    const int n0 = blockIdx.y * 32;
    const int KK = 9 * a.Cout;
    int kd, kw_;
    s9_scales(a.dmax, a.wmax, s_red, kd, kw_);
    const float sd = s9_pow2(kd), sw = s9_pow2(kw_), inv = s9_pow2(-(kd + kw_));
    {                                                              // Wd[kh][k'][ci] = w[kh][8 - k'/Cout][ci][k' % Cout], once
        float wq[18];
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            const int e = tid + 512 * u, kp = e & 31, ci = (e >> 5) & 31, kh = e >> 10;
            const int kq = kp < KK ? kp : 0, kw = 8 - kq / a.Cout, co = kq % a.Cout;
            const float v = a.w[(((size_t)kh * 9 + kw) * a.Cin + n0 + ci) * a.Cout + co];
            wq[u] = kp < KK ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            const int e = tid + 512 * u, kp = e & 31, ci = (e >> 5) & 31, kh = e >> 10;
            h16_t h0, h1;
            s9_split(wq[u], sw, h0, h1);
            sW0[(kh * 32 + ci) * EST + kp] = h0;
            sW1[(kh * 32 + ci) * EST + kp] = h1;
        }
    }
    float vdy[S9_NDI];
    auto origin = [&](int tile, int& b, int& x0, int& y0) {
        b = tile / (tiles_x * tiles_y);
        const int tt = tile - b * (tiles_x * tiles_y);
        x0 = (tt % tiles_x) * S9_TQ;
        y0 = (tt / tiles_x) * S9_TH;
    };
    int tile = blockIdx.x;
    if (tile < total) {
        int b, x0, y0;
        origin(tile, b, x0, y0);
        s9_fetch_dy(a, b, y0, x0, tid, vdy);
    }
    for (; tile < total; tile += gridDim.x) {
        int b, x0, y0;
        origin(tile, b, x0, y0);
        __syncthreads();                                           // every wave is done with the previous tile's E image
#pragma unroll
        for (int u = 0; u < S9_NDI; ++u) {
            const int idx = tid + 512 * u;
            if (idx < S9_NDYE) sDy[idx] = vdy[u];
        }
        __syncthreads();
        {
            int nb, nx0, ny0;
            origin(tile + (int)gridDim.x < total ? tile + (int)gridDim.x : tile, nb, nx0, ny0);
            s9_fetch_dy(a, nb, ny0, nx0, tid, vdy);
        }
        s9_build_E<EST>(a, sDy, sE0, sE1, sd, tid);
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 9; ++kh)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int wo = (kh * 32 + li) * EST + 16 * q + 8 * lh;
                const h16x8 B0 = *(const h16x8*)(sW0 + wo), B1 = *(const h16x8*)(sW1 + wo);
                // dy row of output row wv for this kh: wv - kh + 4 (+4 for the tile's first row y0-4) = wv - kh + 8
                const int eo = ((wv - kh + 8) * S9_TQ + li) * EST + 16 * q + 8 * lh;
                const h16x8 A0 = *(const h16x8*)(sE0 + eo), A1 = *(const h16x8*)(sE1 + eo);
                acc = s9_mma(A0, A1, B0, B1, acc);
            }
        float* dx = (float*)a.out;
        const int gy = y0 + wv;
        if (gy < a.H) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int gx = x0 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                if (gx >= a.W) continue;
                const size_t o = (((size_t)b * a.H + gy) * a.W + gx) * a.Cin + n0 + li;
                float v = acc[g] * inv;
                if (a.accumulate) v += dx[o];
                dx[o] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ wgrad
// blockIdx.x = 32-channel slice of Cin, blockIdx.y = partial index p (tiles p, p+P, ...).  512 threads: wave w owns
// tile row w, all nine kh (9 accumulators = dW[kh] as a 32 ci x 32 k' tile).  K = pixels, so both operands are read
// TRANSPOSED from pixel-major LDS tiles with ds_read_b64_tr_b16 (on 16-bit fp16 elements): A = x^T ([ci][pixel]),
// B = E^T.  Each wave writes its own slab [9][32][32] (times the inverse of the two scales); k_conv9_wgrad_reduce_split
// sums the slabs and un-folds k'.
__device__ __forceinline__ h16x4 s9_read_tr16(const h16_t* p) {
    const bf16x4 r = lds_read_tr16((const bf16_t*)p);
    h16x4 v;
    __builtin_memcpy(&v, &r, 8);
    return v;
}
__global__ void __launch_bounds__(512) k_conv9_wgrad_split(Conv9SplitArgs a) {
    DASR_DYN_SMEM(smem);
    constexpr int EST = 32;                                        // 64-byte pixel stride: conflict-free transposed reads
    constexpr int NX = S9_TH * S9_TQ * 32, NE = (S9_TH + 8) * S9_TQ * EST;
    h16_t* sX0 = (h16_t*)smem;                                     // [8][32][32]
    h16_t* sX1 = sX0 + NX;
    h16_t* sE0 = sX1 + NX;                                         // [16][32][EST]
    h16_t* sE1 = sE0 + NE;
    float* sDy = (float*)(sE1 + NE);                               // [16][DYW]
    float* s_red = sDy + S9_NDYE;                                  // [32]
    const int tid = threadIdx.x, lane = tid & 63, wv = DASR_UNIFORM((int)(tid >> 6)), li = lane & 31, lh = lane >> 5;
    const int tiles_x = (a.W + S9_TQ - 1) / S9_TQ, tiles_y = (a.H + S9_TH - 1) / S9_TH;
    const int ci0 = blockIdx.x * 32;
    int kx, kd;
    s9_scales(a.xmax, a.dmax, s_red, kx, kd);
    const float sx = s9_pow2(kx), sd = s9_pow2(kd), inv = s9_pow2(-(kx + kd));
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    constexpr int NXP = S9_TH * S9_TQ * 8 / 512;                   // 4 pieces of 16 bytes (4 channels) per thread
    u32x4_t vx[NXP];
    float vdy[S9_NDI];
    auto origin = [&](int tile, int& b, int& x0, int& y0) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y;
        b = tile / (tiles_x * tiles_y);
        x0 = tx * S9_TQ;
        y0 = ty * S9_TH;
    };
    auto fetch = [&](int tile) {
        int b, x0, y0;
        origin(tile, b, x0, y0);
        const BufRsrc rx = dasr_make_rsrc(a.x + (size_t)b * a.H * a.W * a.Cin, (size_t)a.H * a.W * a.Cin * sizeof(float));
#pragma unroll
        for (int u = 0; u < NXP; ++u) {
            const int idx = tid + 512 * u, pix = idx >> 3, q8 = idx & 7;
            const int gy = y0 + pix / S9_TQ, gx = x0 + pix % S9_TQ;
            vx[u] = dasr_buffer_load16(rx, (gy < a.H && gx < a.W)
                                               ? (unsigned)(((gy * a.W + gx) * a.Cin + ci0 + 4 * q8) * (int)sizeof(float))
                                               : DASR_OOB);
        }
        s9_fetch_dy(a, b, y0, x0, tid, vdy);
    };
    int tile = blockIdx.y;
    if (tile < a.ntiles) fetch(tile);
    for (; tile < a.ntiles; tile += a.P) {
        __syncthreads();                         // every wave is done with the previous tile
#pragma unroll
        for (int u = 0; u < S9_NDI; ++u) {
            const int idx = tid + 512 * u;
            if (idx < S9_NDYE) sDy[idx] = vdy[u];
        }
#pragma unroll
        for (int u = 0; u < NXP; ++u) {
            const int idx = tid + 512 * u;
            float f[4];
            memcpy(f, &vx[u], 16);
            h16x4 p0, p1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h16_t h0, h1;
                s9_split(f[j], sx, h0, h1);
                p0[j] = h0; p1[j] = h1;
            }
            *(h16x4*)(sX0 + (idx >> 3) * 32 + 4 * (idx & 7)) = p0;
            *(h16x4*)(sX1 + (idx >> 3) * 32 + 4 * (idx & 7)) = p1;
        }
        __syncthreads();
        fetch(tile + a.P < a.ntiles ? tile + a.P : tile);
        s9_build_E<EST>(a, sDy, sE0, sE1, sd, tid);
        __syncthreads();
        // K-step = 16 consecutive pixels of this wave's row
#pragma unroll
        for (int s = 0; s < S9_TQ / 16; ++s) {
            const int px = 16 * s + 8 * lh + tq;
            const int xo = (wv * S9_TQ + px) * 32 + 16 * tg + 4 * tp;
            h16x8 a0, a1;
            {
                const h16x4 l0 = s9_read_tr16(sX0 + xo), u0 = s9_read_tr16(sX0 + xo + 4 * 32);
                const h16x4 l1 = s9_read_tr16(sX1 + xo), u1 = s9_read_tr16(sX1 + xo + 4 * 32);
#pragma unroll
                for (int e = 0; e < 4; ++e) { a0[e] = l0[e]; a0[4 + e] = u0[e]; a1[e] = l1[e]; a1[4 + e] = u1[e]; }
            }
#pragma unroll
            for (int kh = 0; kh < 9; ++kh) {
                const int eo = ((wv - kh + 8) * S9_TQ + px) * EST + 16 * tg + 4 * tp;
                const h16x4 l0 = s9_read_tr16(sE0 + eo), u0 = s9_read_tr16(sE0 + eo + 4 * EST);
                const h16x4 l1 = s9_read_tr16(sE1 + eo), u1 = s9_read_tr16(sE1 + eo + 4 * EST);
                h16x8 b0, b1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { b0[e] = l0[e]; b0[4 + e] = u0[e]; b1[e] = l1[e]; b1[4 + e] = u1[e]; }
                acc[kh] = s9_mma(a0, a1, b0, b1, acc[kh]);
            }
        }
    }
    float* slab = (float*)a.out + ((size_t)(blockIdx.y * 8 + wv) * gridDim.x + blockIdx.x) * (9 * 32 * 32);
#pragma unroll
    for (int kh = 0; kh < 9; ++kh)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int ci = (g & 3) + 8 * (g >> 2) + 4 * lh;
            slab[(kh * 32 + ci) * 32 + li] = acc[kh][g] * inv;
        }
}

// dw[kh][kw][ci][co] = sum over slabs of slab[cig][kh][ci%32][(8-kw)*Cout+co]; the slab range is split over
// blockIdx.y (partial sums meet in the zeroed dw through float atomics)
__global__ void __launch_bounds__(256) k_conv9_wgrad_reduce_split(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                  int Cin, int Cout, int nslabs, int cgroups, int per_y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = 81 * Cin * Cout;
    if (i >= n) return;
    const int co = i % Cout, ci = (i / Cout) % Cin, kw = (i / (Cout * Cin)) % 9, kh = i / (Cout * Cin * 9);
    const int np = (8 - kw) * Cout + co;
    const float* p = slabs + (size_t)(ci / 32) * (9 * 32 * 32) + (kh * 32 + (ci & 31)) * 32 + np;
    const int s0 = blockIdx.y * per_y;
    const int s1 = s0 + per_y < nslabs ? s0 + per_y : nslabs;
    float acc = 0.f;
    for (int s = s0; s < s1; ++s) acc += p[(size_t)s * cgroups * (9 * 32 * 32)];
    atomicAdd(&dw[i], acc);
}

// ------------------------------------------------------------------------------------------ host side
bool conv9_split_supported(const ConvGeom& g) {
    return conv9_mfma_supported(g) && (g.Cin / 16) * 9 * 32 * S9F_ST * 2 * 2 + (S9_TH + 8) * S9F_TQ * S9F_ST * 2 * 2 + 128 <= 160 * 1024 &&
           (size_t)g.H * g.W * g.Cin * sizeof(float) < ((size_t)1 << 31);
}
int conv9_split_fwd(const ConvGeom& g, const float* x, const float* xmax, const float* w, const float* wmax, const float* bias,
                    float* y, void* stream) {
    Conv9SplitArgs a{x, w, bias, nullptr, y, xmax, nullptr, wmax, g.B, g.H, g.W, g.Cin, g.Cout, 0, 0, 0};
    const int TWO = S9F_TQ - 8;
    const int tiles = ((g.W + TWO - 1) / TWO) * ((g.H + S9_TH - 1) / S9_TH) * g.B;
    const size_t lds = sizeof(h16_t) * (size_t)(2 * (S9_TH + 8) * S9F_TQ * S9F_ST + 2 * (g.Cin / 16) * 9 * 32 * S9F_ST) + 128;
    const int cap = (dasr_get_conv_bf16_impl() & 3) == 2 ? 3 : 256;    // (impl 2, tests: long per-workgroup tile lists)
    a.P = (tiles >= 256 && cap == 256 && (dasr_get_conv_bf16_impl() & 4096) == 0) ? 1 : 0;      // XCD-blocked tile order (+ 4096: linear, A/B)
    DASR_LAUNCH(k_conv9_fwd_split, dim3(tiles < cap ? tiles : cap), dim3(512), lds, stream, a);     // one persistent workgroup per CU
    DASR_RETURN_LAUNCH_STATUS();
}
int conv9_split_dgrad(const ConvGeom& g, const float* dconv, const float* dmax, const float* w, const float* wmax, float* dx,
                      int accumulate, void* stream) {
    Conv9SplitArgs a{nullptr, w, nullptr, dconv, dx, nullptr, dmax, wmax, g.B, g.H, g.W, g.Cin, g.Cout, accumulate, 0, 0};
    const int tiles = ((g.W + S9_TQ - 1) / S9_TQ) * ((g.H + S9_TH - 1) / S9_TH) * g.B;
    const size_t lds = sizeof(h16_t) * (size_t)(2 * (S9_TH + 8) * S9_TQ * 40 + 2 * 9 * 32 * 40) + sizeof(float) * (size_t)(S9_NDYE + 32);
    int per = 256 / (g.Cin / 32) > 0 ? 256 / (g.Cin / 32) : 1;               // persistent: one workgroup per CU over all slices
    if ((dasr_get_conv_bf16_impl() & 3) == 2) per = 3;
    DASR_LAUNCH(k_conv9_dgrad_split, dim3(tiles < per ? tiles : per, g.Cin / 32), dim3(512), lds, stream, a);
    DASR_RETURN_LAUNCH_STATUS();
}
static void conv9_split_wgrad_plan(const ConvGeom& g, int& ntiles, int& P) {
    ntiles = g.B * ((g.H + S9_TH - 1) / S9_TH) * ((g.W + S9_TQ - 1) / S9_TQ);
    P = 256 / (g.Cin / 32);
    if ((dasr_get_conv_bf16_impl() & 3) == 2) P = 3;
    if (P > ntiles) P = ntiles;
    if (P < 1) P = 1;
}
size_t conv9_split_wgrad_workspace(const ConvGeom& g) {
    const int ntiles = g.B * ((g.H + S9_TH - 1) / S9_TH) * ((g.W + S9_TQ - 1) / S9_TQ);
    int P = 256 / (g.Cin / 32);
    if (P > ntiles) P = ntiles;
    if (P < 1) P = 1;
    return sizeof(float) * (size_t)P * 8 * (g.Cin / 32) * 9 * 32 * 32;
}
int conv9_split_wgrad(const ConvGeom& g, const float* x, const float* xmax, const float* dconv, const float* dmax, float* dw,
                      void* workspace, void* stream) {
    int ntiles, P;
    conv9_split_wgrad_plan(g, ntiles, P);
    Conv9SplitArgs a{x, nullptr, nullptr, dconv, workspace, xmax, dmax, nullptr, g.B, g.H, g.W, g.Cin, g.Cout, 0, P, ntiles};
    const size_t lds = sizeof(h16_t) * (size_t)(2 * S9_TH * S9_TQ * 32 + 2 * (S9_TH + 8) * S9_TQ * 32) +
                       sizeof(float) * (size_t)(S9_NDYE + 32);
    DASR_LAUNCH(k_conv9_wgrad_split, dim3(g.Cin / 32, P), dim3(512), lds, stream, a);
    const int n = 81 * g.Cin * g.Cout;
    hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * n, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    const int nslabs = P * 8, ysplit = nslabs >= 64 ? 32 : 1;
    DASR_LAUNCH(k_conv9_wgrad_reduce_split, dim3(dasr_cdiv(n, 256), ysplit), dim3(256), 0, stream, (const float*)workspace, dw,
                g.Cin, g.Cout, nslabs, g.Cin / 32, (nslabs + ysplit - 1) / ysplit);
    DASR_RETURN_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int dasr_conv9_split_supported(int H, int W, int Cin, int Cout) {
    ConvGeom g{1, H, W, Cin, H, W, Cout, 9, 9, 1, 4, 0};
    return (H > 0 && W > 0 && Cin > 0 && Cout > 0 && conv9_split_supported(g)) ? 1 : 0;
}
extern "C" int dasr_conv9_fwd_split2(const float* x, const float* xmax, const float* w_hwio, const float* wmax, const float* bias,
                                     float* y, int B, int H, int W, int Cin, int Cout, void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(xmax); DASR_CHECK_PTR(w_hwio); DASR_CHECK_PTR(wmax); DASR_CHECK_PTR(y);
    DASR_CHECK_SHAPE(B > 0);
    if (!dasr_conv9_split_supported(H, W, Cin, Cout)) return DASR_E_UNSUPPORTED;
    ConvGeom g{B, H, W, Cin, H, W, Cout, 9, 9, 1, 4, 0};
    return conv9_split_fwd(g, x, xmax, w_hwio, wmax, bias, y, stream);
}
extern "C" int dasr_conv9_dgrad_split2(const float* dconv, const float* dmax, const float* w_hwio, const float* wmax, float* dx,
                                       int accumulate, int B, int H, int W, int Cin, int Cout, void* stream) {
    DASR_CHECK_PTR(dconv); DASR_CHECK_PTR(dmax); DASR_CHECK_PTR(w_hwio); DASR_CHECK_PTR(wmax); DASR_CHECK_PTR(dx);
    DASR_CHECK_SHAPE(B > 0);
    if (!dasr_conv9_split_supported(H, W, Cin, Cout)) return DASR_E_UNSUPPORTED;
    ConvGeom g{B, H, W, Cin, H, W, Cout, 9, 9, 1, 4, 0};
    return conv9_split_dgrad(g, dconv, dmax, w_hwio, wmax, dx, accumulate, stream);
}
extern "C" size_t dasr_conv9_wgrad_split2_workspace(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || !dasr_conv9_split_supported(H, W, Cin, Cout)) return 0;
    ConvGeom g{B, H, W, Cin, H, W, Cout, 9, 9, 1, 4, 0};
    return conv9_split_wgrad_workspace(g);
}
extern "C" int dasr_conv9_wgrad_split2(const float* x, const float* xmax, const float* dconv, const float* dmax, float* dw_hwio,
                                       float* dbias, void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin,
                                       int Cout, void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(xmax); DASR_CHECK_PTR(dconv); DASR_CHECK_PTR(dmax); DASR_CHECK_PTR(dw_hwio);
    DASR_CHECK_PTR(workspace);
    DASR_CHECK_SHAPE(B > 0);
    if (!dasr_conv9_split_supported(H, W, Cin, Cout)) return DASR_E_UNSUPPORTED;
    if (workspace_bytes < dasr_conv9_wgrad_split2_workspace(B, H, W, Cin, Cout)) return DASR_E_WORKSPACE;
    ConvGeom g{B, H, W, Cin, H, W, Cout, 9, 9, 1, 4, 0};
    int rc = conv9_split_wgrad(g, x, xmax, dconv, dmax, dw_hwio, workspace, stream);
    if (rc) return rc;
    if (dbias) rc = conv_colsum(dconv, dbias, (size_t)B * H * W, Cout, stream);
    return rc;
}
