// ssim.hip — pytorch_ssim.ssim (reference codes/pytorch_ssim/__init__.py:17-37, used by the validation loop train.py:239):
// the five 11 x 11 Gaussian-windowed moments of two [B,C,H,W] images (zero padding) and the SSIM map's mean, in one pass.
// The reference spends five grouped F.conv2d calls and half a dozen elementwise passes on it; here a workgroup takes a
// 32 x 32 tile of one plane, stages the 42 x 42 halo of both images in LDS, filters the five products horizontally into LDS
// and vertically into registers (the window is separable: g (x) g), evaluates the SSIM ratio per pixel and writes ONE partial
// sum; a second tiny kernel adds the partials of a sample in a fixed order (deterministic, no atomics).
#include "dasr_common.h"

#define SS_T 32
#define SS_R 5
#define SS_HALO (SS_T + 2 * SS_R)      // 42

struct SsimArgs {
    const float* a;
    const float* b;
    float* partial;      // [B*C][tiles]
    int H, W, tiles_x, tiles_y;
    float g[11];         // the reference's 1-D window: exp(-(i-5)^2 / (2 * 1.5^2)), normalised in fp32
};

__global__ void __launch_bounds__(256) k_ssim_tiles(SsimArgs s) {
    __shared__ float sA[SS_HALO][SS_HALO + 1], sB[SS_HALO][SS_HALO + 1];
    __shared__ float sHz[5][SS_HALO][SS_T + 1];
    __shared__ float red[4];
    const int tid = threadIdx.x, plane = blockIdx.y;
    const int tile = blockIdx.x, tx = tile % s.tiles_x, ty = tile / s.tiles_x;
    const int x0 = tx * SS_T, y0 = ty * SS_T;
    const float* pa = s.a + (size_t)plane * s.H * s.W;
    const float* pb = s.b + (size_t)plane * s.H * s.W;
    for (int i = tid; i < SS_HALO * SS_HALO; i += 256) {
        const int r = i / SS_HALO, c = i - r * SS_HALO;
        const int gy = y0 + r - SS_R, gx = x0 + c - SS_R;
        const bool in = gy >= 0 && gy < s.H && gx >= 0 && gx < s.W;
        sA[r][c] = in ? pa[(size_t)gy * s.W + gx] : 0.f;
        sB[r][c] = in ? pb[(size_t)gy * s.W + gx] : 0.f;
    }
    __syncthreads();
    // horizontal pass: 42 rows x 32 columns x 5 moments
    for (int i = tid; i < SS_HALO * SS_T; i += 256) {
        const int r = i / SS_T, c = i - r * SS_T;
        float m1 = 0.f, m2 = 0.f, q11 = 0.f, q22 = 0.f, q12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float av = sA[r][c + k], bv = sB[r][c + k], w = s.g[k];
            m1 += w * av;
            m2 += w * bv;
            q11 += w * (av * av);
            q22 += w * (bv * bv);
            q12 += w * (av * bv);
        }
        sHz[0][r][c] = m1; sHz[1][r][c] = m2; sHz[2][r][c] = q11; sHz[3][r][c] = q22; sHz[4][r][c] = q12;
    }
    __syncthreads();
    // vertical pass + SSIM ratio: 4 pixels per thread
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = tid + 256 * u, r = i / SS_T, c = i - r * SS_T;
        if (y0 + r >= s.H || x0 + c >= s.W) continue;
        float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = s.g[k];
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] += w * sHz[q][r + k][c];
        }
        const float mu1_sq = m[0] * m[0], mu2_sq = m[1] * m[1], mu12 = m[0] * m[1];
        const float s1 = m[2] - mu1_sq, s2 = m[3] - mu2_sq, s12 = m[4] - mu12;
        acc += ((2.f * mu12 + C1) * (2.f * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) s.partial[(size_t)plane * gridDim.x + tile] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[b] = sum of the partials of sample b / (C H W), in index order
__global__ void __launch_bounds__(256) k_ssim_finish(const float* __restrict__ partial, float* __restrict__ out, int n_per_sample,
                                                     float inv_count) {
    __shared__ float red[256];
    const float* p = partial + (size_t)blockIdx.x * n_per_sample;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_per_sample; i += 256) acc += p[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] * inv_count;
}

extern "C" size_t dasr_ssim_workspace(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return sizeof(float) * (size_t)B * C * ((H + SS_T - 1) / SS_T) * ((W + SS_T - 1) / SS_T);
}
extern "C" int dasr_ssim(const float* img1, const float* img2, const float* window11, float* out_per_sample, void* workspace,
                         size_t workspace_bytes, int B, int C, int H, int W, void* stream) {
    DASR_CHECK_PTR(img1); DASR_CHECK_PTR(img2); DASR_CHECK_PTR(window11); DASR_CHECK_PTR(out_per_sample); DASR_CHECK_PTR(workspace);
    DASR_CHECK_SHAPE(B > 0 && C > 0 && H > 0 && W > 0 && (size_t)B * C <= 65535);
    if (workspace_bytes < dasr_ssim_workspace(B, C, H, W)) return DASR_E_WORKSPACE;
    SsimArgs s;
    s.a = img1; s.b = img2; s.partial = (float*)workspace;
    s.H = H; s.W = W;
    s.tiles_x = (W + SS_T - 1) / SS_T; s.tiles_y = (H + SS_T - 1) / SS_T;
    for (int i = 0; i < 11; ++i) s.g[i] = window11[i];          // (host pointer: eleven floats, passed by value)
    const int tiles = s.tiles_x * s.tiles_y;
    DASR_LAUNCH(k_ssim_tiles, dim3(tiles, B * C), dim3(256), 0, stream, s);
    DASR_LAUNCH(k_ssim_finish, dim3(B), dim3(256), 0, stream, (const float*)workspace, out_per_sample, C * tiles,
                1.0f / ((float)C * (float)H * (float)W));
    DASR_RETURN_LAUNCH_STATUS();
}
