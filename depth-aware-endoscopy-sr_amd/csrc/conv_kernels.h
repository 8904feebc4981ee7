// conv_kernels.h — geometry struct and internal entry points shared by the convolution sources.
#pragma once
#include "dasr_common.h"
#include "bf16.h"

struct ConvGeom {
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed;
};

// Output element index with the fused PixelShuffle(ps_r) store (nn.PixelShuffle index map:
// out[b, c, oy*r+i, ox*r+j] = in[b, c*r*r + i*r + j, oy, ox], here in NHWC).
__host__ __device__ __forceinline__ size_t conv_out_index(const ConvGeom& g, int b, int oy, int ox, int co, int ps_r) {
    if (ps_r <= 1) return (((size_t)b * g.Ho + oy) * g.Wo + ox) * g.Cout + co;
    int rr = ps_r * ps_r;
    int c = co / rr, i = (co / ps_r) % ps_r, j = co % ps_r;
    return (((size_t)b * g.Ho * ps_r + (size_t)oy * ps_r + i) * ((size_t)g.Wo * ps_r) + (size_t)ox * ps_r + j) *
               (g.Cout / rr) + c;
}

// conv_direct.hip
int conv_direct_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, const float* residual,
                    float* y, int act, int ps_r, void* stream);
int conv_epilogue_bwd(const ConvGeom& g, const float* dy, const float* y, float* dconv, int act, int ps_r,
                      void* stream, float* amax = nullptr);
int conv_direct_dgrad(const ConvGeom& g, const float* dconv, const float* w, float* dx, int accumulate, void* stream);
int conv_direct_wgrad(const ConvGeom& g, const float* x, const float* dconv, float* dw, void* stream);
int conv_colsum(const float* m, float* out, size_t rows, int C, void* stream);

// conv_mfma.hip (3x3, stride 1, pad 1, channel counts multiples of 16/32)
bool conv_mfma_supported(const ConvGeom& g);
bool conv_mfma_fwd_stats_supported(const ConvGeom& g);
size_t conv_mfma_fwd_stats_workspace(const ConvGeom& g);
int conv_mfma_fwd_stats(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, float* mean,
                        float* var, void* workspace, void* stream);
bool conv_mfma_dgrad_supported(const ConvGeom& g);
bool conv_mfma_wgrad_supported(const ConvGeom& g);
int conv_mfma_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, const float* residual,
                  float* y, int act, int ps_r, void* stream);
int conv_mfma_dgrad(const ConvGeom& g, const float* dconv, const float* w, float* dx, int accumulate,
                    const float* mask_src, int mask_act, int unps_r, void* stream);
size_t conv_mfma_wgrad_workspace(const ConvGeom& g);
int conv_mfma_wgrad(const ConvGeom& g, const float* x, const float* dconv, float* dw, float* dbias, void* workspace,
                    void* stream);   // dbias (may be null) is produced by the kernel itself

// conv9_mfma.hip (9x9, pad 4, Cin % 32 == 0, Cout <= 3: the output convolution)
bool conv9_mfma_supported(const ConvGeom& g);
int conv9_mfma_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, void* stream);
int conv9_mfma_dgrad(const ConvGeom& g, const float* dconv, const float* w, float* dx, int accumulate,
                     const float* mask_src, int mask_act, int unps_r, void* stream);
size_t conv9_mfma_wgrad_workspace(const ConvGeom& g);
int conv9_mfma_wgrad(const ConvGeom& g, const float* x, const float* dconv, float* dw, void* workspace, void* stream);

// conv_gather_mfma.hip (any stride / transposed; Cin % 8 == 0, Cout % 32 == 0: the encoder layers)
bool conv_gather_fwd_supported(const ConvGeom& g);
bool conv_gather_dgrad_supported(const ConvGeom& g);
bool conv_gather_wgrad_supported(const ConvGeom& g);
int conv_gather_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, int act,
                    void* stream);
int conv_gather_dgrad(const ConvGeom& g, const float* dconv, const float* w, float* dx, int accumulate, void* stream);
size_t conv_gather_wgrad_workspace(const ConvGeom& g);
int conv_gather_wgrad(const ConvGeom& g, const float* x, const float* dconv, float* dw, void* workspace, void* stream);
int wgrad_reduce_launch(const float* slabs, float* dw, size_t n, int P, void* stream, const float* bslabs = nullptr,
                        float* dbias = nullptr, int Cout = 0);

// conv_c1.hip (3x3, Cin == 1: SEAN.mlp_mask on the depth map)
bool conv_c1_supported(const ConvGeom& g);
int conv_c1_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, int act, void* stream,
                float* amax = nullptr);
// (conv_split_bf16.hip) raise *amax to max |x| over n floats without clearing it first - the follow-up pass of entry points
// whose kernel does not track the maximum of what it stores itself
int absmax_raise(const float* x, size_t n, float* amax, void* stream);
int conv_c1_wgrad(const ConvGeom& g, const float* x, const float* dy, const float* yact, int act, float* dw, float* dbias,
                  void* stream);

// ---- bf16 activations (mixed-precision path) ---------------------------------------------------------------------
// conv_bf16_mfma.hip (3x3, stride 1, pad 1, Cin % 32 == 0, Cout % 32 == 0; bf16 x, packed bf16 w, fp32 bias / dw / dbias)
bool conv_bf16_supported(const ConvGeom& g);
bool conv_bf16_dgrad_supported(const ConvGeom& g);
bool conv_bf16_wgrad_supported(const ConvGeom& g);
int conv_bf16_fwd(const ConvGeom& g, const bf16_t* x, const bf16_t* w, const float* bias, const bf16_t* residual,
                  bf16_t* y, int act, int ps_r, void* stream);
int conv_bf16_dgrad(const ConvGeom& g, const bf16_t* dconv, const bf16_t* w, bf16_t* dx, int accumulate, void* stream);
size_t conv_bf16_wgrad_workspace(const ConvGeom& g);
int conv_bf16_wgrad(const ConvGeom& g, const bf16_t* x, const bf16_t* dconv, float* dw, float* dbias, void* workspace,
                    void* stream);
// conv_bf16_v2.hip (persistent LDS-DMA kernel for forward / dgrad; Cin % 32 == 0, Cout % 64 == 0 in the KERNEL's terms -
// reduction / produced channels -, PixelShuffle 1 or 2); conv_bf16_fwd / conv_bf16_dgrad dispatch to it
bool conv_bf16_v2_supported(int H, int W, int Cin, int Cout, int ps_r, bool has_residual, bool accumulate);
int conv_bf16_v2_fwd(const ConvGeom& g, const bf16_t* x, const bf16_t* w, const float* bias, const bf16_t* residual,
                     bf16_t* y, int act, int ps_r, void* stream);
int conv_bf16_v2_dgrad(const ConvGeom& g, const bf16_t* dconv, const bf16_t* w, bf16_t* dx, int accumulate, void* stream);
// the same kernels as their fp32 namesakes, instantiated for bf16 activations
int conv_epilogue_bwd_bf16(const ConvGeom& g, const bf16_t* dy, const bf16_t* y, bf16_t* dconv, int act, int ps_r,
                           void* stream);
int conv_c1_fwd_bf16(const ConvGeom& g, const float* x, const float* w, const float* bias, bf16_t* y, int act, void* stream);
int conv_c1_wgrad_bf16(const ConvGeom& g, const float* x, const bf16_t* dy, const bf16_t* yact, int act, float* dw,
                       float* dbias, void* stream);
// conv9_bf16_mfma.hip (the 9x9 output convolution on the bf16 matrix cores: bf16 x / dx, fp32 w / y / dy / dw)
int conv9_bf16_fwd(const ConvGeom& g, const bf16_t* x, const float* w, const float* bias, float* y, void* stream);
int conv9_bf16_dgrad(const ConvGeom& g, const float* dconv, const float* w, bf16_t* dx, int accumulate, void* stream);
size_t conv9_bf16_wgrad_workspace(const ConvGeom& g);
int conv9_bf16_wgrad(const ConvGeom& g, const bf16_t* x, const float* dconv, float* dw, void* workspace, void* stream);

// conv9_split.hip (the 9x9 output convolution of the fp32 path as fp16 x 2 split products)
bool conv9_split_supported(const ConvGeom& g);
int conv9_split_fwd(const ConvGeom& g, const float* x, const float* xmax, const float* w, const float* wmax, const float* bias,
                    float* y, void* stream);
int conv9_split_dgrad(const ConvGeom& g, const float* dconv, const float* dmax, const float* w, const float* wmax, float* dx,
                      int accumulate, void* stream);
size_t conv9_split_wgrad_workspace(const ConvGeom& g);
int conv9_split_wgrad(const ConvGeom& g, const float* x, const float* xmax, const float* dconv, const float* dmax, float* dw,
                      void* workspace, void* stream);
