// conv_api.hip — C-ABI entry points of the convolution family and dispatch between the generic direct
// kernels (conv_direct.hip) and the MFMA implicit-GEMM kernels for the hot shapes (conv_mfma.hip).
#include "dasr_common.h"
#include "conv_kernels.h"

static int check_geom(const ConvGeom& g) {
    if (g.B <= 0 || g.H <= 0 || g.W <= 0 || g.Cin <= 0 || g.Ho <= 0 || g.Wo <= 0 || g.Cout <= 0 || g.KH <= 0 ||
        g.KW <= 0 || g.stride <= 0 || g.pad < 0)
        return DASR_E_SHAPE;
    int eh, ew;
    if (!g.transposed) {
        eh = (g.H + 2 * g.pad - g.KH) / g.stride + 1;
        ew = (g.W + 2 * g.pad - g.KW) / g.stride + 1;
    } else {
        eh = (g.H - 1) * g.stride - 2 * g.pad + g.KH;
        ew = (g.W - 1) * g.stride - 2 * g.pad + g.KW;
    }
    if (eh != g.Ho || ew != g.Wo) return DASR_E_SHAPE;
    return DASR_OK;
}

static int conv2d_fwd_dispatch(const ConvGeom& g, const float* x, const float* w, const float* bias, const float* residual,
                               float* y, int act, int ps_r, void* stream) {
    if (conv_mfma_supported(g)) return conv_mfma_fwd(g, x, w, bias, residual, y, act, ps_r, stream);
    if (conv_c1_supported(g) && ps_r == 1 && !residual) return conv_c1_fwd(g, x, w, bias, y, act, stream);
    if (conv9_mfma_supported(g) && act == DASR_ACT_NONE && ps_r == 1 && !residual)
        return conv9_mfma_fwd(g, x, w, bias, y, stream);
    if (conv_gather_fwd_supported(g) && ps_r == 1 && !residual) return conv_gather_fwd(g, x, w, bias, y, act, stream);
    return conv_direct_fwd(g, x, w, bias, residual, y, act, ps_r, stream);
}
extern "C" int dasr_conv2d_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y,
                               float* y_amax, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                               int stride, int pad, int transposed, int act, int ps_r, void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(w); DASR_CHECK_PTR(y);
    ConvGeom g{B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed};
    int rc = check_geom(g);
    if (rc) return rc;
    if (act < 0 || act > 2) return DASR_E_UNSUPPORTED;
    if (ps_r < 1) ps_r = 1;
    if (ps_r > 1 && (Cout % (ps_r * ps_r)) != 0) return DASR_E_SHAPE;
    if (!y_amax) return conv2d_fwd_dispatch(g, x, w, bias, residual, y, act, ps_r, stream);
    // max |y| for the consumer's fp16 x 2 split convolution: the mask layer's kernel tracks it while it stores (the one
    // producer of such a tensor on this entry point in the net); every other kernel is followed by a pass over y
    if (!conv_mfma_supported(g) && conv_c1_supported(g) && ps_r == 1 && !residual)
        return conv_c1_fwd(g, x, w, bias, y, act, stream, y_amax);
    rc = conv2d_fwd_dispatch(g, x, w, bias, residual, y, act, ps_r, stream);
    if (rc) return rc;
    return absmax_raise(y, (size_t)B * Ho * Wo * Cout, y_amax, stream);
}

// conv 3x3 / stride 1 / pad 1 (+bias) and the InstanceNorm statistics of its output in one pass (the DGB convs)
extern "C" size_t dasr_conv2d_fwd_stats_workspace(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    ConvGeom g{B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 0};
    size_t fused = conv_mfma_fwd_stats_supported(g) ? conv_mfma_fwd_stats_workspace(g) : 0;
    size_t plain = dasr_instnorm_stats_workspace(B, H * W, Cout);
    return fused > plain ? fused : plain;
}
extern "C" int dasr_conv2d_fwd_stats(const float* x, const float* w, const float* bias, float* y, float* mean, float* var,
                                     void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout,
                                     void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(w); DASR_CHECK_PTR(y); DASR_CHECK_PTR(mean); DASR_CHECK_PTR(var);
    DASR_CHECK_PTR(workspace);
    ConvGeom g{B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 0};
    int rc = check_geom(g);
    if (rc) return rc;
    if (workspace_bytes < dasr_conv2d_fwd_stats_workspace(B, H, W, Cin, Cout)) return DASR_E_WORKSPACE;
    if (conv_mfma_fwd_stats_supported(g)) return conv_mfma_fwd_stats(g, x, w, bias, y, mean, var, workspace, stream);
    rc = dasr_conv2d_fwd(x, w, bias, nullptr, y, nullptr, B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 0, DASR_ACT_NONE, 1, stream);
    if (rc) return rc;
    return dasr_instnorm_stats(y, mean, var, workspace, workspace_bytes, B, H * W, Cout, stream);
}

extern "C" int dasr_conv2d_epilogue_bwd(const float* dy, const float* y, float* dconv, float* dconv_amax, int B, int Ho, int Wo,
                                        int Cout, int act, int ps_r, void* stream) {
    DASR_CHECK_PTR(dy); DASR_CHECK_PTR(y); DASR_CHECK_PTR(dconv);
    DASR_CHECK_SHAPE(B > 0 && Ho > 0 && Wo > 0 && Cout > 0);
    if (act < 0 || act > 2) return DASR_E_UNSUPPORTED;
    if (ps_r < 1) ps_r = 1;
    if (ps_r > 1 && (Cout % (ps_r * ps_r)) != 0) return DASR_E_SHAPE;
    ConvGeom g{B, Ho, Wo, 1, Ho, Wo, Cout, 1, 1, 1, 0, 0};
    return conv_epilogue_bwd(g, dy, y, dconv, act, ps_r, stream, dconv_amax);
}

extern "C" int dasr_conv2d_dgrad(const float* dconv, const float* w, float* dx, int accumulate, int B, int H, int W,
                                 int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                                 int transposed, void* stream) {
    DASR_CHECK_PTR(dconv); DASR_CHECK_PTR(w); DASR_CHECK_PTR(dx);
    ConvGeom g{B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed};
    int rc = check_geom(g);
    if (rc) return rc;
    if (conv_mfma_dgrad_supported(g)) return conv_mfma_dgrad(g, dconv, w, dx, accumulate, nullptr, 0, 1, stream);
    if (conv9_mfma_supported(g)) return conv9_mfma_dgrad(g, dconv, w, dx, accumulate, nullptr, 0, 1, stream);
    if (conv_gather_dgrad_supported(g)) return conv_gather_dgrad(g, dconv, w, dx, accumulate, stream);
    return conv_direct_dgrad(g, dconv, w, dx, accumulate, stream);
}

extern "C" size_t dasr_conv2d_wgrad_workspace(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                                              int stride, int pad, int transposed) {
    ConvGeom g{B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed};
    if (check_geom(g) == DASR_OK && conv_mfma_wgrad_supported(g)) return conv_mfma_wgrad_workspace(g);
    if (check_geom(g) == DASR_OK && conv9_mfma_supported(g)) return conv9_mfma_wgrad_workspace(g);
    if (check_geom(g) == DASR_OK && conv_gather_wgrad_supported(g)) return conv_gather_wgrad_workspace(g);
    return 0;
}

extern "C" int dasr_conv2d_wgrad(const float* x, const float* dconv, float* dw, float* dbias, void* workspace,
                                 size_t workspace_bytes, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH,
                                 int KW, int stride, int pad, int transposed, void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(dconv); DASR_CHECK_PTR(dw);
    ConvGeom g{B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed};
    int rc = check_geom(g);
    if (rc) return rc;
    if (conv_c1_supported(g)) return conv_c1_wgrad(g, x, dconv, nullptr, DASR_ACT_NONE, dw, dbias, stream);
    if (conv_mfma_wgrad_supported(g)) {
        if (!workspace) return DASR_E_NULL;
        if (workspace_bytes < conv_mfma_wgrad_workspace(g)) return DASR_E_WORKSPACE;
        return conv_mfma_wgrad(g, x, dconv, dw, dbias, workspace, stream);
    } else if (conv9_mfma_supported(g)) {
        if (!workspace) return DASR_E_NULL;
        if (workspace_bytes < conv9_mfma_wgrad_workspace(g)) return DASR_E_WORKSPACE;
        rc = conv9_mfma_wgrad(g, x, dconv, dw, workspace, stream);
    } else if (conv_gather_wgrad_supported(g)) {
        if (!workspace) return DASR_E_NULL;
        if (workspace_bytes < conv_gather_wgrad_workspace(g)) return DASR_E_WORKSPACE;
        rc = conv_gather_wgrad(g, x, dconv, dw, workspace, stream);
    } else {
        rc = conv_direct_wgrad(g, x, dconv, dw, stream);
    }
    if (rc) return rc;
    if (dbias) rc = conv_colsum(dconv, dbias, (size_t)B * Ho * Wo, Cout, stream);
    return rc;
}

extern "C" int dasr_conv2d_wgrad_act(const float* x, const float* dy, const float* y, float* dw, float* dbias,
                                     float* dconv_scratch, void* workspace, size_t workspace_bytes, int B, int H, int W,
                                     int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                                     int transposed, int act, void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(dy); DASR_CHECK_PTR(y); DASR_CHECK_PTR(dw);
    ConvGeom g{B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed};
    int rc = check_geom(g);
    if (rc) return rc;
    if (act < 0 || act > 2) return DASR_E_UNSUPPORTED;
    if (conv_c1_supported(g)) return conv_c1_wgrad(g, x, dy, y, act, dw, dbias, stream);
    // no fused kernel for this shape: materialise dconv, then the ordinary weight gradient
    DASR_CHECK_PTR(dconv_scratch);
    rc = conv_epilogue_bwd(g, dy, y, dconv_scratch, act, 1, stream);
    if (rc) return rc;
    return dasr_conv2d_wgrad(x, dconv_scratch, dw, dbias, workspace, workspace_bytes, B, H, W, Cin, Ho, Wo, Cout, KH, KW,
                             stride, pad, transposed, stream);
}

// 1 when dasr_conv2d_wgrad_act runs its fused kernel for this geometry (no dconv_scratch needed), else 0
extern "C" int dasr_conv2d_wgrad_act_fused(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                                           int stride, int pad, int transposed) {
    ConvGeom g{B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed};
    if (check_geom(g) != DASR_OK) return 0;
    return conv_c1_supported(g) ? 1 : 0;
}

// 1 when dasr_conv2d_dgrad_act has a kernel for this geometry
extern "C" int dasr_conv2d_dgrad_act_supported(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                                               int stride, int pad, int transposed, int ps_r) {
    ConvGeom g{B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed};
    if (check_geom(g) != DASR_OK) return 0;
    if (ps_r < 1) ps_r = 1;
    if ((H % ps_r) != 0 || (W % ps_r) != 0) return 0;
    return (conv_mfma_dgrad_supported(g) || conv9_mfma_supported(g)) ? 1 : 0;
}

extern "C" int dasr_conv2d_dgrad_act(const float* dconv, const float* w, const float* x_act, float* dprev, int B, int H,
                                     int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                                     int transposed, int act, int ps_r, void* stream) {
    DASR_CHECK_PTR(dconv); DASR_CHECK_PTR(w); DASR_CHECK_PTR(x_act); DASR_CHECK_PTR(dprev);
    if (act < 0 || act > 2) return DASR_E_UNSUPPORTED;
    if (!dasr_conv2d_dgrad_act_supported(B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed, ps_r))
        return DASR_E_UNSUPPORTED;
    ConvGeom g{B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed};
    if (conv_mfma_dgrad_supported(g)) return conv_mfma_dgrad(g, dconv, w, dprev, 0, x_act, act, ps_r, stream);
    return conv9_mfma_dgrad(g, dconv, w, dprev, 0, x_act, act, ps_r, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Mixed-precision entry points (bf16 activations; BASELINE.json configs[2..3]).  Three layer kinds exist on that path,
// told apart by the geometry; anything else returns DASR_E_UNSUPPORTED (the bf16 path has no generic fallback):
//   trunk   3x3 / stride 1 / pad 1, Cin % 32 == 0, Cout % 32 == 0   x bf16, w packed bf16, y / residual bf16
//   mask    3x3, Cin == 1 (SEAN.mlp_mask on the fp32 depth map)     x f32,  w packed f32,  y bf16
//   output  9x9 / pad 4, Cout <= 3 (conv_output)                    x bf16, w packed f32,  y f32
// bias, dw, dbias are always fp32.
static int bf16_kind(const ConvGeom& g) {
    if (conv_bf16_supported(g)) return 1;
    if (conv_c1_supported(g)) return 2;
    if (conv9_mfma_supported(g)) return 3;
    return 0;
}
extern "C" int dasr_conv2d_fwd_bf16(const void* x, const void* w, const float* bias, const unsigned short* residual,
                                    void* y, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                                    int stride, int pad, int transposed, int act, int ps_r, void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(w); DASR_CHECK_PTR(y);
    ConvGeom g{B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed};
    int rc = check_geom(g);
    if (rc) return rc;
    if (act < 0 || act > 2) return DASR_E_UNSUPPORTED;
    if (ps_r < 1) ps_r = 1;
    if (ps_r > 1 && (Cout % (ps_r * ps_r)) != 0) return DASR_E_SHAPE;
    switch (bf16_kind(g)) {
        case 1:
            return conv_bf16_fwd(g, (const bf16_t*)x, (const bf16_t*)w, bias, (const bf16_t*)residual, (bf16_t*)y, act, ps_r,
                                 stream);
        case 2:
            if (ps_r != 1 || residual) return DASR_E_UNSUPPORTED;
            return conv_c1_fwd_bf16(g, (const float*)x, (const float*)w, bias, (bf16_t*)y, act, stream);
        case 3:
            if (ps_r != 1 || residual || act != DASR_ACT_NONE) return DASR_E_UNSUPPORTED;
            return conv9_bf16_fwd(g, (const bf16_t*)x, (const float*)w, bias, (float*)y, stream);
    }
    return DASR_E_UNSUPPORTED;
}
extern "C" int dasr_conv2d_epilogue_bwd_bf16(const unsigned short* dy, const unsigned short* y, unsigned short* dconv,
                                             int B, int Ho, int Wo, int Cout, int act, int ps_r, void* stream) {
    DASR_CHECK_PTR(dy); DASR_CHECK_PTR(y); DASR_CHECK_PTR(dconv);
    DASR_CHECK_SHAPE(B > 0 && Ho > 0 && Wo > 0 && Cout > 0);
    if (act < 0 || act > 2) return DASR_E_UNSUPPORTED;
    if (ps_r < 1) ps_r = 1;
    if (ps_r > 1 && (Cout % (ps_r * ps_r)) != 0) return DASR_E_SHAPE;
    ConvGeom g{B, Ho, Wo, 1, Ho, Wo, Cout, 1, 1, 1, 0, 0};
    return conv_epilogue_bwd_bf16(g, (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)dconv, act, ps_r, stream);
}
// trunk: dconv bf16, w packed bf16, dx bf16;  output conv: dconv f32, w packed f32, dx bf16
extern "C" int dasr_conv2d_dgrad_bf16(const void* dconv, const void* w, unsigned short* dx, int accumulate, int B, int H,
                                      int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                                      int transposed, void* stream) {
    DASR_CHECK_PTR(dconv); DASR_CHECK_PTR(w); DASR_CHECK_PTR(dx);
    ConvGeom g{B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed};
    int rc = check_geom(g);
    if (rc) return rc;
    if (conv_bf16_dgrad_supported(g))
        return conv_bf16_dgrad(g, (const bf16_t*)dconv, (const bf16_t*)w, (bf16_t*)dx, accumulate, stream);
    if (conv9_mfma_supported(g))
        return conv9_bf16_dgrad(g, (const float*)dconv, (const float*)w, (bf16_t*)dx, accumulate, stream);
    return DASR_E_UNSUPPORTED;
}
extern "C" size_t dasr_conv2d_wgrad_workspace_bf16(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                                                   int stride, int pad, int transposed) {
    ConvGeom g{B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed};
    if (check_geom(g) != DASR_OK) return 0;
    if (conv_bf16_wgrad_supported(g)) return conv_bf16_wgrad_workspace(g);
    if (conv9_mfma_supported(g)) return conv9_bf16_wgrad_workspace(g);
    return 0;
}
// trunk: x bf16, dconv bf16;  output conv: x bf16, dconv f32.  dw (plain HWIO) and dbias are fp32.
extern "C" int dasr_conv2d_wgrad_bf16(const unsigned short* x, const void* dconv, float* dw, float* dbias, void* workspace,
                                      size_t workspace_bytes, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH,
                                      int KW, int stride, int pad, int transposed, void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(dconv); DASR_CHECK_PTR(dw); DASR_CHECK_PTR(workspace);
    ConvGeom g{B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed};
    int rc = check_geom(g);
    if (rc) return rc;
    if (conv_bf16_wgrad_supported(g)) {
        if (workspace_bytes < conv_bf16_wgrad_workspace(g)) return DASR_E_WORKSPACE;
        return conv_bf16_wgrad(g, (const bf16_t*)x, (const bf16_t*)dconv, dw, dbias, workspace, stream);
    }
    if (conv9_mfma_supported(g)) {
        if (workspace_bytes < conv9_bf16_wgrad_workspace(g)) return DASR_E_WORKSPACE;
        rc = conv9_bf16_wgrad(g, (const bf16_t*)x, (const float*)dconv, dw, workspace, stream);
        if (rc) return rc;
        if (dbias) rc = conv_colsum((const float*)dconv, dbias, (size_t)B * Ho * Wo, Cout, stream);
        return rc;
    }
    return DASR_E_UNSUPPORTED;
}
// SEAN.mlp_mask: weight / bias gradient with the ReLU backward fused; x = depth map (f32), dy / y bf16
extern "C" int dasr_conv2d_wgrad_act_bf16(const float* x, const unsigned short* dy, const unsigned short* y, float* dw,
                                          float* dbias, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH,
                                          int KW, int stride, int pad, int transposed, int act, void* stream) {
    DASR_CHECK_PTR(x); DASR_CHECK_PTR(dy); DASR_CHECK_PTR(y); DASR_CHECK_PTR(dw);
    ConvGeom g{B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed};
    int rc = check_geom(g);
    if (rc) return rc;
    if (act < 0 || act > 2) return DASR_E_UNSUPPORTED;
    if (!conv_c1_supported(g)) return DASR_E_UNSUPPORTED;
    return conv_c1_wgrad_bf16(g, x, (const bf16_t*)dy, (const bf16_t*)y, act, dw, dbias, stream);
}
