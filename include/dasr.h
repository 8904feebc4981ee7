/*
 * dasr.h — C ABI of the MI355X-native DepthNet hot path (libdasr_hip.so).
 *
 * The reference (CUHK-AIM-Group/Depth-Aware-Endoscopy-SR) is pure Python: the generator
 * DepthNet (codes/models/modules/sftmd_arch.py:837-950) and its SEAN/DFN normalisation
 * (codes/models/modules/normalization.py:7-92) dispatch every device op through torch.nn
 * (cuDNN/cuBLAS/ATen). It has no FFI of its own; these entry points are what a binding
 * for this path replaces, one per group of torch ops, each citing the reference lines.
 * INTEGRATION.md shows the ctypes stub a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain pointers and sizes, no torch types; every pointer is DEVICE memory (HBM),
 *     fp32 unless stated; the caller owns all buffers (outputs and workspaces included).
 *   - activations are NHWC  [B][H][W][C]  (C fastest); the NCHW tensors of the reference
 *     API (input image, depth masks, output image) are converted at the edge by
 *     dasr_nchw_to_nhwc / dasr_clamp_to_nchw or read in place (masks).
 *   - convolution kernels are packed by dasr_weight_pack_fwd into 2*KH*KW*Cin*Cout floats: "HWIO"
 *     [KH][KW][Cin][Cout] followed by the per-tap transpose [KH][KW][Cout][Cin] (so that forward and
 *     data-gradient kernels both stream their K axis contiguously).  Every `w_hwio` argument of the
 *     convolution entry points is such a packed buffer; weight GRADIENTS (dw_hwio) are plain HWIO.
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it, never
 *     synchronise, never allocate, never throw.  Re-entrant: no global mutable state.
 *   - return value: 0 on success, a DASR_E_* code (<0) for bad arguments, or a positive
 *     hipError_t from the launch.  dasr_error_string() names the DASR_E_* codes.
 */
#ifndef DASR_H
#define DASR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DASR_OK 0
#define DASR_E_NULL (-1)        /* a required pointer is NULL                         */
#define DASR_E_SHAPE (-2)       /* sizes inconsistent with each other or non-positive  */
#define DASR_E_UNSUPPORTED (-3) /* valid request this build has no kernel for          */
#define DASR_E_WORKSPACE (-4)   /* workspace too small                                 */

/* activation / epilogue selectors for the convolution entry points */
#define DASR_ACT_NONE 0
#define DASR_ACT_RELU 1
#define DASR_ACT_LRELU02 2 /* LeakyReLU(0.2): sftmd_arch.py:741,861-866,891-908 */

int dasr_version(void);
const char* dasr_error_string(int code);
/* 1 when the library was built for the GPU (gfx950), 0 for the CPU kernel emulator used by unit tests */
int dasr_is_device_build(void);

/* ---- layout at the API edge ------------------------------------------------------------ */
/* [B,C,H,W] -> [B,H,W,C]; replaces nothing in the reference (it is NCHW throughout). */
int dasr_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, void* stream);
int dasr_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, void* stream);
/* torch.clamp(out, min, max) (sftmd_arch.py:950) fused with the NHWC->NCHW edge conversion. */
int dasr_clamp_to_nchw(const float* y_nhwc, float* out_nchw, int B, int C, int H, int W, float lo, float hi,
                       void* stream);
/* backward of the above: dy = dout where lo <= y <= hi, else 0. */
int dasr_clamp_to_nchw_bwd(const float* dout_nchw, const float* y_nhwc, float* dy_nhwc, int B, int C, int H, int W,
                           float lo, float hi, void* stream);
/* F.interpolate(mode='nearest') of an NCHW tensor (normalization.py:58-59). */
int dasr_resize_nearest_nchw(const float* src, float* dst, int BC, int h, int w, int H, int W, void* stream);
/* The same index map applied to the region bytes [B,h,w] -> [B,H,W] of one-hot masks (nearest resize of the K planes
 * == one-hot of the resized region byte): SEAN's mask resize (normalization.py:59) without touching the planes. */
int dasr_resize_nearest_u8(const unsigned char* src, unsigned char* dst, int B, int h, int w, int H, int W,
                           void* stream);

/* ---- weights -------------------------------------------------------------------------------
 * torch.nn.utils.weight_norm, dim 0 (sftmd_arch.py:741,851 and every wn(...) site):
 *   w = g * v / ||v||  with the norm over all dims but 0.
 * v is [O][I][KH][KW] for Conv2d, [I][O][KH][KW] for ConvTranspose2d (transposed=1: dim 0 is the
 * IN-channel axis, encoder.layer4).  g == NULL packs a plain weight (no normalisation).
 * Output: w_hwio = [KH][KW][I][ldo] then [KH][KW][ldo][I] (2*KH*KW*I*ldo floats), written at output-channel offset o_off (ldo >= o_off + O lets several
 * modules share one packed kernel: mlp_gamma_o | mlp_beta_o, normalization.py:41-42, run as one 2C->2C conv);
 * inv_norm [O or I] (1/||v||, kept for the backward; may be NULL when g is).
 */
int dasr_weight_pack_fwd(const float* v, const float* g, float* w_hwio, float* inv_norm, int O, int I, int KH, int KW,
                         int transposed, int ldo, int o_off, void* stream);
/* Every kernel of a network in one launch (a training step re-packs ~120 weight tensors: one launch instead of ~120).
 * One job = one dasr_weight_pack_fwd call; `jobs_host` and `jobs_device` hold the same njobs entries in host / device memory
 * (the host copy is validated, the kernel reads the device copy; the caller keeps both alive until the launch has run).
 *   w, bf16   : the packed kernel, float (bf16 = 0) or bf16 bits (bf16 = 1, as dasr_weight_pack_fwd_bf16)
 *   KK        : KH * KW
 *   amax      : NULL, or an amax buffer (DASR_AMAX_FLOATS floats, see "amax buffers" below) of the packed kernel: the job's norm
 *               groups leave their max |w| in parts o_off .. o_off + O - 1 and the part count ldo (so that jobs sharing one
 *               packed kernel fill one buffer) - what dasr_conv3x3_split2_weights / the 9x9 split kernels take as wmax, without
 *               a dasr_absmax pass; conv kernels only (transposed = 0), ldo <= DASR_AMAX_MAX_PARTS
 *   plain     : 1 = write the HWIO half only (two bias vectors side by side: O = n, I = KK = 1)
 *   wg_begin  : sum of the norm-group counts (O, or I when transposed) of the jobs before this one */
typedef struct dasr_pack_job {
    const float* v; const float* g; void* w; float* inv_norm; float* amax;
    int O, I, KK, transposed, ldo, o_off, bf16, plain, wg_begin, reserved;
} dasr_pack_job;
int dasr_weight_pack_multi(const dasr_pack_job* jobs_host, const dasr_pack_job* jobs_device, int njobs, void* stream);
/* given dW (HWIO): dv (layout of v) and dg ([O or I]); g == NULL: dv = unpacked dW, dg untouched. */
int dasr_weight_pack_bwd(const float* dw_hwio, const float* v, const float* g, const float* inv_norm, float* dv,
                         float* dg, int O, int I, int KH, int KW, int transposed, int ldo, int o_off, void* stream);

/* ---- convolution (NHWC, HWIO) -------------------------------------------------------------
 * Replaces nn.Conv2d / nn.ConvTranspose2d (+ bias, + LeakyReLU/ReLU, + nn.PixelShuffle, + the
 * residual add of Classic_Residual_Block) at sftmd_arch.py:743-749,811-820,861-866,133-151,891-910
 * and normalization.py:37-42.
 *   transposed = 0:  y[b,oy,ox,co] = bias[co] + sum x[b, oy*stride-pad+kh, ox*stride-pad+kw, ci] * w[kh,kw,ci,co]
 *   transposed = 1:  ConvTranspose2d geometry, Ho = (H-1)*stride - 2*pad + KH.
 * Epilogue: (+ residual[b,oy,ox,co]) -> act -> PixelShuffle(ps_r):
 *   ps_r > 1 writes y as [B][Ho*r][Wo*r][Cout/r^2] with
 *   y[b, oy*r+i, ox*r+j, c] = act(conv[b,oy,ox, c*r*r + i*r + j])   (bit-exact index map of nn.PixelShuffle).
 * bias, residual may be NULL.
 * y_amax (may be NULL): amax buffer (see dasr_absmax) for max |y|, filled by the mask layer's kernel while it stores and
 * by a pass over y after every other kernel.
 */
int dasr_conv2d_fwd(const float* x, const float* w_hwio, const float* bias, const float* residual, float* y, float* y_amax,
                    int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                    int transposed, int act, int ps_r, void* stream);
/* dasr_conv2d_fwd for a 3x3 / stride 1 / pad 1 convolution with bias and no activation, PLUS the statistics
 * nn.InstanceNorm2d(affine=False) needs of its output (sftmd_arch.py:811-820: `conv1 = Sequential(Conv2d, InstanceNorm2d)`):
 * mean[b,c], var[b,c] (biased) over the H*W pixels, as dasr_instnorm_stats(y) returns them.  On the MFMA path the
 * per-tile (mean, M2) pairs come out of the convolution's epilogue (no second pass over y) and are merged in tile
 * order with Chan's update; other shapes run the two entry points back to back.
 * workspace: dasr_conv2d_fwd_stats_workspace() bytes. */
size_t dasr_conv2d_fwd_stats_workspace(int B, int H, int W, int Cin, int Cout);
int dasr_conv2d_fwd_stats(const float* x, const float* w, const float* bias, float* y, float* mean, float* var,
                          void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, void* stream);

/* Backward of the epilogue: dconv[b,oy,ox,cc] = dy[...shuffled...] * act'(y[...]) (y = saved forward output).
 * dconv_amax (may be NULL): amax buffer (see dasr_absmax) for max |dconv|, filled by the kernel. */
int dasr_conv2d_epilogue_bwd(const float* dy, const float* y, float* dconv, float* dconv_amax, int B, int Ho, int Wo,
                             int Cout, int act, int ps_r, void* stream);
/* dx (+)= conv-transpose of dconv with w; accumulate != 0 adds into dx. */
int dasr_conv2d_dgrad(const float* dconv, const float* w_hwio, float* dx, int accumulate, int B, int H, int W, int Cin,
                      int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, int transposed, void* stream);
/* Data gradient fused with the PRODUCER's epilogue backward.  When the conv input x is the activated (and possibly
 * pixel-shuffled) output of another convolution - x = PixelShuffle_r(act(prev)) - and this conv is its only
 * consumer, the gradient w.r.t. prev is  dprev = unshuffle_r( dgrad(dconv) * act'(x) ):  the activation mask is
 * applied in the dgrad epilogue and the store goes through the inverse PixelShuffle index map, so the producer
 * needs no dasr_conv2d_epilogue_bwd pass.  x_act = this conv's saved input [B,H,W,Cin];
 * dprev [B, H/r, W/r, Cin*r*r].  dasr_conv2d_dgrad_act_supported() tells whether a kernel exists. */
int dasr_conv2d_dgrad_act_supported(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                                    int pad, int transposed, int ps_r);
int dasr_conv2d_dgrad_act(const float* dconv, const float* w_hwio, const float* x_act, float* dprev, int B, int H, int W,
                          int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, int transposed,
                          int act, int ps_r, void* stream);
/* dw_hwio = sum_pixels x (x) dconv ; dbias[co] = sum dconv (dbias may be NULL).
 * workspace: dasr_conv2d_wgrad_workspace() bytes. */
size_t dasr_conv2d_wgrad_workspace(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                                   int pad, int transposed);
int dasr_conv2d_wgrad(const float* x, const float* dconv, float* dw_hwio, float* dbias, void* workspace,
                      size_t workspace_bytes, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                      int stride, int pad, int transposed, void* stream);

/* Weight/bias gradient with the activation backward fused in: dconv = dy * act'(y) is formed on the fly where a
 * fused kernel exists (Cin == 1: SEAN.mlp_mask; Cin == 3: the encoder's first layer) and is materialised into
 * dconv_scratch [B,Ho,Wo,Cout] otherwise (dconv_scratch may be NULL only where dasr_conv2d_wgrad_act_fused() returns 1).
 * For layers whose input needs no gradient. */
int dasr_conv2d_wgrad_act_fused(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                                int transposed);
int dasr_conv2d_wgrad_act(const float* x, const float* dy, const float* y, float* dw_hwio, float* dbias,
                          float* dconv_scratch, void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin,
                          int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, int transposed, int act,
                          void* stream);

/* ---- fp32 convolutions at fp32 accuracy on the BF16 matrix cores ("split bf16", csrc/conv_split_bf16.hip) ----------
 * For the plain (bias-only) 3x3 / stride 1 / pad 1 trunk convolutions of the fp32 path - mlp_gamma_o | mlp_beta_o
 * (normalization.py:41-42,73-74) and the DGB convolutions (sftmd_arch.py:811-820): every fp32 operand is the exact sum
 * of three bf16 pieces, six bf16 MFMAs per product term give the fp32 product to 2^-26, accumulated in fp32 - the
 * accuracy of the exact-fp32 MFMA kernels at 6/16 of their matrix time.  Replaces the same nn.Conv2d calls as
 * dasr_conv2d_fwd / dasr_conv2d_dgrad / dasr_conv2d_wgrad; Cin % 32 == 0, Cout % 32 == 0.
 *   split_weights: the fp32 packed kernel [2][3][3][Cin][Cout] -> a bf16 image (split_weights_bytes) holding, for the
 *                  forward and the dgrad, the three pieces of every K-step's slice contiguously (once per step)
 *   fwd_split:     y = act(conv(x, w) + bias + residual), optionally stored through PixelShuffle(2) (then no residual,
 *                  Cout % 128 == 0)                 x [B,H,W,Cin], y [B,H,W,Cout] fp32
 *   dgrad_split:   dx (+)= conv^T(dconv, w)         dconv [B,H,W,Cout], dx [B,H,W,Cin] fp32
 *   wgrad_split:   dw_hwio [3][3][Cin][Cout] = sum_pixels x (x) dconv, dbias (may be NULL) = sum dconv; workspace:
 *                  dasr_conv3x3_wgrad_split_workspace() bytes (replaces dasr_conv2d_wgrad for these layers) */
int dasr_conv3x3_split_supported(int H, int W, int Cin, int Cout);
size_t dasr_conv3x3_split_weights_bytes(int Cin, int Cout);
int dasr_conv3x3_split_weights(const float* w_packed, unsigned short* w_split, int Cin, int Cout, void* stream);
int dasr_conv3x3_fwd_split(const float* x, const unsigned short* w_split, const float* bias, const float* residual, float* y,
                           int B, int H, int W, int Cin, int Cout, int act, int ps_r, void* stream);
int dasr_conv3x3_dgrad_split(const float* dconv, const unsigned short* w_split, float* dx, int accumulate, int B, int H,
                             int W, int Cin, int Cout, void* stream);
size_t dasr_conv3x3_wgrad_split_workspace(int B, int H, int W, int Cin, int Cout);
int dasr_conv3x3_wgrad_split(const float* x, const float* dconv, float* dw_hwio, float* dbias, void* workspace,
                             size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, void* stream);

/* The same convolutions with TWO fp16 pieces per operand and THREE products per term ("split fp16 x 2", same source file):
 * fp16 holds 11 significant bits, a pair 22-23 of fp32's 24 (0.5 ulp RMS off - one more fp32 rounding per operand), the
 * dropped term is 2^-24 relative, and only half as many partial sums are rounded into the fp32 accumulator: measured
 * against float64 the result is closer than both the bf16 x 3 scheme and the exact-fp32 MFMA kernels.  fp16's narrow range
 * is handled by a power-of-two scale per tensor that puts its largest magnitude in [2^14, 2^15):
 *   dasr_absmax:     max |x| over n floats (x 16-byte aligned) into an AMAX BUFFER: DASR_AMAX_FLOATS floats of device
 *                    memory, word 0 = the number n of partial maxima that follow (int), words 1 .. n = one partial maximum
 *                    per workgroup of the producing launch.  No atomics, nothing to clear, nothing read back by the host:
 *                    the kernels below take the maximum of the parts in their prologue and derive the scale from its
 *                    exponent bits.  Every `*max` / `*_amax` argument of this header is such a buffer; the producer-side
 *                    ones (y_amax, out_amax, dt_amax, ...) may be NULL and are filled by the kernel that writes the
 *                    tensor - no extra pass over it - or, where that kernel does not keep a maximum, by a follow-up pass
 *                    inside the entry point
 *   split2_weights:  wmax = dasr_absmax of the packed kernel's plane 0 (9 Cin Cout floats); image: split2_weights_bytes
 *   fwd / dgrad / wgrad_split2: as the functions above plus the maxima of their tensor operands (xmax of x, dmax of dconv);
 *                    y_amax (may be NULL): amax buffer for max |y| of what the forward stores, for the NEXT split convolution
 * Same nn.Conv2d calls replaced (normalization.py:41-42,73-74; sftmd_arch.py:811-820 and the upscale tail :891-909). */
#define DASR_AMAX_MAX_PARTS 4096
#define DASR_AMAX_FLOATS (1 + DASR_AMAX_MAX_PARTS)
int dasr_absmax(const float* x, size_t n, float* amax, void* stream);
size_t dasr_conv3x3_split2_weights_bytes(int Cin, int Cout);
int dasr_conv3x3_split2_weights(const float* w_packed, const float* wmax, unsigned short* w_split, int Cin, int Cout,
                                void* stream);
/* ... of every split convolution of a network in one launch (as dasr_weight_pack_multi: host and device copy of the job table;
 * wg_begin = sum of dasr_conv3x3_split2_weights_slabs(Cin, Cout) of the jobs before) */
typedef struct dasr_split_job {
    const float* w_packed; const float* wmax; unsigned short* w_split;
    int Cin, Cout, wg_begin, reserved;
} dasr_split_job;
int dasr_conv3x3_split2_weights_slabs(int Cin, int Cout);
int dasr_conv3x3_split2_weights_multi(const dasr_split_job* jobs_host, const dasr_split_job* jobs_device, int njobs, void* stream);
int dasr_conv3x3_fwd_split2(const float* x, const float* xmax, const unsigned short* w_split, const float* wmax,
                            const float* bias, const float* residual, float* y, float* y_amax, int B, int H, int W, int Cin,
                            int Cout, int act, int ps_r, void* stream);
int dasr_conv3x3_dgrad_split2(const float* dconv, const float* dmax, const unsigned short* w_split, const float* wmax,
                              float* dx, int accumulate, int B, int H, int W, int Cin, int Cout, void* stream);
int dasr_conv3x3_wgrad_split2(const float* x, const float* xmax, const float* dconv, const float* dmax, float* dw_hwio,
                              float* dbias, void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout,
                              void* stream);

/* The 9x9 output convolution (conv_output, sftmd_arch.py:910,948: Cin % 32 == 0 -> Cout <= 3, pad 4) in the same fp16 x 2
 * scheme (csrc/conv9_split.hip): fp32 x / w / y / dy / dx / dw, operands split into two scaled fp16 pieces when they are
 * staged.  w_hwio = the packed kernel's plane 0 [9][9][Cin][Cout]; xmax / dmax / wmax: amax buffers (dasr_absmax) of x,
 * dconv and w_hwio.  Replaces dasr_conv2d_fwd / _dgrad / _wgrad for that layer (same results to fp32 accuracy). */
int dasr_conv9_split_supported(int H, int W, int Cin, int Cout);
int dasr_conv9_fwd_split2(const float* x, const float* xmax, const float* w_hwio, const float* wmax, const float* bias, float* y,
                          int B, int H, int W, int Cin, int Cout, void* stream);
int dasr_conv9_dgrad_split2(const float* dconv, const float* dmax, const float* w_hwio, const float* wmax, float* dx,
                            int accumulate, int B, int H, int W, int Cin, int Cout, void* stream);
size_t dasr_conv9_wgrad_split2_workspace(int B, int H, int W, int Cin, int Cout);
int dasr_conv9_wgrad_split2(const float* x, const float* xmax, const float* dconv, const float* dmax, float* dw_hwio, float* dbias,
                            void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, void* stream);

/* ---- instance-norm statistics ---------------------------------------------------------------
 * nn.InstanceNorm2d(affine=False) appears twice in a row on every DGB conv output
 * (sftmd_arch.py:811-820 then normalization.py:16-17,56).  Both collapse to one per-(b,c) scale:
 *   xhat = (x - mean) * rsqrt(var+eps) * rsqrt(var/(var+eps) + eps)      
 * This entry point computes mean and biased variance per (b,c) of an NHWC tensor
 * (chunked two-pass + Chan merge in a fixed order: no E[x^2]-mean^2 cancellation, bitwise reproducible).
 * workspace: dasr_instnorm_stats_workspace() bytes.
 */
size_t dasr_instnorm_stats_workspace(int B, int HW, int C);
int dasr_instnorm_stats(const float* x, float* mean, float* var, void* workspace, size_t workspace_bytes, int B, int HW,
                        int C, void* stream);

/* ---- depth matrix -> per-sample dynamic kernels ------------------------------------------------
 * SEAN.A_i_j (1x1 conv over the K axis) and the collapse of mlp_gamma_s / mlp_beta_s over the
 * style map (normalization.py:27-29,80-85):
 *   stp[b,k,l]       = A_b[k] + sum_j A_w[k,j] * st[b,j,l]
 *   D[b,s,tap,k,c]   = sum_l W_s[c,l,tap] * stp[b,k,l]          s = 0 (gamma), 1 (beta); tap = kh*3+kw
 * W_gamma / W_beta are the module's OIHW tensors [C][L][3][3].
 */
int dasr_dynk_fwd(const float* st, const float* A_w, const float* A_b, const float* W_gamma, const float* W_beta,
                  float* stp, float* D, int B, int K, int L, int C, void* stream);
/* Given dD: dW_gamma, dW_beta (written), dA_w, dA_b (written), dst (ACCUMULATED: the depth matrix feeds every SEAN).
 * dstp is a [B,K,L] scratch. */
int dasr_dynk_bwd(const float* dD, const float* st, const float* stp, const float* A_w, const float* W_gamma,
                  const float* W_beta, float* dW_gamma, float* dW_beta, float* dA_w, float* dA_b, float* dst,
                  float* dstp, int B, int K, int L, int C, void* stream);

/* ---- the Depth-Guided Block's dynamic convolution + DFN modulation (north-star kernel) ----------
 * Replaces, per SEAN call (normalization.py:56,59,80-89) and the surrounding ReLU / residual of
 * Depth_Residual_Block_Mask.forward (sftmd_arch.py:826-834):
 *   xhat   = double instance norm of t (stats from dasr_instnorm_stats)
 *   gamma1 = bias_gamma[c] + sum_{tap,k} mask[b,k,p+tap] * D[b,0,tap,k,c]       (zero padding)
 *   beta1  = bias_beta[c]  + sum_{tap,k} mask[b,k,p+tap] * D[b,1,tap,k,c]
 *   gamma  = a_g*gamma1 + (1-a_g)*gamma2 ;  beta = a_b*beta1 + (1-a_b)*beta2
 *   out    = xhat*(1+gamma) + beta  (+ residual)  -> ReLU if relu != 0
 * t, out, residual: NHWC [B,H,W,C]; gb2: NHWC [B,H,W,2C] (gamma2 | beta2 = mlp_gamma_o | mlp_beta_o outputs);
 * mask: NCHW [B,K,H,W] as the reference delivers it (any float values); bias_gamma/bias_beta [C] are the
 * mlp_gamma_s / mlp_beta_s biases; alpha_gamma / alpha_beta are 1-element DEVICE tensors (trainable, normalization.py:30-35).
 * region / onehot_flag: outputs of dasr_mask_compress for the same mask.  Both NULL: general kernel only.
 * Both given: both kernels are launched and the flag decides ON THE DEVICE which one works.  region given and
 * onehot_flag NULL: the caller has read the flag (== 0) itself and vouches for one-hot masks: gather kernel only.
 */
/* One byte per pixel from the K mask planes: region[b,y,x] = k if mask[b,k,y,x] == 1 and every other plane is 0,
 * K if all planes are 0; *onehot_flag is set to a non-zero value if ANY pixel is neither (soft / overlapping
 * masks).  The SEAN entry points run a gather kernel when the flag is 0 and the general kernel otherwise; the
 * choice is made on the device, the host never reads the flag. */
int dasr_mask_compress(const float* mask, unsigned char* region, int* onehot_flag, int B, int K, int H, int W,
                       void* stream);
int dasr_sean_fwd(const float* t, const float* mean, const float* var, const float* gb2, const float* mask,
                  const unsigned char* region, const int* onehot_flag, const float* D, const float* bias_gamma, const float* bias_beta, const float* alpha_gamma,
                  const float* alpha_beta, const float* residual, float* out, float* out_amax, int relu, int B, int H,
                  int W, int C, int K, float eps, void* stream);
/* Backward. Inputs as forward plus dout and the saved forward output `out` (for the ReLU mask).
 * Outputs: dt [B,H,W,C]; dgb2 [B,H,W,2C]; dD [B,2,9,K,C]; dbias_gamma, dbias_beta [C]; dalpha_gamma, dalpha_beta [1];
 * dres (may be NULL; written = dout*relu') ; workspace: dasr_sean_bwd_workspace() bytes.
 * out_amax / dt_amax / dgb2_amax (fp32 entry points, each may be NULL): amax buffers (see dasr_absmax) that receive
 * max |.| of the tensor just written, for the fp16 x 2 split convolution that reads it next. */
size_t dasr_sean_bwd_workspace(int B, int H, int W, int C, int K);
int dasr_sean_bwd(const float* dout, const float* out, const float* t, const float* mean, const float* var,
                  const float* gb2, const float* mask, const unsigned char* region, const int* onehot_flag,
                  const float* D, const float* bias_gamma, const float* bias_beta,
                  const float* alpha_gamma, const float* alpha_beta, float* dt, float* dgb2, float* dD,
                  float* dbias_gamma, float* dbias_beta, float* dalpha_gamma, float* dalpha_beta, float* dres,
                  float* dt_amax, float* dgb2_amax, void* workspace, size_t workspace_bytes, int relu, int B, int H, int W,
                  int C, int K, float eps, void* stream);

/* Largest K for which the soft-mask (general) kernels run forward and backward (their backward keeps two
 * [2][9][K][64] tables in LDS).  With more regions (up to 16) only one-hot masks are supported: pass region bytes with
 * onehot_flag NULL, i.e. read the flag of dasr_mask_compress yourself; otherwise dasr_sean_bwd returns
 * DASR_E_UNSUPPORTED. */
int dasr_sean_soft_mask_max_regions(void);

/* ---- region-wise average pooling (depth matrix) ---------------------------------------------------
 * RegionWiseAvgPooling.forward (sftmd_arch.py:714-733): masks are resized to the feature size with
 * bilinear(align_corners=True) and re-binarised (>= 0.5) when sizes differ; per region
 *   out[b,k,l] = sum_p m[b,k,p]*feat[b,p,l] / (sum_p m[b,k,p] + 1e-10).
 * feat NHWC [B,h,w,L]; mask NCHW [B,K,H,W]; maskr NCHW [B,K,h,w] and area [B,K] are written and
 * kept for the backward.
 */
int dasr_region_pool_fwd(const float* feat, const float* mask, float* maskr, float* area, float* out, int B, int K,
                         int L, int h, int w, int H, int W, void* stream);
int dasr_region_pool_bwd(const float* dout, const float* maskr, const float* area, float* dfeat, int B, int K, int L,
                         int h, int w, void* stream);

/* ---- device-side input preparation ------------------------------------------------------
 * Replaces LQGTker_Depth_dataset.getDepthMask (codes/data/LQGTker_Depth_dataset.py:204-225) + the host-side
 * stacking of the K float planes (:155-158,196): per sample, K equal-width half-open bins over the map's own
 * [min, max] (float32 arithmetic as torch evaluates it: interval = (max-min)/K, edge_i = min + interval*i); a
 * pixel equal to the maximum belongs to no bin.  depth: [B,HW] float32.  Outputs (either may be NULL, not both):
 * masks NCHW [B,K,HW] of 0/1 floats (what DepthNet.forward takes) and region bytes [B,HW] (k, or K for "no bin":
 * what the one-hot SEAN / loss kernels read - identical to dasr_mask_compress of those planes).
 * fixed_edges: NULL for the data-dependent range (depthFixedRange: false, every shipped yml); else K+1 DEVICE
 * floats holding the edges of the fixed [0,1] range (the reference computes those in Python doubles; the host
 * wrapper passes them rounded to float32, which is what torch compares against).
 * workspace: dasr_depth_to_masks_workspace() bytes (per-chunk minima / maxima). */
size_t dasr_depth_to_masks_workspace(int B, int HW);
int dasr_depth_to_masks(const float* depth, const float* fixed_edges, float* masks, unsigned char* region,
                        void* workspace, size_t workspace_bytes, int B, int HW, int K, void* stream);

/* ---- harness losses in one pass (one-hot masks only) -------------------------------------
 * nn.L1Loss + dynamic_weight_mask_loss('smoothl1') (F_model_depthCond.py:164,188-190; mask_loss.py:64-90) need, per
 * depth region k, sum smooth_l1(m_k*sr, m_k*hr) and sum m_k (masks nearest-upsampled to HR), and sum |sr-hr|.
 * sr, hr: NCHW [B,C,h*scale,w*scale]; region: bytes [B,h,w] from dasr_mask_compress.
 * sums [2K+1] = K numerators | K areas (C * #pixels) | sum|sr-hr|.  Division / softmax weights stay in PyTorch.
 * dasr_loss_bwd: dsr = dsums[2K]*sign(sr-hr) + dsums[region]*smooth_l1'(sr-hr)  (dsums: device [2K+1]). */
int dasr_loss_sums(const float* sr, const float* hr, const unsigned char* region, float* sums, int B, int C, int h,
                   int w, int scale, int K, void* stream);
int dasr_loss_bwd(const float* sr, const float* hr, const unsigned char* region, const float* dsums, float* dsr, int B,
                  int C, int h, int w, int scale, int K, void* stream);

/* ---- small elementwise helpers ---------------------------------------------------------------- */
int dasr_add(const float* a, const float* b, float* out, size_t n, void* stream);  /* torch.add, sftmd_arch.py:931 */
int dasr_accumulate(float* dst, const float* src, size_t n, void* stream);         /* dst += src (gradient fan-in) */
int dasr_copy(float* dst, const float* src, size_t n, void* stream);               /* device-to-device copy */

/* ==== mixed precision: bf16 activations (BASELINE.json configs[2..3]) ================================================
 * Replaces the same torch ops as the fp32 entry points when the reference is run under torch.autocast(bfloat16):
 * nn.Conv2d / weight_norm convs (sftmd_arch.py:811-820,891-910), SEAN (normalization.py:37-42,52-92).  Activation tensors
 * (unsigned short* = bf16 bits, NHWC) are bf16 in HBM; parameters, their gradients, instance-norm statistics, the dynamic
 * kernels D and every accumulator are fp32.  Same conventions as above (asynchronous, no allocation, return codes).
 * The bf16 path has no generic fallback kernels: unsupported geometries return DASR_E_UNSUPPORTED. */

/* dasr_weight_pack_fwd with a bf16 packed kernel (normalisation in fp32, one rounding); dasr_weight_pack_bwd serves both. */
int dasr_weight_pack_fwd_bf16(const float* v, const float* g, unsigned short* w, float* inv_norm, int O, int I, int KH, int KW,
                              int transposed, int ldo, int o_off, void* stream);
/* Convolution forward.  Three layer kinds, told apart by the geometry:
 *   trunk   3x3 / stride 1 / pad 1, Cin % 32 == 0, Cout % 32 == 0 : x bf16, w packed bf16, y / residual bf16
 *           (v_mfma_f32_32x32x16_bf16, fused bias / activation / PixelShuffle / residual as dasr_conv2d_fwd)
 *   mask    3x3, Cin == 1 (SEAN.mlp_mask on the depth map)         : x f32,  w packed f32,  y bf16
 *   output  9x9 / pad 4, Cout <= 3 (conv_output)                   : x bf16, w packed f32,  y f32 */
int dasr_conv2d_fwd_bf16(const void* x, const void* w, const float* bias, const unsigned short* residual, void* y, int B,
                         int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, int transposed,
                         int act, int ps_r, void* stream);
int dasr_conv2d_epilogue_bwd_bf16(const unsigned short* dy, const unsigned short* y, unsigned short* dconv, int B, int Ho,
                                  int Wo, int Cout, int act, int ps_r, void* stream);
/* Which kernel runs the trunk's forward / dgrad: 0 (default) = the persistent LDS-DMA kernel (csrc/conv_bf16_v2.hip) where
 * it applies (Cin % 32 == 0, Cout % 64 == 0, PixelShuffle 1 or 2), 1 = the first, register-staged kernel everywhere,
 * 2 = the persistent kernel with one workgroup per XCD (tests: long per-workgroup item lists on small shapes);
 * + 16 = force its 8-row / 4-wave form, + 32 = its 16-row / 8-wave form (default: by problem size);
 * + 64 = the fp16 x 2 split weight gradient on its first kernel (operands split per K-step; default: split at staging);
 * + 256 = the fp16 x 2 split forward / dgrad at 128 produced channels in its four-wave form (four tile rows per wave);
 * + 512 = ... at 64 produced channels with all nine kernel slices of a chunk per barrier; + 1024 = swap its two eight-wave
 * forms (128 produced channels: kernel-row K-steps instead of channel halves x kernel columns; 64: the reverse);
 * + 2048 = the bf16 dynamic-conv forward (dasr_sean_fwd_bf16) with 4 instead of 8 channels per lane;
 * + 4096 = the 9x9 split forward with the linear instead of the XCD-blocked tile order.
 * For A/B measurements and tests; process-wide, not thread-safe against concurrent launches.
 * dasr_conv_bf16_v2_launches(): how many times the persistent kernel has been launched by this process. */
int dasr_set_conv_bf16_impl(int impl);
int dasr_get_conv_bf16_impl(void);
int dasr_conv_bf16_v2_launches(void);
/* trunk: dconv bf16, w packed bf16; output conv: dconv f32, w packed f32.  dx bf16. */
int dasr_conv2d_dgrad_bf16(const void* dconv, const void* w, unsigned short* dx, int accumulate, int B, int H, int W, int Cin,
                           int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, int transposed, void* stream);
size_t dasr_conv2d_wgrad_workspace_bf16(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                                        int pad, int transposed);
/* trunk: x bf16, dconv bf16; output conv: x bf16, dconv f32.  dw (plain HWIO) and dbias fp32. */
int dasr_conv2d_wgrad_bf16(const unsigned short* x, const void* dconv, float* dw, float* dbias, void* workspace,
                           size_t workspace_bytes, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                           int stride, int pad, int transposed, void* stream);
/* mask layer: weight / bias gradient with the activation backward fused (x = depth map f32, dy / y bf16). */
int dasr_conv2d_wgrad_act_bf16(const float* x, const unsigned short* dy, const unsigned short* y, float* dw, float* dbias,
                               int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                               int transposed, int act, void* stream);
/* dasr_instnorm_stats / dasr_sean_fwd / dasr_sean_bwd on bf16 activations (t, gb2, residual, out and their gradients);
 * workspaces as for the fp32 entry points. */
int dasr_instnorm_stats_bf16(const unsigned short* x, float* mean, float* var, void* workspace, size_t workspace_bytes, int B,
                             int HW, int C, void* stream);
int dasr_sean_fwd_bf16(const unsigned short* t, const float* mean, const float* var, const unsigned short* gb2,
                       const float* mask, const unsigned char* region, const int* onehot_flag, const float* D,
                       const float* bias_gamma, const float* bias_beta, const float* alpha_gamma, const float* alpha_beta,
                       const unsigned short* residual, unsigned short* out, int relu, int B, int H, int W, int C, int K,
                       float eps, void* stream);
int dasr_sean_bwd_bf16(const unsigned short* dout, const unsigned short* out, const unsigned short* t, const float* mean,
                       const float* var, const unsigned short* gb2, const float* mask, const unsigned char* region,
                       const int* onehot_flag, const float* D, const float* bias_gamma, const float* bias_beta,
                       const float* alpha_gamma, const float* alpha_beta, unsigned short* dt, unsigned short* dgb2, float* dD,
                       float* dbias_gamma, float* dbias_beta, float* dalpha_gamma, float* dalpha_beta, unsigned short* dres,
                       void* workspace, size_t workspace_bytes, int relu, int B, int H, int W, int C, int K, float eps,
                       void* stream);
/* elementwise helpers (n % 4 == 0) and the casts at the fp32 encoder <-> bf16 trunk boundary */
int dasr_add_bf16(const unsigned short* a, const unsigned short* b, unsigned short* out, size_t n, void* stream);
int dasr_accumulate_bf16(unsigned short* dst, const unsigned short* src, size_t n, void* stream);
int dasr_cast_f32_to_bf16(const float* src, unsigned short* dst, size_t n, void* stream);
int dasr_cast_bf16_to_f32(const unsigned short* src, float* dst, int accumulate, size_t n, void* stream);

/* ---- the encoder's stride-2 layers on the bf16 stride-1 kernels (csrc/s2d.hip) -----------------------------------
 * Encoder (sftmd_arch.py:745-749, 771-783): nn.Conv2d(.., 3, stride=2, padding=1) x3 and
 * nn.ConvTranspose2d(.., 3, stride=2, padding=1) (no output_padding: (2H-1) x (2W-1) outputs).  With bf16 activations
 * they are expressed as stride-1 3x3 convolutions (dasr_conv2d_*_bf16) of a space-to-depth image / with a PixelShuffle(2)
 * epilogue (a 2H x 2W image whose last row and column the reference's layer does not have):
 *   space_to_depth2:        y[b][i][j][(2py+px)C + c] = x[b][2i+py][2j+px][c] for 2i+py < Hv, 2j+px < Wv, zero beyond
 *                           the VALID extents (Hv <= H, Wv <= W: H, W for a plain tensor; 2H'-1, 2W'-1 for the
 *                           PixelShuffle image of the transposed layer); x fp32 or bf16
 *   depth_to_space2_bwd:    the adjoint (gradient of the above: zero beyond the valid extents), into an fp32 or bf16
 *                           tensor, optionally accumulating
 *   weight_expand_s2:       fp32 HWIO [3][3][Cin][Cout] -> the bf16 packed kernel [2][3][3][4Cin][Cout] of the stride-1 form
 *   weight_collapse_s2:     fp32 gradient of the expanded kernel [3][3][4Cin][Cout] -> [3][3][Cin][Cout]
 *   weight_expand_t2 / weight_collapse_t2: the same for the transposed convolution: [3][3][Cin][Cout] <-> Cin -> 4 Cout
 *                           (output channel 4co + 2a + b = phase (a, b) of the PixelShuffle), bias repeated / summed.
 * All C % 4 == 0. */
int dasr_space_to_depth2_bf16(const void* x, int x_is_bf16, unsigned short* y, int B, int H, int W, int C, int Hv, int Wv,
                              void* stream);
int dasr_depth_to_space2_bwd_bf16(const unsigned short* dy, void* dx, int dx_is_bf16, int accumulate, int B, int H, int W,
                                  int C, int Hv, int Wv, void* stream);
int dasr_weight_expand_s2_bf16(const float* w_hwio, unsigned short* out, int Cin, int Cout, void* stream);
int dasr_weight_collapse_s2(const float* dw_expanded, float* dw_hwio, int Cin, int Cout, void* stream);
int dasr_weight_expand_t2_bf16(const float* w_hwio, const float* bias, unsigned short* out, float* bias_out, int Cin,
                               int Cout, void* stream);
int dasr_weight_collapse_t2(const float* dw_expanded, const float* dbias_expanded, float* dw_hwio, float* dbias, int Cin,
                            int Cout, void* stream);

/* ---- SSIM (validation metric) -------------------------------------------------------------------------------------
 * pytorch_ssim.ssim (reference codes/pytorch_ssim/__init__.py:17-37,65-73; train.py:239): 11 x 11 Gaussian windows
 * (sigma 1.5, zero padding), C1 = 0.01^2, C2 = 0.03^2, images [B,C,H,W] fp32 in [0,1].  One pass over both images:
 * out_per_sample[b] = mean over (C,H,W) of the SSIM map (their mean over b is size_average=True).  window11: HOST
 * pointer to the eleven normalised 1-D window weights.  workspace: dasr_ssim_workspace() bytes. */
size_t dasr_ssim_workspace(int B, int C, int H, int W);
int dasr_ssim(const float* img1, const float* img2, const float* window11, float* out_per_sample, void* workspace,
              size_t workspace_bytes, int B, int C, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DASR_H */
