"""Case definitions and closed-form inputs shared by oracle/make_golden.py (which runs
the reference) and the tests (which run the oracle and the HIP path)."""
import math

import numpy as np
import torch

import dasr_amd  # noqa: F401
from dasr_amd import synth

# ---- SEAN (normalization.py:52-92) -------------------------------------------------------
SEAN_CASES = [
    dict(name="onehot_L32_f32", B=2, C=64, K=10, L=32, H=16, W=20, dtype="float32", soft=False),
    dict(name="soft_L32_f32", B=2, C=64, K=10, L=32, H=16, W=20, dtype="float32", soft=True),
    dict(name="onehot_L256_f32", B=1, C=64, K=10, L=256, H=16, W=20, dtype="float32", soft=False),
    dict(name="onehot_L32_f64", B=2, C=64, K=10, L=32, H=16, W=20, dtype="float64", soft=False),
    dict(name="soft_C32_f64", B=1, C=32, K=10, L=32, H=9, W=13, dtype="float64", soft=True),
]


def _wave(shape, freq, phase, dtype):
    n = int(np.prod(shape))
    return torch.sin(torch.arange(n, dtype=torch.float64) * freq + phase).reshape(shape).to(dtype)


def sean_inputs(case, dtype):
    B, C, K, L, H, W = (case[k] for k in ("B", "C", "K", "L", "H", "W"))
    x = _wave((B, C, H, W), 0.173, 0.4, dtype) * 1.7 + 0.3 * _wave((B, C, H, W), 0.0071, 1.0, dtype)
    _, _, dmap, dmask = synth.closed_form_batch(3, B, H, W, 1, K)
    dmap = dmap.to(dtype)
    dmask = dmask.to(dtype)
    if case["soft"]:
        dmask = (0.6 * dmask + 0.4 * (0.5 + 0.5 * _wave(tuple(dmask.shape), 0.37, 0.2, dtype)))
    st = _wave((B, K, L), 0.29, 0.7, dtype)
    return x, dmap, dmask, st


# ---- RegionWiseAvgPooling (sftmd_arch.py:714-733) ----------------------------------------
def pool_inputs():
    cases = {}
    # same size, one empty region
    feat = _wave((2, 8, 6, 7), 0.31, 0.1, torch.float32)
    _, _, _, mask = synth.closed_form_batch(1, 2, 6, 7, 1, 10)
    mask[:, 4] = 0.0
    cases["same_empty"] = (feat, mask)
    # mask larger than feature -> bilinear(align_corners) + >=0.5
    feat = _wave((2, 8, 4, 5), 0.23, 0.5, torch.float32)
    _, _, _, mask = synth.closed_form_batch(2, 2, 16, 20, 1, 10)
    cases["resize4"] = (feat, mask)
    feat = _wave((1, 8, 5, 6), 0.23, 0.5, torch.float32)
    _, _, _, mask = synth.closed_form_batch(2, 1, 17, 21, 1, 10)
    cases["resize_odd"] = (feat, mask)
    return cases


# ---- blocks ---------------------------------------------------------------------------------
def block_inputs():
    B, C, K, L, H, W = 2, 64, 10, 32, 12, 14
    x = torch.relu(_wave((B, C, H, W), 0.173, 0.4, torch.float32) + 0.2)
    _, _, dmap, dmask = synth.closed_form_batch(5, B, H, W, 1, K)
    st = _wave((B, K, L), 0.29, 0.7, torch.float32)
    return x, dmap, dmask, st


# ---- whole net ------------------------------------------------------------------------------
DEPTHNET_CASES = [
    dict(name="x8_nb4", scale=8, which=[0, 1], L=32, nb=4, B=2, H=6, W=8),
    dict(name="x4_nb4", scale=4, which=[0, 1], L=32, nb=4, B=1, H=12, W=16),
    dict(name="x3_nb4", scale=3, which=[0, 1], L=32, nb=4, B=1, H=9, W=12),
    dict(name="x2_nb4", scale=2, which=[0, 1, 2, 3], L=32, nb=4, B=1, H=8, W=12),
    dict(name="x8_nb5_odd", scale=8, which=[0, 1, 2], L=16, nb=5, B=1, H=7, W=9),
]

TRAIN_CASE = dict(name="train", scale=8, which=[0, 1], L=32, nb=4, B=2, H=8, W=12,
                  watch=["conv_output.bias", "depth-residual1.norm1.alpha_gamma", "depth-residual1.norm2.alpha_beta",
                         "depth-residual1.norm1.A_i_j.weight", "encoder.layer4.weight_g", "head.0.bias",
                         "upscale3.0.weight_g", "depth-residual1.conv1.0.bias"])


# BASELINE.json configs[1]'s network on one frame: the reference is run in float64 once (oracle/make_golden.py,
# section 10) and only digests are kept
FULL_X8_CASE = dict(name="full_x8", scale=8, which=list(range(14)), L=256, nb=16, B=1, H=128, W=160)
FULL_DIGEST_STRIDE = 499
OUT_SAMPLE_STRIDE = 193


def depth_mask_cases():
    """RNG-free depth maps for the getDepthMask fixtures: name -> (depth [1,h,w] float32, depthFixedRange, K)."""
    cases = {}
    for i, (h, w, K) in enumerate(((16, 20, 10), (33, 47, 10), (24, 31, 7), (19, 23, 16))):
        d = synth.closed_form_frame(i, h, w, 1)[2]                      # smooth field in (0.01, 10): data range
        cases["smooth%d_K%d" % (i, K)] = (d, False, K)
        cases["smooth%d_K%d_fixed" % (i, K)] = (d / 10.0, True, K)      # inside [0, 1): fixed range
    u = synth.hash_uniform(2 * 19 * 23, "depthmask.u").reshape(2, 19, 23).to(torch.float32)
    cases["noise_data"] = (u[:1] * 9.99 + 0.01, False, 10)
    cases["noise_outside_fixed"] = (u[1:] * 1.4 - 0.2, True, 10)        # values below 0 and above 1: no bin
    cases["constant"] = (torch.full((1, 8, 12), 3.25), False, 10)       # interval 0: every bin empty
    cases["constant_fixed"] = (torch.full((1, 8, 12), 0.25), True, 10)
    edges = ((torch.arange(24 * 40, dtype=torch.float32) % 11) / 10.0).reshape(1, 24, 40)
    cases["edges_data"] = (edges, False, 10)                            # pixels exactly on bin edges and on the max
    cases["edges_fixed"] = (edges, True, 10)
    cases["edges_fixed_K7"] = (edges, True, 7)
    return cases


def make_case_cfg(case):
    return dict(which_ResBlk_depth=list(case["which"]), in_nc=3, out_nc=3, nf=64, nb=case["nb"], scale=case["scale"],
                depth_latent_ch=case["L"], depthRangeNum=10, use_trainable_params=True, norm_gamma=0.1,
                norm_beta=0.1, out_min=0.0, out_max=1.0)


# ---- gradient digests -----------------------------------------------------------------------
DIGEST_FULL_MAX = 2048
DIGEST_STRIDE = 97


def grad_digest(g, stride=DIGEST_STRIDE):
    """Small tensors in full; large ones as [L2 norm, sum, strided sample...]."""
    g = torch.as_tensor(g).detach().reshape(-1).to(torch.float64)
    if g.numel() <= DIGEST_FULL_MAX:
        return g.numpy()
    head = torch.stack([g.norm(), g.sum()])
    return torch.cat([head, g[::stride]]).numpy()


def digest_close(got, want, rtol, atol):
    """Compare a digest of ``got`` (tensor) with the stored digest ``want`` (numpy)."""
    d = grad_digest(got)
    want = np.asarray(want, dtype=np.float64)
    assert d.shape == want.shape, (d.shape, want.shape)
    scale = max(1e-30, float(np.abs(want).max()))
    err = float(np.abs(d - want).max())
    return err <= atol + rtol * scale, err, scale


# ---- SSIM (validation loop) -----------------------------------------------------------------
def ssim_inputs():
    """RNG-free [0,1] image pairs: smooth + hashed texture, the second a perturbed copy of the first."""
    out = {}
    for name, (B, H, W) in (("small", (2, 24, 31)), ("hr", (1, 96, 120))):
        n = B * 3 * H * W
        base = 0.5 + 0.35 * _wave((B, 3, H, W), 0.0131, 0.3, torch.float32) * _wave((B, 3, H, W), 0.00071, 1.1, torch.float32)
        tex = synth.hash_uniform(n, "ssim." + name).reshape(B, 3, H, W).to(torch.float32)
        a = (base + 0.1 * (tex - 0.5)).clamp(0, 1)
        b = (a + 0.08 * (synth.hash_uniform(n, "ssim.noise." + name).reshape(B, 3, H, W).to(torch.float32) - 0.5)).clamp(0, 1)
        out[name] = (a, b)
    out["identical"] = (out["small"][0], out["small"][0].clone())
    return out
