"""Pin the CPU oracle (oracle/depthnet_oracle.py) against vectors produced by the
reference itself (tests/golden/*.npz, written by oracle/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dasr_amd import synth
from oracle import depthnet_oracle as O
from tests.golden_cases import (DEPTHNET_CASES, SEAN_CASES, TRAIN_CASE, block_inputs, digest_close, make_case_cfg,
                                pool_inputs, sean_inputs)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _sd(shapes, dtype=torch.float32, requires_grad=True):
    sd = {k: torch.zeros(s, dtype=dtype) for k, s in shapes.items()}
    synth.closed_form_fill_(sd.items())
    if requires_grad:
        for v in sd.values():
            v.requires_grad_(True)
    return sd


def _sean_shapes(C, K, L, prefix="n"):
    return {
        prefix + ".alpha_beta": (1,), prefix + ".alpha_gamma": (1,),
        prefix + ".A_i_j.weight": (K, K, 1, 1), prefix + ".A_i_j.bias": (K,),
        prefix + ".mlp_gamma_s.weight": (C, L, 3, 3), prefix + ".mlp_gamma_s.bias": (C,),
        prefix + ".mlp_beta_s.weight": (C, L, 3, 3), prefix + ".mlp_beta_s.bias": (C,),
        prefix + ".mlp_mask.0.weight": (2 * C, 1, 3, 3), prefix + ".mlp_mask.0.bias": (2 * C,),
        prefix + ".mlp_gamma_o.weight": (C, 2 * C, 3, 3), prefix + ".mlp_gamma_o.bias": (C,),
        prefix + ".mlp_beta_o.weight": (C, 2 * C, 3, 3), prefix + ".mlp_beta_o.bias": (C,),
    }


@pytest.mark.parametrize("case", SEAN_CASES, ids=[c["name"] for c in SEAN_CASES])
def test_sean_matches_reference(case, golden_dir):
    g = _load(golden_dir, "sean_" + case["name"])
    dtype = getattr(torch, case["dtype"])
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    # the reference module was filled under its own key names (no prefix)
    sd_plain = {k[2:]: torch.zeros(s, dtype=dtype) for k, s in _sean_shapes(case["C"], case["K"], case["L"]).items()}
    synth.closed_form_fill_(sd_plain.items())
    sd = {"n." + k: v.requires_grad_(True) for k, v in sd_plain.items()}
    x, dmap, dmask, st = sean_inputs(case, dtype)
    x.requires_grad_(True)
    st.requires_grad_(True)
    cfg = O.make_cfg(depth_latent_ch=case["L"], depthRangeNum=case["K"])
    out = O.sean(sd, "n", x, dmap, dmask, st, cfg)
    wgt = torch.cos(torch.arange(out.numel(), dtype=dtype) * 0.013).reshape(out.shape)
    (out * wgt).sum().backward()
    assert np.abs(out.detach().numpy() - g["out"]).max() <= tol * max(1.0, np.abs(g["out"]).max())
    assert np.abs(x.grad.numpy() - g["dx"]).max() <= tol * max(1.0, np.abs(g["dx"]).max())
    assert np.abs(st.grad.numpy() - g["dst"]).max() <= tol * max(1.0, np.abs(g["dst"]).max())
    for k, v in sd.items():
        ok, err, scale = digest_close(v.grad, g["g." + k[2:]], rtol=tol * 10, atol=0)
        assert ok, (k, err, scale)


def test_region_pool_matches_reference(golden_dir):
    for name, (feat, mask) in pool_inputs().items():
        g = _load(golden_dir, "pool_" + name)
        feat.requires_grad_(True)
        out = O.region_avg_pool(feat, mask)
        wgt = torch.sin(torch.arange(out.numel(), dtype=out.dtype) * 0.7).reshape(out.shape)
        (out * wgt).sum().backward()
        assert np.abs(out.detach().numpy() - g["out"]).max() <= 1e-6
        assert np.abs(feat.grad.numpy() - g["dfeat"]).max() <= 1e-6
    # empty region -> zero vector (sum/(0+1e-10))
    g = _load(golden_dir, "pool_same_empty")
    assert np.all(g["out"][:, 4] == 0)


@pytest.mark.parametrize("hw", [(16, 20), (17, 21), (18, 23)])
def test_encoder_matches_reference(hw, golden_dir):
    H, W = hw
    g = _load(golden_dir, "encoder_%dx%d" % (H, W))
    shapes = {k: v for k, v in O.param_shapes(O.make_cfg(depth_latent_ch=8)).items() if k.startswith("encoder.")}
    plain = {k[len("encoder."):]: torch.zeros(s) for k, s in shapes.items()}
    synth.closed_form_fill_(plain.items())          # reference filled Encoder under its own (un-prefixed) names
    sd = {"encoder." + k: v for k, v in plain.items()}
    lq, _, _, masks = synth.closed_form_batch(0, 1, H, W, 1)
    feat, vec = O.encoder(sd, lq, masks)
    assert np.abs(feat.numpy() - g["feat"]).max() <= 2e-6
    assert np.abs(vec.numpy() - g["vec"]).max() <= 2e-6
    exp = json.load(open(os.path.join(golden_dir, "encoder_shapes.json")))["%dx%d" % (H, W)]
    assert list(feat.shape) == exp[5] and list(vec.shape) == exp[6]


def test_blocks_match_reference(golden_dir):
    g = _load(golden_dir, "dgb_block")
    shapes = {}
    for n in ("norm1", "norm2"):
        shapes.update({k[2:]: s for k, s in _sean_shapes(64, 10, 32, "b." + n).items()})
    shapes.update({"conv1.0.weight": (64, 64, 3, 3), "conv1.0.bias": (64,),
                   "conv2.0.weight": (64, 64, 3, 3), "conv2.0.bias": (64,)})
    plain = {k: torch.zeros(s) for k, s in shapes.items()}
    synth.closed_form_fill_(plain.items())
    sd = {"b." + k: v.requires_grad_(True) for k, v in plain.items()}
    x, dmap, dmask, st = block_inputs()
    x.requires_grad_(True)
    st.requires_grad_(True)
    cfg = O.make_cfg(depth_latent_ch=32)
    out = O.depth_block(sd, "b", x, dmap, dmask, st, cfg)
    wgt = torch.cos(torch.arange(out.numel(), dtype=out.dtype) * 0.011).reshape(out.shape)
    (out * wgt).sum().backward()
    assert np.abs(out.detach().numpy() - g["out"]).max() <= 5e-6
    assert np.abs(x.grad.numpy() - g["dx"]).max() <= 5e-6 * max(1, np.abs(g["dx"]).max())
    assert np.abs(st.grad.numpy() - g["dst"]).max() <= 5e-6 * max(1, np.abs(g["dst"]).max())
    for k, v in sd.items():
        ok, err, scale = digest_close(v.grad, g["g." + k[2:]], rtol=2e-5, atol=0)
        assert ok, (k, err, scale)

    g = _load(golden_dir, "classic_block")
    plain = {"block.0.bias": torch.zeros(32), "block.0.weight_g": torch.zeros(32, 1, 1, 1),
             "block.0.weight_v": torch.zeros(32, 32, 3, 3), "block.2.bias": torch.zeros(32),
             "block.2.weight_g": torch.zeros(32, 1, 1, 1), "block.2.weight_v": torch.zeros(32, 32, 3, 3)}
    synth.closed_form_fill_(plain.items())
    sd = {"c." + k: v.requires_grad_(True) for k, v in plain.items()}
    xc = block_inputs()[0][:, :32].clone().requires_grad_(True)
    out = O.classic_block(sd, "c", xc)
    wgt = torch.cos(torch.arange(out.numel(), dtype=out.dtype) * 0.011).reshape(out.shape)
    (out * wgt).sum().backward()
    assert np.abs(out.detach().numpy() - g["out"]).max() <= 2e-6
    assert np.abs(xc.grad.numpy() - g["dx"]).max() <= 2e-6 * max(1, np.abs(g["dx"]).max())
    for k, v in sd.items():
        ok, err, scale = digest_close(v.grad, g["g." + k[2:]], rtol=1e-5, atol=0)
        assert ok, (k, err, scale)


@pytest.mark.parametrize("r", [2, 3])
def test_pixel_shuffle_index_map(r, golden_dir):
    g = _load(golden_dir, "pixel_shuffle_r%d" % r)
    shape = tuple(int(v) for v in g["src_shape"])
    src = torch.arange(int(np.prod(shape)), dtype=torch.float32).reshape(shape)
    out = F.pixel_shuffle(src, r).numpy().astype(np.int32)
    assert np.array_equal(out, g["out"])
    # closed form of the index map
    B, Crr, H, W = shape
    C = Crr // (r * r)
    idx = np.zeros((B, C, H * r, W * r), dtype=np.int32)
    for c in range(C):
        for h in range(H * r):
            for w in range(W * r):
                idx[0, c, h, w] = ((c * r * r + (h % r) * r + (w % r)) * H + h // r) * W + w // r
    assert np.array_equal(idx, g["out"])


@pytest.mark.parametrize("case", DEPTHNET_CASES, ids=[c["name"] for c in DEPTHNET_CASES])
def test_depthnet_matches_reference(case, golden_dir):
    g = _load(golden_dir, "depthnet_" + case["name"])
    cfg = make_case_cfg(case)
    sd = _sd(O.param_shapes(cfg))
    lq, gt, dmap, dmask = synth.closed_form_batch(0, case["B"], case["H"], case["W"], cfg["scale"])
    sr = O.depthnet_forward(sd, cfg, lq, dmap, dmask)
    assert tuple(sr.shape) == g["sr"].shape
    assert np.abs(sr.detach().numpy() - g["sr"]).max() <= 5e-6
    w = torch.ones(cfg["depthRangeNum"], requires_grad=True)
    total, l_pix, l_dyn, per = O.total_loss(sr, gt, dmask, w)
    assert abs(l_pix.item() - float(g["l_pix"])) <= 1e-6
    assert abs(l_dyn.item() - float(g["l_dyn"])) <= 1e-5
    assert np.abs(np.array([p.item() for p in per]) - g["per_region"]).max() <= 1e-6
    total.backward()
    nograd = set(g["nograd"].tolist())
    assert np.abs(w.grad.numpy() - g["g.loss_w"]).max() <= 1e-6
    for k, v in sd.items():
        if k in nograd:
            assert v.grad is None, k
            continue
        ok, err, scale = digest_close(v.grad, g["g." + k], rtol=2e-4, atol=1e-9)
        assert ok, (k, err, scale)


@pytest.mark.parametrize("case", DEPTHNET_CASES, ids=[c["name"] for c in DEPTHNET_CASES])
def test_depthnet_fp64_matches_reference_fp64(case, golden_dir):
    """The oracle run in float64 against the reference run in float64 on the same fp32-valued parameters:
    algorithmic identity to 1e-9, with no fp32 rounding (or ReLU flips) in the way."""
    g = _load(golden_dir, "depthnet_" + case["name"])
    cfg = make_case_cfg(case)
    sd = {k: v.detach().double().requires_grad_(True) for k, v in _sd(O.param_shapes(cfg), requires_grad=False).items()}
    lq, gt, dmap, dmask = synth.closed_form_batch(0, case["B"], case["H"], case["W"], cfg["scale"])
    sr = O.depthnet_forward(sd, cfg, lq.double(), dmap.double(), dmask.double())
    assert np.abs(sr.detach().numpy() - g["sr64"]).max() <= 1e-11
    wgt = torch.cos(torch.arange(sr.numel(), dtype=torch.float32) * 0.013).reshape(sr.shape).double()
    (sr * wgt).sum().backward()
    nograd = set(g["nograd"].tolist())
    for k, v in sd.items():
        if k in nograd:
            continue
        ok, err, scale = digest_close(v.grad, g["gl64." + k], rtol=1e-9, atol=1e-12)
        assert ok, (k, err, scale)


def test_state_dict_keys_match_reference(golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    for scale, which, L in ((8, list(range(14)), 256), (4, list(range(14)), 256), (2, list(range(16)), 32)):
        cfg = O.make_cfg(scale=scale, which_ResBlk_depth=which, depth_latent_ch=L)
        shapes = O.param_shapes(cfg)
        ref = keys["x%d" % scale]
        assert [k for k, _ in ref] == list(shapes.keys())
        assert [tuple(s) for _, s in ref] == [tuple(s) for s in shapes.values()]
        assert keys["x%d_nparams" % scale] == sum(int(np.prod(s)) for s in shapes.values())
    assert keys["x8_nparams"] == 14795971


def test_lr_schedule_matches_reference(golden_dir):
    lrs = json.load(open(os.path.join(golden_dir, "lr_schedule.json")))
    for step, lr in lrs.items():
        assert abs(O.cosine_restart_lr(int(step)) - lr) <= 1e-9 + 1e-6 * lr, step


def test_train_step_matches_reference(golden_dir):
    g = _load(golden_dir, "train_step")
    case = TRAIN_CASE
    cfg = make_case_cfg(case)
    sd = _sd(O.param_shapes(cfg))
    w = torch.ones(cfg["depthRangeNum"], requires_grad=True)
    lq, gt, dmap, dmask = synth.closed_form_batch(0, case["B"], case["H"], case["W"], cfg["scale"])
    # parameters that never receive a gradient are skipped by Adam in both implementations
    params = list(sd.values()) + [w]
    optim = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.99), weight_decay=0)
    for step in range(1, 3):
        lr = O.cosine_restart_lr(step)
        assert abs(lr - float(g["lr%d" % step])) <= 1e-12
        for grp in optim.param_groups:
            grp["lr"] = lr
        optim.zero_grad()
        sr = O.depthnet_forward(sd, cfg, lq, dmap, dmask)
        total, l_pix, l_dyn, _ = O.total_loss(sr, gt, dmask, w)
        total.backward()
        optim.step()
        assert abs(l_pix.item() - float(g["l_pix%d" % step])) <= 2e-6
        assert abs(l_dyn.item() - float(g["l_dyn%d" % step])) <= 2e-5
    for k in case["watch"]:
        assert np.abs(sd[k].detach().numpy() - g["p." + k]).max() <= 2e-5, k
    assert np.abs(w.detach().numpy() - g["p.loss_w"]).max() <= 2e-5
