"""Parity checks of the HIP path (dasr_amd) shared by the CPU-emulator tests (-m "not gpu") and the
GPU tests (-m gpu).  ``device`` is 'cpu' when the kernels run in tests/hipemu, 'cuda' on the MI355X."""
import json
import math
import os

import numpy as np
import ctypes
import torch
import torch.nn.functional as F

from dasr_amd import graph, ops, synth
from dasr_amd.depthnet import DepthNet
from dasr_amd.tape import Tape, Var
from oracle import depthnet_oracle as O
from tests.golden_cases import (DEPTHNET_CASES, SEAN_CASES, block_inputs, grad_digest, make_case_cfg, pool_inputs,
                                sean_inputs)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_max(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return (a - b).abs().max().item() / max(1e-30, b.abs().max().item())


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def build_net(case, device):
    cfg = make_case_cfg(case)
    net = DepthNet(which_ResBlk_depth=cfg["which_ResBlk_depth"], in_nc=3, out_nc=3, nf=64, nb=cfg["nb"],
                   scale=cfg["scale"], depth_latent_ch=cfg["depth_latent_ch"], depthRangeNum=10)
    synth.closed_form_fill_(net.state_dict().items())
    return net.to(device), cfg


def digest_global_rel_l2(named_grads, golden, prefix, skip=()):
    num = den = 0.0
    worst = (0.0, None)
    for k, gten in named_grads:
        if any(s in k for s in skip):
            continue
        want = np.asarray(golden[prefix + k], dtype=np.float64)
        got = grad_digest(gten.cpu())
        assert got.shape == want.shape, (k, got.shape, want.shape)
        num += float(((got - want) ** 2).sum())
        den += float((want ** 2).sum())
        r = float(np.abs(got - want).max()) / max(1e-30, float(np.abs(want).max()))
        if r > worst[0]:
            worst = (r, k)
    return math.sqrt(num / max(den, 1e-300)), worst


# conv biases directly in front of an InstanceNorm have a mathematically zero gradient: pure rounding noise
ZERO_GRAD_KEYS = (".conv1.0.bias", ".conv2.0.bias")


# Gates for check_depthnet_case, per case: (linear-functional gradient rel-L2 vs the reference's FLOAT64 run,
# harness-loss gradient rel-L2 vs the reference's fp32 run).  Each is <= 10x the larger of the values measured on
# the MI355X and on the kernel emulator (profiles/r02_gpu_tests.log keeps the printed dicts):
#   x8_nb4      lin64 1.65e-6 (GPU) / 1.62e-6 (emulator), loss 3.4e-7 / 4.2e-7  (an earlier tree flipped one ReLU decision
#               at 6x8 and measured 5.7e-4; the gate that tolerated it is gone with the flip)
#   x4_nb4      lin64 1.2e-6, loss 8.2e-7          x3_nb4      lin64 4.1e-7, loss 1.05e-5
#   x2_nb4      lin64 5.4e-7, loss 1.3e-5          x8_nb5_odd  lin64 1.2e-6, loss 9.3e-7
DEPTHNET_GATES = {"x8_nb4": (1.6e-5, 4e-6), "x4_nb4": (1e-5, 8e-6), "x3_nb4": (4e-6, 1e-4), "x2_nb4": (5e-6, 1.3e-4),
                  "x8_nb5_odd": (1.2e-5, 9e-6)}


def check_depthnet_case(case, device, lin64_tol=None, loss_tol=None):
    g = load("depthnet_" + case["name"])
    if lin64_tol is None:
        lin64_tol = DEPTHNET_GATES[case["name"]][0]
    if loss_tol is None:
        loss_tol = DEPTHNET_GATES[case["name"]][1]
    net, cfg = build_net(case, device)
    lq, gt, dm, mk = [t.to(device) for t in synth.closed_form_batch(0, case["B"], case["H"], case["W"], cfg["scale"])]
    # forward (no grad) vs the reference's output
    with torch.no_grad():
        sr0 = net(lq, dm, mk)
    ref = torch.from_numpy(g["sr"])
    assert tuple(sr0.shape) == tuple(ref.shape)
    err = (sr0.cpu() - ref).abs().max().item()
    assert err <= 1e-4, ("forward", case["name"], err)           # measured 5e-6 .. 3.3e-5
    gt_c = gt.cpu()
    dpsnr = abs(O.psnr_255(sr0.cpu(), gt_c) - O.psnr_255(ref, gt_c))
    assert dpsnr <= 1e-5, ("psnr", dpsnr)            # north_star: within 1e-3 PSNR (fp32); measured <= 3.2e-7
    nograd = set(g["nograd"].tolist())
    # (a) linear functional of the output
    sr = net(lq, dm, mk)
    assert torch.equal(sr.detach(), sr0)             # recorded and unrecorded forwards are the same kernels
    wgt = torch.cos(torch.arange(sr.numel(), dtype=torch.float32) * 0.013).reshape(sr.shape).to(device)
    (sr * wgt).sum().backward()
    named = []
    for k, p in net.named_parameters():
        if k in nograd:
            assert p.grad is None, k
        else:
            assert p.grad is not None, k
            named.append((k, p.grad.detach().clone()))
    # the gate: the reference run in float64 on the same fp32-valued parameters (the fp32 reference run sits
    # 1e-6..3e-3 away from it itself - ReLU / clamp decisions flipping on 1e-7 forward differences - and is reported)
    l64, worst64 = digest_global_rel_l2(named, g, "gl64.", skip=ZERO_GRAD_KEYS)
    l2, worst = digest_global_rel_l2(named, g, "gl.", skip=ZERO_GRAD_KEYS)
    err64 = float(np.abs(sr0.cpu().numpy().astype(np.float64) - g["sr64"]).max())
    assert err64 <= 1e-4, ("forward vs fp64 reference", case["name"], err64)
    assert l64 <= lin64_tol, ("linear-functional grads vs fp64 reference", case["name"], l64, worst64)
    # (b) the harness loss (L1 + dynamic), against the reference's fp32 run (the only one recorded for it)
    net.zero_grad(set_to_none=True)
    sr = net(lq, dm, mk)
    w = torch.ones(cfg["depthRangeNum"], device=device, requires_grad=True)
    total, l_pix, l_dyn, per = O.total_loss(sr, gt, mk, w)
    total.backward()
    assert abs(l_pix.item() - float(g["l_pix"])) <= 2e-6
    assert abs(l_dyn.item() - float(g["l_dyn"])) <= 2e-5
    named = [(k, p.grad) for k, p in net.named_parameters() if k not in nograd]
    l2b, worstb = digest_global_rel_l2(named, g, "g.", skip=ZERO_GRAD_KEYS)
    assert l2b <= loss_tol, ("loss grads", case["name"], l2b, worstb)
    return dict(fwd_err=err, fwd_err64=err64, dpsnr=dpsnr, lin_l2=l2, lin_worst=worst, lin64_l2=l64, lin64_worst=worst64,
                loss_l2=l2b, l_pix_err=abs(l_pix.item() - float(g["l_pix"])), l_dyn_err=abs(l_dyn.item() - float(g["l_dyn"])))


def check_conv_variants(device, seed=0):
    """dasr_conv2d_{fwd,epilogue_bwd,dgrad,wgrad} + weight pack vs torch autograd on random data."""
    gen = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=gen)
    cases = [  # cin, cout, k, stride, pad, transposed, act, ps_r, residual, H, W
        (5, 8, 3, 1, 1, False, 0, 1, False, 7, 9), (5, 8, 3, 2, 1, False, 2, 1, False, 7, 9),
        (5, 8, 3, 2, 1, False, 2, 1, False, 8, 10), (6, 4, 3, 2, 1, True, 2, 1, False, 5, 6),
        (4, 16, 3, 1, 1, False, 2, 2, False, 6, 7), (4, 18, 3, 1, 1, False, 2, 3, False, 5, 4),
        (8, 8, 3, 1, 1, False, 1, 1, True, 9, 9), (4, 3, 9, 1, 4, False, 0, 1, False, 11, 13),
        (1, 16, 3, 1, 1, False, 1, 1, False, 6, 5), (64, 64, 3, 1, 1, False, 0, 1, False, 9, 33),
        (32, 32, 3, 1, 1, False, 1, 1, True, 8, 34), (128, 128, 3, 1, 1, False, 0, 1, False, 5, 33),
        (64, 256, 3, 1, 1, False, 2, 2, False, 6, 9), (32, 128, 3, 1, 1, False, 2, 2, False, 10, 35),
        (32, 3, 9, 1, 4, False, 0, 1, False, 11, 70), (32, 3, 9, 1, 4, False, 0, 1, False, 19, 60),
        (32, 64, 3, 2, 1, False, 2, 1, False, 16, 20), (64, 128, 3, 2, 1, False, 2, 1, False, 9, 11),
        (128, 64, 3, 2, 1, True, 2, 1, False, 5, 6), (64, 64, 3, 2, 1, False, 0, 1, False, 11, 13),
        (8, 32, 3, 1, 1, False, 1, 1, False, 7, 9),
        (64, 64, 3, 1, 1, False, 2, 1, False, 16, 33), (32, 32, 3, 1, 1, False, 1, 1, True, 32, 20),
        (16, 128, 3, 1, 1, False, 2, 2, False, 16, 40),
        # full-width tiles (fast epilogue), 16-row tiles, several N slices, a tile count that is not a multiple of 8
        # (padded workgroups of the XCD-aware order), residual / PixelShuffle through the fast epilogue, the 32x128
        # and 64x32 wgrad blocks
        (128, 128, 3, 1, 1, False, 0, 1, False, 16, 32, 3), (64, 256, 3, 1, 1, False, 2, 2, False, 32, 64, 1),
        (32, 64, 3, 1, 1, False, 1, 1, False, 48, 32, 2), (64, 64, 3, 1, 1, False, 0, 1, True, 16, 64, 2),
        (64, 32, 3, 1, 1, False, 2, 1, False, 16, 32, 2), (32, 128, 3, 1, 1, False, 2, 2, False, 16, 32, 2),
    ]
    worst = 0.0
    for case in cases:
        (cin, cout, k, stride, pad, tr, act, ps, res, H, W) = case[:11]
        B = case[11] if len(case) > 11 else 2
        x = rn(B, cin, H, W).requires_grad_(True)
        wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
        v = rn(*wshape).requires_grad_(True)
        g = (torch.rand(wshape[0], 1, 1, 1, generator=gen) + 0.5).requires_grad_(True)
        b = rn(cout).requires_grad_(True)
        w = torch._weight_norm(v, g, 0)
        y = F.conv_transpose2d(x, w, b, stride=stride, padding=pad) if tr else \
            F.conv2d(x, w, b, stride=stride, padding=pad)
        r = None
        if res:
            r = rn(*y.shape).requires_grad_(True)
            y = y + r
        if act == 1:
            y = F.relu(y)
        if act == 2:
            y = F.leaky_relu(y, 0.2)
        if ps > 1:
            y = F.pixel_shuffle(y, ps)
        wgt = rn(*y.shape)
        (y * wgt).sum().backward()
        tape = Tape()
        xv = Var(nhwc(x.detach()).to(device), True)
        vv, gv, bv = (Var(t.detach().clone().to(device), True) for t in (v, g, b))
        rv = Var(nhwc(r.detach()).to(device), True) if res else None
        wv = graph.pack(tape, vv, gv, tr)
        yv = graph.conv(tape, xv, wv, bv, stride=stride, pad=pad, transposed=tr, act=act, ps_r=ps, residual=rv)
        errs = [rel_max(nchw(yv.data), y.detach())]
        yv.grad = nhwc(wgt).to(device)
        tape.backward()
        errs += [rel_max(nchw(xv.grad), x.grad), rel_max(vv.grad, v.grad), rel_max(gv.grad, g.grad),
                 rel_max(bv.grad, b.grad)]
        if res:
            errs.append(rel_max(nchw(rv.grad), r.grad))
        tol = 2e-5 if cin * k * k < 600 else 2e-4
        assert max(errs) <= tol, ((cin, cout, k, stride, pad, tr, act, ps, res, H, W), errs)
        worst = max(worst, max(errs))
    return worst


def check_pixel_shuffle_bit_exact(device):
    """The fused PixelShuffle store must reproduce nn.PixelShuffle's index map bit for bit."""
    for r in (2, 3):
        g = load("pixel_shuffle_r%d" % r)
        B, Crr, H, W = (int(v) for v in g["src_shape"])
        src = torch.arange(B * Crr * H * W, dtype=torch.float32).reshape(B, Crr, H, W)
        # identity 1x1 convolution (w = I) routes the input through the conv epilogue unchanged
        w = ops.pack_hwio(torch.eye(Crr).reshape(1, 1, Crr, Crr).contiguous().to(device))
        y = ops.conv2d_fwd(nhwc(src).to(device), w, None, None, stride=1, pad=0, act=ops.ACT_NONE, ps_r=r)
        out = nchw(y).cpu().numpy().astype(np.int32)
        assert np.array_equal(out, g["out"]), r
        # and the backward of the epilogue is the inverse permutation
        dy = torch.arange(y.numel(), dtype=torch.float32).reshape(y.shape).to(device)
        dconv = ops.conv2d_epilogue_bwd(dy, y, H, W, Crr, ops.ACT_NONE, r)
        want = F.pixel_unshuffle(nchw(dy.cpu()), r)
        assert torch.equal(nchw(dconv.cpu()), want)


def check_sean_golden(device):
    worst = 0.0
    for case in SEAN_CASES:
        if case["dtype"] != "float32":
            continue
        g = load("sean_" + case["name"])
        B, C, K, L, H, W = (case[k] for k in ("B", "C", "K", "L", "H", "W"))
        shapes = {"alpha_beta": (1,), "alpha_gamma": (1,), "A_i_j.weight": (K, K, 1, 1), "A_i_j.bias": (K,),
                  "mlp_gamma_s.weight": (C, L, 3, 3), "mlp_gamma_s.bias": (C,), "mlp_beta_s.weight": (C, L, 3, 3),
                  "mlp_beta_s.bias": (C,), "mlp_mask.0.weight": (2 * C, 1, 3, 3), "mlp_mask.0.bias": (2 * C,),
                  "mlp_gamma_o.weight": (C, 2 * C, 3, 3), "mlp_gamma_o.bias": (C,),
                  "mlp_beta_o.weight": (C, 2 * C, 3, 3), "mlp_beta_o.bias": (C,)}
        sd = {k: torch.zeros(s) for k, s in shapes.items()}
        synth.closed_form_fill_(sd.items())
        x, dmap, dmask, st = sean_inputs(case, torch.float32)
        # The reference SEAN normalises its input once (normalization.py:56); the fused kernel applies the
        # closed form of TWO instance norms to a raw conv output.  Feed it a tensor whose first instance norm
        # is the identity up to rounding:  x_in = IN(x)  (then IN(IN(x_in)) == the reference's IN(x_in)) ...
        # that identity does not hold exactly, so compare through the oracle instead (pinned to the
        # reference by tests/test_oracle_golden.py) and keep the golden file for the raw SEAN output check.
        xi = F.instance_norm(x, eps=1e-5)
        P = {"n." + k: Var(v.clone().to(device), True, "n." + k) for k, v in sd.items()}
        sdo = {"n." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
        cfg = O.make_cfg(depth_latent_ch=L, depthRangeNum=K)
        xo = x.clone().requires_grad_(True)
        sto = st.clone().requires_grad_(True)
        ref = O.sean(sdo, "n", F.instance_norm(xo, eps=1e-5), dmap, dmask, sto, cfg)
        wgt = torch.cos(torch.arange(ref.numel(), dtype=torch.float32) * 0.013).reshape(ref.shape)
        (ref * wgt).sum().backward()
        # golden pin of the oracle's SEAN on the same inputs (reference output)
        assert np.abs(O.sean(sdo, "n", x, dmap, dmask, st, cfg).detach().numpy() - g["out"]).max() <= 2e-5
        tape = Tape()
        tv = Var(nhwc(x).to(device), True)
        stv = Var(st.clone().to(device), True)
        dv = Var(dmap.reshape(B, H, W, 1).contiguous().to(device))
        ov = graph.sean(tape, P, "n", tv, dv, graph.MaskPack(dmask.contiguous().to(device)), stv, None, False,
                        {"alpha_gamma": None, "alpha_beta": None})
        errs = [rel_max(nchw(ov.data), ref.detach())]
        ov.grad = nhwc(wgt).to(device)
        tape.backward()
        errs += [rel_max(nchw(tv.grad), xo.grad), rel_max(stv.grad, sto.grad)]
        for k in sd:
            errs.append(rel_max(P["n." + k].grad.reshape(sd[k].shape), sdo["n." + k].grad))
        assert max(errs) <= 2e-4, (case["name"], errs)
        worst = max(worst, max(errs))
        del xi
    return worst


def check_region_pool(device):
    for name, (feat, mask) in pool_inputs().items():
        g = load("pool_" + name)
        tape = Tape()
        fv = Var(nhwc(feat).to(device), True)
        out = graph.region_pool(tape, fv, mask.contiguous().to(device))
        assert np.abs(out.data.cpu().numpy() - g["out"]).max() <= 2e-6, name
        wgt = torch.sin(torch.arange(out.data.numel(), dtype=torch.float32) * 0.7).reshape(out.data.shape)
        out.grad = wgt.to(device)
        tape.backward()
        assert np.abs(nchw(fv.grad).cpu().numpy() - g["dfeat"]).max() <= 2e-6, name
    g = load("pool_same_empty")
    assert np.all(g["out"][:, 4] == 0)


def check_blocks(device):
    """Depth_Residual_Block_Mask and Classic_Residual_Block vs the reference vectors."""
    g = load("dgb_block")
    x, dmap, dmask, st = block_inputs()
    B, C, H, W = x.shape
    names = ["norm1." + k for k in _sean_keys()] + ["norm2." + k for k in _sean_keys()] + \
        ["conv1.0.weight", "conv1.0.bias", "conv2.0.weight", "conv2.0.bias"]
    shapes = _dgb_shapes(64, 10, 32)
    sd = {k: torch.zeros(shapes[k]) for k in names}
    synth.closed_form_fill_(sd.items())
    P = {"b." + k: Var(v.to(device), True, "b." + k) for k, v in sd.items()}
    tape = Tape()
    xv = Var(nhwc(x).to(device), True)
    stv = Var(st.clone().to(device), True)
    dv = Var(dmap.reshape(B, H, W, 1).contiguous().to(device))
    out = graph.depth_block(tape, P, "b", xv, dv, graph.MaskPack(dmask.contiguous().to(device)), stv, {})
    assert rel_max(nchw(out.data), g["out"]) <= 1e-5
    wgt = torch.cos(torch.arange(out.data.numel(), dtype=torch.float32) * 0.011).reshape(B, C, H, W)
    out.grad = nhwc(wgt).to(device)
    tape.backward()
    assert rel_max(nchw(xv.grad), g["dx"]) <= 5e-4
    assert rel_max(stv.grad, g["dst"]) <= 5e-4
    named = [(k, P["b." + k].grad) for k in names]
    l2, worst = digest_global_rel_l2(named, g, "g.", skip=ZERO_GRAD_KEYS)
    assert l2 <= 5e-4, (l2, worst)

    g = load("classic_block")
    sd = {"block.0.bias": torch.zeros(32), "block.0.weight_g": torch.zeros(32, 1, 1, 1),
          "block.0.weight_v": torch.zeros(32, 32, 3, 3), "block.2.bias": torch.zeros(32),
          "block.2.weight_g": torch.zeros(32, 1, 1, 1), "block.2.weight_v": torch.zeros(32, 32, 3, 3)}
    synth.closed_form_fill_(sd.items())
    P = {"c." + k: Var(v.to(device), True, "c." + k) for k, v in sd.items()}
    tape = Tape()
    xc = block_inputs()[0][:, :32].contiguous()
    xv = Var(nhwc(xc).to(device), True)
    out = graph.classic_block(tape, P, "c", xv)
    assert rel_max(nchw(out.data), g["out"]) <= 1e-5
    wgt = torch.cos(torch.arange(out.data.numel(), dtype=torch.float32) * 0.011).reshape(xc.shape)
    out.grad = nhwc(wgt).to(device)
    tape.backward()
    assert rel_max(nchw(xv.grad), g["dx"]) <= 1e-4
    l2, worst = digest_global_rel_l2([(k, P["c." + k].grad) for k in sd], g, "g.")
    assert l2 <= 1e-4, (l2, worst)


def _sean_keys():
    return ["alpha_beta", "alpha_gamma", "A_i_j.weight", "A_i_j.bias", "mlp_gamma_s.weight", "mlp_gamma_s.bias",
            "mlp_beta_s.weight", "mlp_beta_s.bias", "mlp_mask.0.weight", "mlp_mask.0.bias", "mlp_gamma_o.weight",
            "mlp_gamma_o.bias", "mlp_beta_o.weight", "mlp_beta_o.bias"]


def _dgb_shapes(C, K, L):
    s = {}
    for n in ("norm1.", "norm2."):
        s.update({n + "alpha_beta": (1,), n + "alpha_gamma": (1,), n + "A_i_j.weight": (K, K, 1, 1),
                  n + "A_i_j.bias": (K,), n + "mlp_gamma_s.weight": (C, L, 3, 3), n + "mlp_gamma_s.bias": (C,),
                  n + "mlp_beta_s.weight": (C, L, 3, 3), n + "mlp_beta_s.bias": (C,),
                  n + "mlp_mask.0.weight": (2 * C, 1, 3, 3), n + "mlp_mask.0.bias": (2 * C,),
                  n + "mlp_gamma_o.weight": (C, 2 * C, 3, 3), n + "mlp_gamma_o.bias": (C,),
                  n + "mlp_beta_o.weight": (C, 2 * C, 3, 3), n + "mlp_beta_o.bias": (C,)})
    s.update({"conv1.0.weight": (C, C, 3, 3), "conv1.0.bias": (C,), "conv2.0.weight": (C, C, 3, 3),
              "conv2.0.bias": (C,)})
    return s


def check_encoder_geometry(device):
    """Encoder (ConvTranspose geometry: 128x160 -> 63x79 -> 32x40) for odd and even sizes."""
    shapes_all = json.load(open(os.path.join(GOLDEN, "encoder_shapes.json")))
    for (H, W) in [(16, 20), (17, 21), (18, 23)]:
        g = load("encoder_%dx%d" % (H, W))
        shp = {k[len("encoder."):]: v for k, v in O.param_shapes(O.make_cfg(depth_latent_ch=8)).items()
               if k.startswith("encoder.")}
        sd = {k: torch.zeros(s) for k, s in shp.items()}
        synth.closed_form_fill_(sd.items())
        P = {"encoder." + k: Var(v.to(device), False) for k, v in sd.items()}
        lq, _, _, masks = synth.closed_form_batch(0, 1, H, W, 1)
        tape = Tape(False)
        x0 = Var(nhwc(lq).to(device))
        L = ops.ACT_LRELU
        e1 = graph.conv(tape, x0, graph._wn(tape, P, "encoder.layer1"), P["encoder.layer1.bias"], act=L)
        e2 = graph.conv(tape, e1, graph._wn(tape, P, "encoder.layer2"), P["encoder.layer2.bias"], stride=2, act=L)
        e3 = graph.conv(tape, e2, graph._wn(tape, P, "encoder.layer3"), P["encoder.layer3.bias"], stride=2, act=L)
        e4 = graph.conv(tape, e3, graph._wn(tape, P, "encoder.layer4", True), P["encoder.layer4.bias"], stride=2,
                        transposed=True, act=L)
        e5 = graph.conv(tape, e4, graph._wn(tape, P, "encoder.layer5"), P["encoder.layer5.bias"], stride=2)
        st = graph.region_pool(tape, e5, masks.contiguous().to(device))
        exp = shapes_all["%dx%d" % (H, W)]
        for got, want in zip((e1, e2, e3, e4, e5), exp[:5]):
            b, hh, ww, c = got.data.shape
            assert [b, c, hh, ww] == want, (got.data.shape, want)
        assert np.abs(nchw(e1.data).cpu().numpy() - g["feat"]).max() <= 2e-6
        assert np.abs(nchw(e5.data).cpu().numpy() - g["l5"]).max() <= 5e-6
        assert np.abs(st.data.cpu().numpy() - g["vec"]).max() <= 5e-6


def check_fused_loss(device):
    """dasr_loss_sums / dasr_loss_bwd vs the oracle's L1 + dynamic loss (values and d/dsr, d/dw)."""
    from dasr_amd import harness
    for (B, h, w, s) in [(2, 6, 7, 8), (1, 5, 9, 2), (2, 4, 5, 3)]:
        lq, gt, dm, mk = synth.seeded_batch(7, B, h, w, s)
        gen = torch.Generator().manual_seed(3)
        sr = (gt + 0.6 * torch.randn(gt.shape, generator=gen)).clamp(-1, 2)
        sr[0, 0, 0, :3] += 3.0                        # |d| > 1: the linear branch of smooth-L1
        sr_o = sr.clone().requires_grad_(True)
        w_o = (1.0 + 0.1 * torch.arange(10, dtype=torch.float32)).requires_grad_(True)
        total_o, l_pix_o, l_dyn_o, per_o = O.total_loss(sr_o, gt, mk, w_o)
        total_o.backward()
        region, flag = ops.mask_compress(mk.contiguous().to(device))
        assert int(flag.item()) == 0
        sr_d = sr.clone().to(device).requires_grad_(True)
        w_d = w_o.detach().clone().to(device).requires_grad_(True)
        sums = harness._RegionSums.apply(sr_d, gt.to(device), region, 10)
        num, den, l1 = sums[:10], sums[10:20].detach(), sums[20]
        l_pix = l1 / sr.numel()
        l_dyn = (F.softmax(w_d, 0) * (num / den)).sum() * 10.0
        (l_pix + l_dyn).backward()
        assert abs(l_pix.item() - l_pix_o.item()) <= 2e-6 * max(1, abs(l_pix_o.item()))
        assert abs(l_dyn.item() - l_dyn_o.item()) <= 2e-5 * max(1, abs(l_dyn_o.item()))
        assert rel_max(sr_d.grad, sr_o.grad) <= 2e-5
        assert rel_max(w_d.grad, w_o.grad) <= 2e-5


def check_dgrad_act(device):
    """dasr_conv2d_dgrad_act == dgrad, times act'(x), through the inverse PixelShuffle map (3x3 MFMA and 9x9 paths)."""
    gen = torch.Generator().manual_seed(5)
    for (cin, cout, k, pad, H, W, r, act) in [(32, 64, 3, 1, 8, 34, 1, ops.ACT_RELU), (32, 32, 3, 1, 16, 36, 2, ops.ACT_LRELU),
                                              (64, 32, 3, 1, 6, 33, 3, ops.ACT_LRELU), (32, 3, 9, 4, 10, 66, 2, ops.ACT_LRELU),
                                              (32, 3, 9, 4, 12, 30, 1, ops.ACT_RELU)]:
        B = 2
        x_act = torch.randn(B, H, W, cin, generator=gen).to(device)
        w = ops.pack_hwio((torch.randn(k, k, cin, cout, generator=gen) * 0.1).to(device))
        dconv = torch.randn(B, H, W, cout, generator=gen).to(device)
        assert ops.conv2d_dgrad_act_supported(x_act.shape, w, dconv.shape, 1, pad, False, r)
        got = ops.conv2d_dgrad_act(dconv, w, x_act, act, r, 1, pad, False)
        ref = ops.conv2d_dgrad(dconv, w, x_act.shape, 1, pad, False)
        slope = 0.0 if act == ops.ACT_RELU else 0.2
        ref = ref * torch.where(x_act > 0, torch.ones_like(x_act), torch.full_like(x_act, slope))
        ref = nhwc(F.pixel_unshuffle(nchw(ref.cpu()), r)) if r > 1 else ref.cpu()
        assert got.shape == ref.shape
        assert rel_max(got, ref) <= 1e-6, (cin, cout, k, r)


# ------------------------------------------------------------------------------------------------------------
# edge cases of the module boundary
# ------------------------------------------------------------------------------------------------------------
def _oracle_sd(net):
    return {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()}


def _compare_with_oracle(net, cfg, lq, dm, mk, device, fwd_tol=2e-4, grad_tol=2e-5):
    """Forward max error and gradient relative L2 against the CPU oracle run in this process.  Gates at ~10x the values
    measured on the MI355X (forward 2.3e-5 .. 2.6e-5, gradients 1.7e-6 .. 1.9e-6 for the cases that use it)."""
    sd = _oracle_sd(net)
    sr = net(lq.to(device), dm.to(device), mk.to(device))
    ref = O.depthnet_forward(sd, cfg, lq, dm, mk)
    err = (sr.detach().cpu() - ref.detach()).abs().max().item()
    assert err <= fwd_tol, err
    wgt = torch.cos(torch.arange(sr.numel(), dtype=torch.float32) * 0.013).reshape(sr.shape)
    (sr * wgt.to(device)).sum().backward()
    (ref * wgt).sum().backward()
    num = den = 0.0
    for k, p in net.named_parameters():
        if sd[k].grad is None:
            assert p.grad is None, k
            continue
        assert p.grad is not None, k
        if any(s in k for s in ZERO_GRAD_KEYS):
            continue
        num += (p.grad.detach().cpu().double() - sd[k].grad.double()).pow(2).sum().item()
        den += sd[k].grad.double().pow(2).sum().item()
    rel = math.sqrt(num / max(den, 1e-300))
    assert rel <= grad_tol, rel
    return err, rel


def check_soft_masks_whole_net(device):
    """Arbitrary float masks (not one-hot): the general kernels carry the whole net."""
    case = dict(scale=2, which=[0, 1, 2, 3], L=16, nb=4, B=1, H=8, W=12)
    net, cfg = build_net(case, device)
    lq, gt, dm, mk = synth.closed_form_batch(2, 1, 8, 12, 2)
    soft = 0.7 * mk + 0.3 * synth.hash_uniform(mk.numel(), "soft").reshape(mk.shape).float()
    return _compare_with_oracle(net, cfg, lq, dm, soft, device)


def check_other_region_counts(device):
    """depthRangeNum != 10 (the reference takes it from the yml's depthMaskNum): K = 7 and K = 16 (the kernels'
    SEAN_MAXK), one-hot masks, nf-64 DGBs, against the oracle - forward and all gradients."""
    out = {}
    for K, L in ((7, 16), (16, 32)):
        cfg = O.make_cfg(which_ResBlk_depth=[0, 1], nb=4, scale=2, depth_latent_ch=L, depthRangeNum=K)
        net = DepthNet(which_ResBlk_depth=[0, 1], nb=4, scale=2, depth_latent_ch=L, depthRangeNum=K)
        synth.closed_form_fill_(net.state_dict().items())
        net = net.to(device)
        lq, _, dm, mk = synth.closed_form_batch(1, 2, 8, 12, 2, K)
        assert mk.shape[1] == K
        out["K%d" % K] = _compare_with_oracle(net, cfg, lq, dm, mk, device)
    return out


def check_constant_alpha_and_mask_resize(device):
    """use_trainable_params=False (constant blend weights) and a depth map / masks at HALF the LR resolution:
    F.interpolate(nearest) inside SEAN (normalization.py:58-59) and the bilinear+threshold resize of the region
    pooling (sftmd_arch.py:715-718)."""
    cfg = O.make_cfg(which_ResBlk_depth=[0, 1], nb=4, scale=4, depth_latent_ch=16, use_trainable_params=False,
                     norm_gamma=0.3, norm_beta=0.6)
    net = DepthNet(which_ResBlk_depth=[0, 1], nb=4, scale=4, depth_latent_ch=16, use_trainable_params=False,
                   norm_gamma=0.3, norm_beta=0.6)
    synth.closed_form_fill_(net.state_dict().items())
    net = net.to(device)
    assert not any("alpha" in k for k in net.state_dict())
    lq, _, _, _ = synth.closed_form_batch(0, 2, 12, 16, 4)
    _, _, dm, mk = synth.closed_form_batch(0, 2, 6, 8, 4)          # half-resolution depth inputs
    return _compare_with_oracle(net, cfg, lq, dm, mk, device)


def check_batch_independence_and_determinism(device):
    """Instance norm is per sample: a frame's output must not depend on its batch mates; repeated runs are bitwise
    identical in the forward (no atomics on the forward path).  Bitwise batch independence holds for the exact-fp32 kernels.
    The split convolutions (graph.SPLIT_BF16) are chosen by launch size and, in the fp16 x 2 scheme, scale every tensor by a
    power of two taken from the WHOLE batch's max |.|: a frame's low-order bits then depend on its batch mates (as they do
    when a library picks its algorithm by batch size) - bounded here at a few fp32 roundings of a [0, 1] image."""
    case = dict(scale=8, which=[0, 1], L=16, nb=4, B=3, H=8, W=12)
    net, cfg = build_net(case, device)
    lq, gt, dm, mk = [t.to(device) for t in synth.closed_form_batch(0, 3, 8, 12, 8)]
    old = graph.SPLIT_BF16, graph.SPLIT_MIN_PIXELS
    try:
        for split, min_px in ((False, old[1]), (True, old[1]), (True, 0)):
            graph.SPLIT_BF16, graph.SPLIT_MIN_PIXELS = split, min_px
            with torch.no_grad():
                full = net(lq, dm, mk)
                again = net(lq, dm, mk)
                solo = net(lq[1:2].contiguous(), dm[1:2].contiguous(), mk[1:2].contiguous())
            assert torch.equal(full, again)
            if not split:
                assert torch.equal(full[1:2], solo)
            else:
                assert (full[1:2] - solo).abs().max().item() <= 2e-5, (split, min_px, (full[1:2] - solo).abs().max().item())
            assert full.min().item() >= 0.0 and full.max().item() <= 1.0          # torch.clamp(out, 0, 1)
            assert tuple(full.shape) == (3, 3, 64, 96)
    finally:
        graph.SPLIT_BF16, graph.SPLIT_MIN_PIXELS = old


def check_state_dict_roundtrip(device):
    """state_dict keys == the reference's (dumped in tests/golden/state_dict_keys.json); strict load works."""
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    for scale, which, L in ((8, list(range(14)), 256), (4, list(range(14)), 256), (2, list(range(16)), 32)):
        net = DepthNet(which_ResBlk_depth=which, nb=16, scale=scale, depth_latent_ch=L)
        ref = keys["x%d" % scale]
        sd = net.state_dict()
        assert [k for k, _ in ref] == list(sd.keys())
        assert [tuple(s) for _, s in ref] == [tuple(v.shape) for v in sd.values()]
        assert sum(p.numel() for p in net.parameters()) == keys["x%d_nparams" % scale]
        other = DepthNet(which_ResBlk_depth=which, nb=16, scale=scale, depth_latent_ch=L)
        other.load_state_dict(sd, strict=True)
    for bad in (dict(norm_type="instance"), dict(ablate_depth_block=True), dict(ablate_depth_matrix=True)):
        try:
            DepthNet(which_ResBlk_depth=[0], nb=4, scale=2, **bad)
            raise AssertionError("expected NotImplementedError")
        except NotImplementedError:
            pass


def check_reference_assertion(device):
    """The reference asserts len_latent == st.size(2) and st.size(1) == depthMask.size(1) (normalization.py:54)."""
    case = dict(scale=2, which=[0, 1, 2, 3], L=16, nb=4, B=1, H=8, W=12)
    net, cfg = build_net(case, device)
    lq, gt, dm, mk = [t.to(device) for t in synth.closed_form_batch(0, 1, 8, 12, 2, num_masks=7)]
    try:
        net(lq, dm, mk)          # 7 mask planes against a net built for 10 regions
        raise RuntimeError("expected an AssertionError / shape error")
    except (AssertionError, RuntimeError) as e:
        assert "expected an" not in str(e)


def check_full_size_x8(device):
    """BASELINE.json's full shapes (x8 net, nb=16, L=256, 128x160 LR -> 1024x1280) on one frame.  Forward against the
    CPU oracle (max error, PSNR agreement: north_star 1e-3 dB) and against the reference's float64 run (sampled);
    gradients of a linear functional against the digests of the reference's FLOAT64 run
    (tests/golden/depthnet_full_x8_f64.npz, oracle/make_golden.py section 10)."""
    from tests.golden_cases import FULL_DIGEST_STRIDE, FULL_X8_CASE, OUT_SAMPLE_STRIDE
    g = load("depthnet_full_x8_f64")
    case = FULL_X8_CASE
    net, cfg = build_net(case, device)
    lq, gt, dm, mk = synth.closed_form_batch(0, 1, case["H"], case["W"], 8)
    sd = _oracle_sd(net)
    sr = net(lq.to(device), dm.to(device), mk.to(device))
    with torch.no_grad():
        ref = O.depthnet_forward(sd, cfg, lq, dm, mk)
    assert tuple(sr.shape) == (1, 3, 1024, 1280)
    out = sr.detach().cpu()
    err = (out - ref).abs().max().item()
    dpsnr = abs(O.psnr_255(out, gt) - O.psnr_255(ref, gt))
    psnr_vs_ref = O.psnr_255(out, ref)
    err64 = float(np.abs(out.reshape(-1)[::OUT_SAMPLE_STRIDE].double().numpy() - g["sr64_sample"]).max())
    dpsnr64 = abs(O.psnr_255(out, gt) - float(g["psnr64_gt"]))
    assert err <= 5e-4, err                       # measured 2.2e-4 (round 1)
    assert err64 <= 5e-4, err64
    assert dpsnr <= 1e-3 and dpsnr64 <= 1e-3, (dpsnr, dpsnr64)
    assert psnr_vs_ref >= 100.0, psnr_vs_ref
    wgt = torch.cos(torch.arange(sr.numel(), dtype=torch.float32) * 0.013).reshape(sr.shape)
    (sr * wgt.to(device)).sum().backward()
    nograd = set(g["nograd"].tolist())
    num = den = 0.0
    for k, p in net.named_parameters():
        if k in nograd:
            assert p.grad is None, k
            continue
        assert p.grad is not None, k
        if any(s in k for s in ZERO_GRAD_KEYS):
            continue
        got = grad_digest(p.grad.cpu(), FULL_DIGEST_STRIDE)
        want = np.asarray(g["gl64." + k], dtype=np.float64)
        assert got.shape == want.shape, k
        num += float(((got - want) ** 2).sum())
        den += float((want ** 2).sum())
    rel = math.sqrt(num / den)
    # At this depth (13 DGBs, 3.9 M clamped outputs) fp32 gradients are only defined to ~1 %: the reference's own
    # fp32 run sits 1.47e-2 from its fp64 run, this implementation 1.05e-2 (profiles/r01_full_size_fp64_diagnostic.txt).
    assert rel <= 1.5e-2, rel
    return dict(max_err=err, max_err64_sampled=err64, dpsnr=dpsnr, dpsnr64=dpsnr64, psnr_vs_ref=psnr_vs_ref,
                grad_rel_l2_vs_fp64=rel)


def check_depth_prep(device):
    """dasr_depth_to_masks (prep.hip) against the host restatement of getDepthMask (synth.depth_to_masks,
    LQGTker_Depth_dataset.py:204-225): planes and region bytes bit-exact, incl. the pixel(s) equal to the maximum
    (no bin), a constant map (every bin empty), a map with values outside [0,1) in fixed-range mode; the region
    bytes equal dasr_mask_compress of the planes; a net fed with prepared masks returns the same bits."""
    from dasr_amd import prep
    K = 10
    maps = []
    for i, (h, w) in enumerate(((16, 20), (33, 47), (128, 160))):
        _, _, dm, _ = synth.seeded_batch(3 * i, 2, h, w, 1, K)
        maps.append(dm)
    g = torch.Generator().manual_seed(5)
    maps.append(torch.rand(2, 1, 19, 23, generator=g) * 1.4 - 0.2)                  # outside [0,1) too
    maps.append(torch.full((1, 1, 8, 12), 3.25))                                     # constant: all bins empty
    maps.append((torch.arange(2 * 24 * 40, dtype=torch.float32) % 11).reshape(2, 1, 24, 40) / 10.0)   # exact edges
    n_nobin = 0
    for dm in maps:
        for fixed in (False, True):
            want = torch.stack([synth.depth_to_masks(dm[b], K, fixed) for b in range(dm.shape[0])])
            got = prep.depth_to_masks(dm.to(device), K, fixed)
            assert got.shape == want.shape and torch.equal(got.cpu(), want), ("planes", tuple(dm.shape), fixed)
            region = got._dasr_region.cpu()
            idx = torch.where(want.sum(1) > 0, want.argmax(1), torch.full_like(want.argmax(1), K)).to(torch.uint8)
            assert torch.equal(region, idx), ("region", tuple(dm.shape), fixed)
            r2, flag = ops.mask_compress(got)
            assert int(flag.item()) == 0 and torch.equal(r2.cpu(), region)
            assert torch.equal(prep.depth_to_region(dm.to(device), K, fixed).cpu(), region)
            n_nobin += int((region == K).sum())
    assert n_nobin > 0
    # the net: prepared masks (region bytes attached) vs plain float planes
    case = dict(name="prep", scale=4, which=[0, 1], L=32, nb=4, B=2, H=12, W=16)
    net, cfg = build_net(case, device)
    lq, gt, dm, mk = [t.to(device) for t in synth.seeded_batch(0, 2, 12, 16, 4)]
    mk2 = prep.depth_to_masks(dm, K)
    assert torch.equal(mk2, mk)
    with torch.no_grad():
        a = net(lq, dm, mk)
        b = net(lq, dm, mk2)
    assert torch.equal(a, b)
    return dict(no_bin_pixels=n_nobin)


def check_validation_and_folding(device):
    """validate.validate (train.py:219-262) on two frames: the eval/no_grad forward equals the training-mode
    forward bit for bit, folded weights are reused (same tensors on the second call) and refreshed when a
    parameter changes; PSNR/SSIM equal the metrics computed from the oracle's output."""
    from dasr_amd import validate
    case = dict(name="val", scale=4, which=[0, 1], L=32, nb=4, B=1, H=12, W=16)
    net, cfg = build_net(case, device)
    frames = []
    for i in range(2):
        lq, gt, dm, mk = [t.to(device) for t in synth.seeded_batch(i, 1, 12, 16, 4)]
        frames.append((lq, gt, dm, mk))
    lq, gt, dm, mk = frames[0]
    with torch.no_grad():
        ref_out = net(lq, dm, mk)                       # training mode, no folding
    assert not hasattr(net, "_fold_cache") or not net._fold_cache
    out1 = validate.test(net, lq, dm, mk)
    assert net.training and torch.equal(out1, ref_out)
    n_folded = len(net._fold_cache)
    assert n_folded > 10
    ids = {k: id(v[1]) for k, v in net._fold_cache.items()}
    out2 = validate.test(net, lq, dm, mk)
    assert torch.equal(out2, ref_out) and {k: id(v[1]) for k, v in net._fold_cache.items()} == ids
    # a parameter update invalidates exactly the folded kernels built from it
    with torch.no_grad():
        net.get_parameter("conv_output.weight").mul_(1.5)
        net.get_parameter("head.0.weight_g").mul_(0.5)
    out3 = validate.test(net, lq, dm, mk)
    with torch.no_grad():
        fresh = net(lq, dm, mk)
    assert torch.equal(out3, fresh) and not torch.equal(out3, ref_out)
    changed = [k for k, v in net._fold_cache.items() if id(v[1]) != ids[k]]
    assert sorted(k[1] for k in changed) == ["conv_output.weight", "head.0.weight_v"], changed
    # metrics against the oracle's image
    psnr, ssim_v, n = validate.validate(net, frames, cfg["scale"])
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ps = ss = 0.0
    for lq, gt, dm, mk in frames:
        o = O.depthnet_forward(sd, cfg, lq.cpu(), dm.cpu(), mk.cpu())
        ss += float(O.ssim_ref(o, gt.cpu()))
        a, b = validate.tensor2img(o[0]) / 255.0, validate.tensor2img(gt[0]) / 255.0
        ps += validate.calculate_psnr(a[4:-4, 4:-4] * 255, b[4:-4, 4:-4] * 255)
    assert n == 2 and abs(psnr - ps / 2) <= 1e-2 and abs(ssim_v - ss / 2) <= 2e-5, (psnr, ps / 2, ssim_v, ss / 2)
    return dict(psnr=psnr, ssim=ssim_v, folded=n_folded)


def check_ssim_kernel(device):
    """dasr_ssim (one HIP pass: windowed moments + SSIM ratio + mean) against the REFERENCE's pytorch_ssim values
    (tests/golden/ssim.npz, 2e-6) and against the CPU restatement on frames with ragged tiles / more samples (2e-6)."""
    from dasr_amd import validate
    from tests.golden_cases import ssim_inputs
    g = np.load(os.path.join(GOLDEN, "ssim.npz"))
    worst = 0.0
    for name, (a, b) in ssim_inputs().items():
        ad, bd = a.to(device), b.to(device)
        m = float(validate.ssim(ad, bd))
        per = validate.ssim(ad, bd, size_average=False).cpu().numpy()
        worst = max(worst, abs(m - float(g[name + ".mean"])), float(np.abs(per - g[name + ".per_image"]).max()))
        assert abs(m - float(g[name + ".mean"])) <= 2e-6 and np.abs(per - g[name + ".per_image"]).max() <= 2e-6, name
    gen = torch.Generator().manual_seed(9)
    for (B, C, H, W) in [(3, 3, 45, 70), (1, 1, 7, 9), (2, 3, 64, 96)]:
        a = torch.rand(B, C, H, W, generator=gen)
        b = (a + 0.1 * torch.randn(B, C, H, W, generator=gen)).clamp(0, 1)
        got = validate.ssim(a.to(device), b.to(device), size_average=False).cpu()
        want = O.ssim_ref(a, b, size_average=False)
        d = (got - want).abs().max().item()
        worst = max(worst, d)
        assert d <= 2e-6, ((B, C, H, W), d)
    return dict(worst=worst)


def check_checkpoint_interop(device, tmpdir):
    """base_model.save_network / load_network / save_training_state / resume_training (base_model.py:77-119):
    reference key names, a DataParallel-style 'module.' checkpoint, strict failures, and a resumed run that
    continues where the original left off (bit-identically on the deterministic emulator)."""
    from dasr_amd import harness
    case = dict(name="ckpt", scale=2, which=[0, 1], L=16, nb=4, B=1, H=8, W=12)
    net, cfg = build_net(case, device)
    path = os.path.join(tmpdir, "100_G.pth")
    harness.save_network(net, path)
    loaded = torch.load(path)
    assert list(loaded.keys()) == list(net.state_dict().keys()) and all(v.device.type == "cpu" for v in loaded.values())
    torch.save({"module." + k: v for k, v in loaded.items()}, os.path.join(tmpdir, "dp_G.pth"))
    net2, _ = build_net(case, device)
    with torch.no_grad():
        for p in net2.parameters():
            p.add_(1.0)
    harness.load_network(os.path.join(tmpdir, "dp_G.pth"), net2, strict=True)
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), net2.state_dict().values()))
    bad = dict(loaded)
    bad.pop("conv_output.bias")
    torch.save(bad, os.path.join(tmpdir, "bad_G.pth"))
    try:
        harness.load_network(os.path.join(tmpdir, "bad_G.pth"), net2, strict=True)
        raise AssertionError("strict load of an incomplete checkpoint must fail")
    except RuntimeError:
        pass
    # resume: 2 steps + save + 1 step  ==  load + 1 step
    batch = [t.to(device) for t in synth.seeded_batch(0, 1, 8, 12, 2)]
    tr = harness.Trainer(net)
    for _ in range(2):
        tr.optimize_parameters(*batch)
    harness.save_network(net, os.path.join(tmpdir, "2_G.pth"))
    tr.save_training_state(os.path.join(tmpdir, "2.state"), epoch=3)
    tr.optimize_parameters(*batch)
    want = {k: v.detach().clone() for k, v in net.state_dict().items()}
    want_w = tr.dynamic_loss.trainable_weight.detach().clone()
    net3, _ = build_net(case, device)
    harness.load_network(os.path.join(tmpdir, "2_G.pth"), net3)
    tr3 = harness.Trainer(net3)
    epoch, it = tr3.resume_training(os.path.join(tmpdir, "2.state"))
    assert (epoch, it) == (3, 2) and tr3.step_count == 2
    tr3.optimize_parameters(*batch)
    # bit-identical on the (deterministic) emulator; on the GPU the bias / small-wgrad sums meet in float atomics, so
    # two runs of the same step differ in the last bits of the gradient
    exact = device == "cpu"
    for k, v in net3.state_dict().items():
        if exact:
            assert torch.equal(v, want[k]), k
        elif not any(z in k for z in ZERO_GRAD_KEYS):   # their gradient is rounding noise, which Adam turns into +-lr steps
            assert torch.allclose(v, want[k], rtol=1e-4, atol=2e-6), (k, (v - want[k]).abs().max().item())
    assert torch.allclose(tr3.dynamic_loss.trainable_weight.detach(), want_w, rtol=0 if exact else 1e-5, atol=0 if exact else 1e-6)
    state = torch.load(os.path.join(tmpdir, "2.state"))
    assert set(("epoch", "iter", "schedulers", "optimizers")) <= set(state.keys())
    return dict(keys=len(loaded), resumed_iter=it)


def check_x4_config_shape(device):
    """BASELINE.json configs[2] shape in fp32 (Kvasir x4, LR 256x320, DGBs 0..13, L=256; two frames): same
    size-independent properties as check_large_frame_x2, plus batch independence (frame 1 alone == frame 1 of the pair to a
    few fp32 roundings of the [0, 1] image)."""
    from dasr_amd import harness
    net = DepthNet(which_ResBlk_depth=list(range(14)), in_nc=3, out_nc=3, nf=64, nb=16, scale=4, depth_latent_ch=256,
                   depthRangeNum=10)
    synth.closed_form_fill_(net.state_dict().items())
    net = net.to(device)
    lq, gt, dm, mk = [t.to(device) for t in synth.seeded_batch(0, 2, 256, 320, 4)]
    with torch.no_grad():
        fast = net(lq, dm, mk)
        single = net(lq[1:], dm[1:], mk[1:])
        graph.FORCE_GENERAL_SEAN = True
        try:
            general = net(lq, dm, mk)
        finally:
            graph.FORCE_GENERAL_SEAN = False
    assert tuple(fast.shape) == (2, 3, 1024, 1280)
    # (the fp16 x 2 split convolutions scale each tensor by a power of two taken from the whole batch's max |.|: a frame's
    # low-order bits depend on its batch mates - check_batch_independence_and_determinism)
    bd = (fast[1:] - single).abs().max().item()
    assert bd <= 1e-4, bd          # measured 3.1e-5 through 14 DGBs at this size (the forward's own gate against the reference is 1e-4)
    diff = (fast - general).abs().max().item()
    psnr = O.psnr_255(fast.cpu(), general.cpu())
    assert diff <= 5e-4 and psnr > 90.0, (diff, psnr)
    del general, single
    tr = harness.Trainer(net)
    log = tr.optimize_parameters(lq, gt, dm, mk)
    torch.cuda.synchronize()
    assert math.isfinite(float(log["l_all"]))
    n_grads = sum(1 for p in net.parameters() if p.grad is not None and bool(torch.isfinite(p.grad).all()))
    assert n_grads > 300
    return dict(fast_vs_general=diff, psnr=psnr, loss=float(log["l_all"]), grads=n_grads)


def check_large_frame_x2(device, H=1080, W=1920):
    """BASELINE.json configs[4] shape (EndoScene x2, one 1080p-class LR frame per GPU, L=256, DGBs 0..15): the
    oracle cannot finish this size in test time, so the checks are size-independent properties: (i) the one-hot
    gather kernels and the general soft-mask kernels - two independent implementations - agree on the whole net's
    output, (ii) one training step runs with finite loss and gradients for every trained parameter, (iii) the
    device-side mask preparation equals the host rule on the full map."""
    from dasr_amd import harness, prep
    net = DepthNet(which_ResBlk_depth=list(range(16)), in_nc=3, out_nc=3, nf=64, nb=16, scale=2, depth_latent_ch=256,
                   depthRangeNum=10)
    synth.closed_form_fill_(net.state_dict().items())
    net = net.to(device)
    lq, gt, dm, mk = [t.to(device) for t in synth.seeded_batch(0, 1, H, W, 2)]
    mk2 = prep.depth_to_masks(dm, 10)
    assert torch.equal(mk2, mk)
    with torch.no_grad():
        fast = net(lq, dm, mk2)
        graph.FORCE_GENERAL_SEAN = True
        try:
            general = net(lq, dm, mk)
        finally:
            graph.FORCE_GENERAL_SEAN = False
    assert tuple(fast.shape) == (1, 3, 2 * H, 2 * W)
    diff = (fast - general).abs().max().item()
    psnr = O.psnr_255(fast.cpu(), general.cpu())
    assert diff <= 5e-4 and psnr > 90.0, (diff, psnr)
    del general
    tr = harness.Trainer(net)
    log = tr.optimize_parameters(lq, gt, dm, mk2)
    torch.cuda.synchronize()
    assert math.isfinite(float(log["l_all"])) and float(log["l_pix"]) > 0
    n_grads = 0
    for k, p in net.named_parameters():
        if p.grad is not None:
            assert bool(torch.isfinite(p.grad).all()), k
            n_grads += 1
    assert n_grads > 300
    return dict(fast_vs_general=diff, psnr=psnr, loss=float(log["l_all"]), grads=n_grads)


def check_conv_fwd_stats(device):
    """dasr_conv2d_fwd_stats: the convolution output is bit-identical to dasr_conv2d_fwd and the statistics match
    dasr_instnorm_stats of that output (and torch's biased variance), on the fused path (W % 32 == 0: 8- and 16-row
    tiles, ragged last tile row, 32 / 64 / 128 channels) and on the fallback (W % 32 != 0, Cin not a multiple of 16)."""
    worst = 0.0
    for (B, H, W, Cin, Cout) in ((2, 16, 32, 64, 64), (1, 21, 64, 64, 64), (2, 40, 32, 32, 32), (1, 32, 96, 128, 128),
                                 (2, 9, 20, 64, 64), (1, 16, 32, 8, 32)):
        g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin)
        x = (torch.rand(B, H, W, Cin, generator=g) - 0.4).to(device)
        wt = ops.pack_hwio(((torch.rand(3, 3, Cin, Cout, generator=g) - 0.5) * 0.2).to(device))
        bias = (torch.rand(Cout, generator=g) - 0.5).to(device)
        y0 = ops.conv2d_fwd(x, wt, bias)
        m0, v0 = ops.instnorm_stats(y0)
        y1, m1, v1 = ops.conv2d_fwd_stats(x, wt, bias)
        assert torch.equal(y0, y1), (B, H, W, Cin, Cout)
        yt = y0.double().reshape(B, H * W, Cout).cpu()
        mt, vt = yt.mean(1), yt.var(1, unbiased=False)
        for got_m, got_v in ((m0, v0), (m1, v1)):
            em = ((got_m.cpu().double() - mt).abs() / (mt.abs() + vt.sqrt())).max().item()
            ev = ((got_v.cpu().double() - vt).abs() / vt).max().item()
            assert em <= 2e-6 and ev <= 2e-5, (B, H, W, Cin, Cout, em, ev)
            worst = max(worst, em, ev)
    return dict(worst_rel=worst)


def check_train_step(device):
    """harness.Trainer.optimize_parameters x2 with the HIP net against what the reference's own pieces recorded for the
    same two steps (tests/golden/train_step.npz: scheduler stepped first, train.py:194; L1 + dynamic loss + Adam,
    F_model_depthCond.py:159-192): learning rates, l_pix, l_dynamic, eight watched parameters and the 10 loss weights
    after the second Adam step."""
    from dasr_amd import harness
    from tests.golden_cases import TRAIN_CASE
    g = load("train_step")
    case = TRAIN_CASE
    net, cfg = build_net(case, device)
    tr = harness.Trainer(net, cfg["depthRangeNum"])
    lq, gt, dm, mk = [t.to(device) for t in synth.closed_form_batch(0, case["B"], case["H"], case["W"], cfg["scale"])]
    errs = {}
    for step in (1, 2):
        log = tr.optimize_parameters(lq, gt, dm, mk)
        lr = tr.optimizer.param_groups[0]["lr"]
        assert abs(lr - float(g["lr%d" % step])) <= 1e-12, (step, lr)
        errs["l_pix%d" % step] = abs(float(log["l_pix"]) - float(g["l_pix%d" % step]))
        errs["l_dyn%d" % step] = abs(float(log["l_dynamic"]) - float(g["l_dyn%d" % step]))
        assert errs["l_pix%d" % step] <= 2e-6 and errs["l_dyn%d" % step] <= 2e-5, (step, errs)
    lr_sum = float(g["lr1"]) + float(g["lr2"])
    sd = net.state_dict()
    for k in case["watch"]:
        d = float(np.abs(sd[k].detach().cpu().numpy() - g["p." + k]).max())
        errs["p." + k] = d
        if any(z in k for z in ZERO_GRAD_KEYS):
            # mathematically zero gradient (a bias in front of an InstanceNorm): Adam turns its rounding noise into
            # +-lr steps, in the reference as well; only the bound |delta| <= 2 * sum(lr) is meaningful
            assert d <= 2.0 * lr_sum + 1e-7, (k, d)
        else:
            # Adam normalises every element by its own gradient history, so an element whose gradient is small moves
            # by a visible fraction of lr under fp32 rounding differences: measured 3.2e-5 of the 2e-3 total movement
            # for A_i_j.weight (emulator), printed by the test for the other tensors
            assert d <= 1e-4, (k, d)
    d = float(np.abs(tr.dynamic_loss.trainable_weight.detach().cpu().numpy() - g["p.loss_w"]).max())
    errs["p.loss_w"] = d
    assert d <= 2e-5, d
    return errs


def check_define_g(device):
    """networks.define_G (codes/models/networks.py:41-49) from the keys of the shipped x8 yml
    (options/train/train_depthNet_SEAN_depthMask_x8.yml:31-63): the constructed net has the reference's state_dict
    (keys, shapes, 14 795 971 parameters); depthRangeNum comes from datasets.train.depthMaskNum, or from
    datasets.test_1 when the first dataset is not 'train'; other generators are refused."""
    from dasr_amd import networks
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    opt = {"network_G": dict(networks.X8_NETWORK_G),
           "datasets": {"train": {"depthMaskNum": 10, "depthFixedRange": False}, "val": {"depthMaskNum": 10}}}
    net = networks.define_G(opt)
    sd = net.state_dict()
    assert [k for k, _ in keys["x8"]] == list(sd.keys())
    assert [tuple(s) for _, s in keys["x8"]] == [tuple(v.shape) for v in sd.values()]
    assert sum(p.numel() for p in net.parameters()) == keys["x8_nparams"] == 14795971
    assert net.scale == 8 and net.cfg["depthRangeNum"] == 10 and net.cfg["depth_latent_ch"] == 256
    # test-time option files list test_1 first (networks.py:44-47)
    opt_t = {"network_G": dict(networks.X8_NETWORK_G, nb=4, which_ResBlk_depth=[0, 1], depth_latent_ch=16),
             "datasets": {"test_1": {"depthMaskNum": 7}}}
    net_t = networks.define_G(opt_t)
    assert net_t.cfg["depthRangeNum"] == 7
    assert tuple(net_t.state_dict()["depth-residual1.norm1.A_i_j.weight"].shape) == (7, 7, 1, 1)
    # a small one runs
    net_t = net_t.to(device)
    synth.closed_form_fill_(net_t.state_dict().items())
    lq, _, dm, mk = [t.to(device) for t in synth.closed_form_batch(0, 1, 6, 8, 8, 7)]
    with torch.no_grad():
        out = net_t(lq, dm, mk)
    assert tuple(out.shape) == (1, 3, 48, 64)
    try:
        networks.define_G({"network_G": {"which_model_G": "RRDBNet"}, "datasets": {"train": {}}})
        raise AssertionError("expected NotImplementedError")
    except NotImplementedError:
        pass
    return dict(params=keys["x8_nparams"])


def check_depth_mask_golden(device):
    """getDepthMask pinned to the REFERENCE's own function (tests/golden/depth_masks.npz, oracle/make_golden.py section
    9): the host restatement synth.depth_to_masks and the device kernel prep.depth_to_masks (planes and region
    bytes) reproduce it bit for bit - data-range and fixed-range modes, pixels equal to the maximum, values outside
    [0,1), a constant map, pixels exactly on bin edges, K in {7, 10, 16}."""
    from dasr_amd import prep
    from tests.golden_cases import depth_mask_cases
    g = load("depth_masks")
    n = nobin = 0
    for name, (depth, fixed, K) in depth_mask_cases().items():
        want = torch.from_numpy(g[name].astype(np.float32))
        assert want.shape[0] == K
        host = synth.depth_to_masks(depth, K, fixed)
        assert torch.equal(host, want), ("synth.depth_to_masks", name)
        got = prep.depth_to_masks(depth.unsqueeze(0).to(device), K, fixed)
        assert torch.equal(got.cpu()[0], want), ("prep.depth_to_masks planes", name)
        idx = torch.where(want.sum(0) > 0, want.argmax(0), torch.full_like(want.argmax(0), K)).to(torch.uint8)
        assert torch.equal(graph.attached_region(got).cpu()[0], idx), ("region bytes", name)
        assert float(want.sum(0).max()) <= 1.0           # mutually exclusive bins
        nobin += int((idx == K).sum())
        n += 1
    assert n >= 15 and nobin > 0
    return dict(cases=n, no_bin_pixels=nobin)


def _fake_replica(net):
    """What torch.nn.parallel.replicate makes of a module (torch/nn/parallel/replicate.py): every module is shallow-
    copied with EMPTY _parameters, and the weights are set as plain (non-leaf, broadcast) tensor attributes."""
    mods = list(net.modules())
    copies = {m: m._replicate_for_data_parallel() for m in mods}
    for m in mods:
        r = copies[m]
        for key, child in m._modules.items():
            r._modules[key] = copies[child] if child is not None else None
        for key, p in m._parameters.items():
            if p is not None:
                setattr(r, key, p * 1.0)          # non-leaf stand-in for Broadcast.apply's output
    return copies[net]


def check_replica_protocol(device):
    """An nn.DataParallel replica exposes no parameters (torch >= 1.5): DepthNet must find its weights through the
    module tree's attributes.  A replica built the way replicate() builds one gives the bare net's output bit for
    bit, and its gradients flow back to the original parameters."""
    case = dict(name="dp", scale=2, which=[0, 1], L=16, nb=4, B=2, H=8, W=12)
    net, cfg = build_net(case, device)
    lq, gt, dm, mk = [t.to(device) for t in synth.closed_form_batch(0, 2, 8, 12, 2)]
    sr = net(lq, dm, mk)
    wgt = torch.cos(torch.arange(sr.numel(), dtype=torch.float32) * 0.013).reshape(sr.shape).to(device)
    (sr * wgt).sum().backward()
    want = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    net.zero_grad(set_to_none=True)
    rep = _fake_replica(net)
    assert len(list(rep.parameters())) == 0
    out = rep(lq, dm, mk)
    assert torch.equal(out.detach(), sr.detach())
    (out * wgt).sum().backward()
    worst = 0.0
    for k, p in net.named_parameters():
        if k in want:
            assert p.grad is not None, k
            worst = max(worst, rel_max(p.grad, want[k]) if want[k].abs().max() > 0 else 0.0)
        else:
            assert p.grad is None, k
    assert worst <= (0.0 if device == "cpu" else 1e-4), worst     # GPU: float atomics in the small wgrad sums
    return dict(worst_grad_rel=worst)


def check_data_parallel_gpu():
    """torch.nn.parallel.replicate(net, [0, 0]) + parallel_apply on ONE GPU (two replicas, two threads, two halves
    of the batch - the reference's default nn.DataParallel wrap, F_model_depthCond.py:31-35): outputs bit-identical
    to the bare net's, reduced gradients equal to the bare net's on the whole batch."""
    from torch.nn.parallel import parallel_apply, replicate
    case = dict(name="dp", scale=8, which=[0, 1], L=16, nb=4, B=4, H=8, W=12)
    net, cfg = build_net(case, "cuda")
    lq, gt, dm, mk = [t.cuda() for t in synth.closed_form_batch(0, 4, 8, 12, 8)]
    # (what is checked is the replica protocol, bit for bit: on the exact-fp32 kernels - the split convolutions are chosen by
    # launch size, which differs between the whole batch and its halves)
    split_was = graph.SPLIT_BF16
    graph.SPLIT_BF16 = False
    try:
        return _data_parallel_gpu_body(net, lq, gt, dm, mk)
    finally:
        graph.SPLIT_BF16 = split_was


def _data_parallel_gpu_body(net, lq, gt, dm, mk):
    from torch.nn.parallel import parallel_apply, replicate
    sr = net(lq, dm, mk)
    wgt = torch.cos(torch.arange(sr.numel(), dtype=torch.float32) * 0.013).reshape(sr.shape).cuda()
    (sr * wgt).sum().backward()
    want = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    net.zero_grad(set_to_none=True)
    reps = replicate(net, [0, 0])
    assert len(list(reps[0].parameters())) == 0
    halves = [(lq[:2].contiguous(), dm[:2].contiguous(), mk[:2].contiguous()),
              (lq[2:].contiguous(), dm[2:].contiguous(), mk[2:].contiguous())]
    outs = parallel_apply(reps, halves, devices=[0, 0])
    out = torch.cat(outs, 0)
    assert torch.equal(out.detach(), sr.detach())
    (out * wgt).sum().backward()
    torch.cuda.synchronize()
    num = den = 0.0
    for k, p in net.named_parameters():
        if k in want:
            assert p.grad is not None, k
            if any(z in k for z in ZERO_GRAD_KEYS):
                continue
            num += (p.grad.double() - want[k].double()).pow(2).sum().item()
            den += want[k].double().pow(2).sum().item()
        else:
            # the never-called block: Broadcast's backward hands zeros to unused replica weights (as it does for the
            # reference's own module under nn.DataParallel)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    rel = math.sqrt(num / den)
    assert rel <= 1e-5, rel
    # and through the wrapper class itself (one device: DataParallel calls the module directly)
    dp = torch.nn.DataParallel(net, device_ids=[0])
    with torch.no_grad():
        assert torch.equal(dp(lq, dm, mk), sr.detach())
    return dict(grad_rel_l2=rel)


def check_ddp_single_rank_gpu():
    """DistributedDataParallel(net, find_unused_parameters=True) on a one-rank nccl (= RCCL) group
    (F_model_depthCond.py:31-33): one step's output and gradients equal the bare net's (the never-called block's
    parameters stay without a gradient)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29573")
    case = dict(name="ddp", scale=4, which=[0, 1], L=16, nb=5, B=2, H=8, W=12)
    net, cfg = build_net(case, "cuda")
    lq, gt, dm, mk = [t.cuda() for t in synth.closed_form_batch(0, 2, 8, 12, 4)]
    sr = net(lq, dm, mk)
    wgt = torch.cos(torch.arange(sr.numel(), dtype=torch.float32) * 0.013).reshape(sr.shape).cuda()
    (sr * wgt).sum().backward()
    want = {k: (p.grad.clone() if p.grad is not None else None) for k, p in net.named_parameters()}
    assert any(v is None for v in want.values())        # block nb-2 is constructed but never called
    net.zero_grad(set_to_none=True)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0], find_unused_parameters=True)
        out = ddp(lq, dm, mk)
        assert torch.equal(out.detach(), sr.detach())
        (out * wgt).sum().backward()
        torch.cuda.synchronize()
        num = den = 0.0
        for k, p in net.named_parameters():
            if want[k] is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
                continue
            if any(z in k for z in ZERO_GRAD_KEYS):
                continue
            num += (p.grad.double() - want[k].double()).pow(2).sum().item()
            den += want[k].double().pow(2).sum().item()
        rel = math.sqrt(num / den)
        assert rel <= 1e-5, rel
    finally:
        dist.destroy_process_group()
    return dict(grad_rel_l2=rel)


def check_region_shortcut_invalidation(device):
    """The region bytes attached by prep.depth_to_masks are dropped when the mask tensor is edited in place afterwards
    (torch's version counter), so a flipped / overwritten mask cannot be read through stale bytes."""
    from dasr_amd import prep
    _, _, dm, _ = synth.closed_form_batch(0, 2, 12, 16, 1)
    mk = prep.depth_to_masks(dm.to(device), 10)
    assert graph.attached_region(mk) is not None
    flipped = torch.flip(mk, dims=[3]).contiguous()
    mk.copy_(flipped)                                   # in-place edit
    assert graph.attached_region(mk) is None
    assert prep.attach_region(mk) if device != "cpu" else True
    case = dict(name="inv", scale=2, which=[0, 1], L=16, nb=4, B=2, H=12, W=16)
    net, cfg = build_net(case, device)
    lq = synth.closed_form_batch(0, 2, 12, 16, 2)[0].to(device)
    with torch.no_grad():
        a = net(lq, dm.to(device), mk)
        b = net(lq, dm.to(device), flipped.clone())
    assert torch.equal(a, b)
    return dict(ok=True)


class _SplitApi:
    """The two split schemes behind one face: pieces = 3 (bf16 x 3, six products) / 2 (fp16 x 2, three products, every
    tensor operand with its dasr_absmax)."""

    def __init__(self, pieces):
        self.pieces = pieces

    def weights(self, wp):
        return ops.conv3x3_split_weights(wp) if self.pieces == 3 else ops.conv3x3_split2_weights(wp)

    def fwd(self, x, ws, bias, cout, residual=None, act=0, ps=1):
        if self.pieces == 3:
            return ops.conv3x3_fwd_split(x, ws, bias, cout, residual, act, ps)
        return ops.conv3x3_fwd_split2(x, ops.absmax(x), ws, bias, cout, residual, act, ps)

    def dgrad(self, dy, ws, x_shape, out=None):
        if self.pieces == 3:
            return ops.conv3x3_dgrad_split(dy, ws, x_shape, out=out)
        return ops.conv3x3_dgrad_split2(dy, ops.absmax(dy), ws, x_shape, out=out)

    def wgrad(self, x, dy):
        if self.pieces == 3:
            return ops.conv3x3_wgrad_split(x, dy)
        return ops.conv3x3_wgrad_split2(x, ops.absmax(x), dy, ops.absmax(dy))


def check_absmax(device, seed=3):
    """dasr_absmax against torch, incl. lengths that are not a multiple of four, negative extremes, zeros."""
    gen = torch.Generator().manual_seed(seed)
    for n in (4, 7, 1024, 4099, 300001):
        x = torch.randn(n + 4, generator=gen)[:n].clone()
        x[n // 2] = -77.5 if n % 2 else 91.25
        got = ops.amax_value(ops.absmax(x.to(device)))
        assert got == x.abs().max().item(), (n, got)
    assert ops.amax_value(ops.absmax(torch.zeros(64).to(device))) == 0.0
    big = torch.randn(3 * 1024 * 1024 + 5, generator=gen)          # many workgroups: many partial maxima
    assert ops.amax_value(ops.absmax(big.to(device))) == big.abs().max().item()
    return dict(ok=True)


def check_conv9_split(device, seed=9):
    """The 9x9 output convolution as fp16 x 2 split products (csrc/conv9_split.hip: forward, dgrad, accumulating dgrad,
    wgrad + bias gradient) against torch's FLOAT64 convolution, next to the exact-fp32 MFMA kernels it replaces - the same
    gates as check_split_conv (error <= 1.25x the fp32 kernel's + 1e-7 on the hardware, 3x + 2e-7 on the emulator, whose
    MFMA model rounds per product).  Ragged tiles, two samples, Cout = 3 and 1, operands scaled far from 1, and - impl 2 -
    three persistent workgroups walking long tile lists with the next tile's loads in flight."""
    gen = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=gen)
    fac, slack = (3.0, 2e-7) if device == "cpu" else (1.25, 1e-7)
    out = {}
    try:
        for (B, H, W, cout, impl, sx_, sd_) in ((1, 11, 70, 3, 0, 1.0, 1.0), (2, 19, 60, 3, 0, 37.0, 2e-6), (2, 19, 60, 3, 2, 1.0, 1.0),
                                               (1, 9, 33, 1, 2, 0.01, 5.0)):
            ops.set_conv_bf16_impl(impl)
            cin = 32
            assert ops.conv9_split_supported(H, W, cin, cout) and not ops.conv9_split_supported(H, W, 64, cout)
            x = rn(B, cin, H, W) * sx_
            w = rn(cout, cin, 9, 9) * 0.02
            bias = rn(cout) * 0.3 * sx_
            dy = rn(B, cout, H, W) * sd_
            x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
            ref = F.conv2d(x64, w64, bias.double(), padding=4)
            gx64, gw64 = torch.autograd.grad(ref, (x64, w64), dy.double())
            xd, dyd = nhwc(x).to(device), nhwc(dy).to(device)
            wp = ops.pack_hwio(w.permute(2, 3, 1, 0).contiguous().to(device))
            bd = bias.to(device)
            wm, xm, dm = ops.absmax(wp[0]), ops.absmax(xd), ops.absmax(dyd)
            y_sp = ops.conv9_fwd_split2(xd, xm, wp, wm, bd)
            y_32 = ops.conv2d_fwd(xd, wp, bd, pad=4)
            refd = ref.detach()
            e_sp = (nchw(y_sp.cpu()).double() - refd).abs().max().item() / refd.abs().max().item()
            e_32 = (nchw(y_32.cpu()).double() - refd).abs().max().item() / refd.abs().max().item()
            assert e_sp <= fac * e_32 + slack, ("conv9 fwd", B, H, W, cout, e_sp, e_32)
            dx_sp = ops.conv9_dgrad_split2(dyd, dm, wp, wm, xd.shape)
            dx_32 = ops.conv2d_dgrad(dyd, wp, xd.shape, pad=4)
            g_sp = (nchw(dx_sp.cpu()).double() - gx64).abs().max().item() / gx64.abs().max().item()
            g_32 = (nchw(dx_32.cpu()).double() - gx64).abs().max().item() / gx64.abs().max().item()
            assert g_sp <= fac * g_32 + slack, ("conv9 dgrad", B, H, W, cout, g_sp, g_32)
            base = rn(B, H, W, cin) * sd_
            accd = base.to(device).clone()
            ops.conv9_dgrad_split2(dyd, dm, wp, wm, xd.shape, out=accd)
            want = nhwc(gx64) + base.double()
            g_acc = (accd.cpu().double() - want).abs().max().item() / want.abs().max().item()
            assert g_acc <= fac * g_32 + 2 * slack, ("conv9 dgrad accumulate", g_acc, g_32)
            dw_sp, db_sp = ops.conv9_wgrad_split2(xd, xm, dyd, dm)
            dw_32, db_32 = ops.conv2d_wgrad(xd, dyd, (9, 9, cin, cout), pad=4)
            w_sp, w_32 = rel_max(dw_sp.permute(3, 2, 0, 1), gw64), rel_max(dw_32.permute(3, 2, 0, 1), gw64)
            assert w_sp <= fac * w_32 + slack, ("conv9 wgrad", B, H, W, cout, w_sp, w_32)
            assert rel_max(db_sp, dy.double().sum((0, 2, 3))) <= 2 * rel_max(db_32, dy.double().sum((0, 2, 3))) + 1e-6
            out["%dx%dx%d co%d impl%d" % (B, H, W, cout, impl)] = tuple(float("%.3g" % v) for v in (e_sp, e_32, g_sp, g_32, g_acc, w_sp, w_32))
        # >= 256 tiles: the forward deals its tiles to the XCDs in 4 x 8 blocks (ragged blocks in both directions here, and more
        # blocks than one trip holds) - bit-identical to the linear tile order (impl + 4096), which the cases above ran
        B, H, W, cout = 3, 76, 500, 3
        x = (rn(B, H, W, 32) * 1.7).to(device)
        wp = ops.pack_hwio((rn(9, 9, 32, cout) * 0.02).to(device))
        bd = (rn(cout) * 0.3).to(device)
        wm, xm = ops.absmax(wp[0]), ops.absmax(x)
        y_blocked = ops.conv9_fwd_split2(x, xm, wp, wm, bd)
        ops.set_conv_bf16_impl(4096)
        y_linear = ops.conv9_fwd_split2(x, xm, wp, wm, bd)
        assert torch.equal(y_blocked, y_linear), "conv9 fwd: XCD-blocked tile order differs from the linear one"
        y_32 = ops.conv2d_fwd(x, wp, bd, pad=4)
        assert rel_max(y_blocked, y_32.double()) <= 1e-5
        out["blocked tile order"] = "bit-identical to linear (%d tiles)" % (B * ((H + 7) // 8) * ((W + 55) // 56))
    finally:
        ops.set_conv_bf16_impl(0)
    return out


def check_fused_amax(device, seed=11):
    """Every producer kernel that leaves max |.| of what it stores behind (the *_amax buffers of dasr.h) against torch's
    abs().max() of the tensor it wrote - EXACT (the kernels take maxima of the stored values themselves): SEAN forward
    (one-hot and soft masks, with / without ReLU and residual), SEAN backward (dt, dgb2), the mask layer, a convolution
    whose kernel does not track it (follow-up pass inside the entry point), the activation / PixelShuffle backward in its
    three forms, the fp16 x 2 split forward (plain, residual + ReLU, PixelShuffle(2), the 32-channel tile)."""
    gen = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=gen)
    dev = lambda t: t.to(device)
    z = lambda: ops.amax_buffer(torch.zeros(1).to(device)).fill_(float("nan"))       # (poisoned: the producer must write every word it declares)
    amx = lambda t: t.detach().abs().max().item()
    out = {}
    # ---- SEAN
    B, H, W, C, K = 2, 9, 37, 64, 10
    t, gb2, res = rn(B, H, W, C), rn(B, H, W, 2 * C), rn(B, H, W, C) * 3.0
    _, _, _, mk = synth.closed_form_batch(1, B, H, W, 1, K)
    D = rn(B, 2, 9, K, C) * 0.1
    bg, bb = rn(C) * 0.1, rn(C) * 0.1
    ag, ab = torch.full((1,), 0.7), torch.full((1,), 0.74)
    dout = rn(B, H, W, C) * 1e-5
    mean, var = ops.instnorm_stats(dev(t))
    for soft in (False, True):
        mask = dev(mk if not soft else (0.7 * mk + 0.3 * torch.rand(mk.shape, generator=gen)))
        region, flag = ops.mask_compress(mask)
        for relu, r in ((True, None), (False, None), (True, res), (False, res)):
            a = z()
            y = ops.sean_fwd(dev(t), mean, var, dev(gb2), mask, region, flag, dev(D), dev(bg), dev(bb), dev(ag), dev(ab),
                             dev(r) if r is not None else None, relu, amax=a)
            assert ops.get_amax(y) is a and ops.amax_value(a) == amx(y), ("sean fwd", soft, relu, r is not None, ops.amax_value(a), amx(y))
            y0 = ops.sean_fwd(dev(t), mean, var, dev(gb2), mask, region, flag, dev(D), dev(bg), dev(bb), dev(ag), dev(ab),
                              dev(r) if r is not None else None, relu)
            assert torch.equal(y, y0) and ops.get_amax(y0) is None
            a1, a2 = z(), z()
            rr = ops.sean_bwd(dev(dout), y, dev(t), mean, var, dev(gb2), mask, region, flag, dev(D), dev(bg), dev(bb), dev(ag),
                              dev(ab), relu, r is not None, dt_amax=a1, dgb2_amax=a2)
            assert ops.amax_value(a1) == amx(rr[0]) and ops.amax_value(a2) == amx(rr[1]), ("sean bwd", soft, relu, ops.amax_value(a1), amx(rr[0]), ops.amax_value(a2), amx(rr[1]))
            assert ops.amax_value(a1) > 0 and ops.amax_value(a2) > 0
    out["sean"] = True
    # ---- the mask layer (kernel tracks it) and a 3 -> 32 layer on the same entry point, a 3x3 MFMA convolution (follow-up pass)
    for (cin, cout, act) in ((1, 128, ops.ACT_RELU), (3, 32, ops.ACT_LRELU), (32, 64, ops.ACT_NONE)):
        x = rn(2, 11, 37, cin)
        w = ops.pack_hwio(dev(rn(3, 3, cin, cout) * 0.3))
        b = dev(rn(cout))
        a = z()
        y = ops.conv2d_fwd(dev(x), w, b, act=act, amax=a)
        assert ops.amax_value(a) == amx(y), ("conv fwd amax", cin, cout, ops.amax_value(a), amx(y))
        assert torch.equal(y, ops.conv2d_fwd(dev(x), w, b, act=act))
    # ---- activation / PixelShuffle backward
    for (Cq, act, ps, Hs, Ws) in ((8, 1, 1, 6, 5), (8, 2, 2, 6, 5), (4, 2, 3, 5, 7), (3, 2, 1, 5, 7)):
        yv, dyv = rn(2, Hs * ps, Ws * ps, Cq), rn(2, Hs * ps, Ws * ps, Cq) * 1e-4
        a = z()
        d = ops.conv2d_epilogue_bwd(dev(dyv), dev(yv), Hs, Ws, Cq * ps * ps, act, ps, amax=a)
        assert ops.amax_value(a) == amx(d), ("epilogue bwd amax", Cq, act, ps, ops.amax_value(a), amx(d))
        assert torch.equal(d, ops.conv2d_epilogue_bwd(dev(dyv), dev(yv), Hs, Ws, Cq * ps * ps, act, ps))
    # ---- fp16 x 2 split forward
    for (cin, cout, act, with_res, ps, Hs, Ws) in ((64, 64, 0, False, 1, 17, 35), (32, 128, 1, True, 1, 9, 33), (32, 128, 2, False, 2, 9, 33),
                                                    (64, 32, 2, False, 1, 9, 40)):
        x = rn(1, Hs, Ws, cin) * 5.0
        wp = ops.pack_hwio(dev(rn(3, 3, cin, cout) * (1.0 / math.sqrt(9 * cin))))
        ws = ops.conv3x3_split2_weights(wp)
        b = dev(rn(cout))
        r = dev(rn(1, Hs, Ws, cout)) if with_res else None
        xm = ops.absmax(dev(x))
        a = z()
        y = ops.conv3x3_fwd_split2(dev(x), xm, ws, b, cout, r, act, ps, amax=a)
        assert ops.amax_value(a) == amx(y), ("split fwd amax", cin, cout, act, ps, ops.amax_value(a), amx(y))
        assert torch.equal(y, ops.conv3x3_fwd_split2(dev(x), xm, ws, b, cout, r, act, ps))
    # ---- an in-place writer drops what the tensor carried
    y = ops.conv2d_fwd(dev(rn(1, 5, 5, 1)), ops.pack_hwio(dev(rn(3, 3, 1, 8))), None, amax=z())
    assert ops.get_amax(y) is not None
    ops.accumulate_(y, y.clone())
    assert ops.get_amax(y) is None
    return out


def check_fused_amax_net(device, case_name="x8_nb4"):
    """Whole net with the fp16 x 2 split convolutions forced on: the maxima left behind by the producing kernels
    (graph.FUSE_AMAX) against one dasr_absmax pass per operand - the same scales, so the output is BIT-identical (and, on the
    deterministic emulator, every gradient); and the fused run must launch far fewer absmax passes."""
    from dasr_amd import graph
    case = [c for c in DEPTHNET_CASES if c["name"] == case_name][0]
    net, cfg = build_net(case, device)
    lq, gt, dm, mk = [t.to(device) for t in synth.closed_form_batch(0, case["B"], case["H"], case["W"], cfg["scale"])]
    old = graph.SPLIT_MIN_PIXELS, graph.SPLIT_PIECES, graph.FUSE_AMAX, graph.PREPACK
    calls = {}
    orig_absmax = ops.absmax
    res = {}
    try:
        graph.SPLIT_MIN_PIXELS, graph.SPLIT_PIECES, graph.PREPACK = 0, 2, False     # (per-tensor weight maxima in both runs)
        for fused in (True, False):
            graph.FUSE_AMAX = fused
            n = [0]

            def counted(x, _n=n):
                _n[0] += 1
                return orig_absmax(x)

            ops.absmax = counted
            net.zero_grad(set_to_none=True)
            sr = net(lq, dm, mk)
            wgt = torch.cos(torch.arange(sr.numel(), dtype=torch.float32) * 0.013).reshape(sr.shape).to(device)
            (sr * wgt).sum().backward()
            res[fused] = (sr.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
            calls[fused] = n[0]
    finally:
        ops.absmax = orig_absmax
        graph.SPLIT_MIN_PIXELS, graph.SPLIT_PIECES, graph.FUSE_AMAX, graph.PREPACK = old
    assert torch.equal(res[True][0], res[False][0]), "forward differs between fused and separate maxima"
    # (the GPU's weight / bias gradient reductions end in float atomics: two identical runs differ in the last bits there)
    for k, gten in res[False][1].items():
        if device == "cpu":
            assert torch.equal(res[True][1][k], gten), ("gradient differs", k)
        else:
            d = (res[True][1][k] - gten).double().norm().item()
            assert d <= 2e-5 * max(gten.double().norm().item(), 1e-30) or any(z in k for z in ZERO_GRAD_KEYS), ("gradient differs", k, d)
    assert calls[True] * 2 <= calls[False], calls          # (what is left: the kernels' own maxima, a few unfused producers)
    return dict(absmax_passes_fused=calls[True], absmax_passes_separate=calls[False])


def check_prepack_ops(device, seed=9):
    """dasr_weight_pack_multi / dasr_conv3x3_split2_weights_multi against the per-tensor entry points they batch: a
    weight-normed kernel, a transposed one, two plain kernels side by side in one packed buffer, a bf16 packed kernel and a
    bias pair in ONE launch - bit-identical buffers, 1/||v|| vectors, and (through the amax buffers the jobs fill) the same
    max |w| and fp16 x 2 images as dasr_absmax + dasr_conv3x3_split2_weights."""
    gen = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=gen).to(device)
    v1, g1 = rn(64, 32, 3, 3), rn(64, 1, 1, 1).abs() + 0.5
    v2, g2 = rn(32, 48, 3, 3), rn(32, 1, 1, 1).abs() + 0.5            # ConvTranspose2d layout [I][O][KH][KW]
    va, vb = rn(64, 128, 3, 3) * 0.05, rn(64, 128, 3, 3) * 3.0
    v3, g3 = rn(32, 32, 3, 3), rn(32, 1, 1, 1).abs() + 0.5
    v9 = rn(3, 32, 9, 9)
    ba, bb = rn(64), rn(64)
    # per-tensor reference
    w1, i1 = ops.weight_pack(v1, g1)
    w2, i2 = ops.weight_pack(v2, g2, transposed=True)
    wp = ops.empty((2, 3, 3, 128, 128), va)
    ops.weight_pack(va, None, False, out=wp, o_off=0)
    ops.weight_pack(vb, None, False, out=wp, o_off=64)
    w3, i3 = ops.weight_pack(v3, g3, dtype=torch.bfloat16)
    w9, _ = ops.weight_pack(v9, None)
    s1, sp = ops.conv3x3_split2_weights(w1), ops.conv3x3_split2_weights(wp)
    m9 = ops.amax_value(ops.absmax(w9[0]))
    # one launch
    W1, W2, WP, W3, W9 = (torch.full_like(t, 7.0) for t in (w1, w2, wp, w3, w9))
    I1, I2, I3 = (torch.full_like(t, 7.0) for t in (i1, i2, i3))
    BP = torch.full((128,), 7.0, device=device)
    A1, AP, A9 = ops.amax_buffer(v1), ops.amax_buffer(v1), ops.amax_buffer(v1)
    jobs, n = [], 0
    for args in ((v1, g1, W1, I1, A1, False, 0, False), (v2, g2, W2, I2, None, True, 0, False),
                 (va, None, WP, None, AP, False, 0, False), (vb, None, WP, None, AP, False, 64, False),
                 (v3, g3, W3, I3, None, False, 0, False), (ba, None, BP, None, None, False, 0, True),
                 (bb, None, BP, None, None, False, 64, True), (v9, None, W9, None, A9, False, 0, False)):
        job, k = ops.pack_job(*args, n)
        jobs.append(job)
        n += k
    assert ctypes.sizeof(ops.PackJob) == 80 and ctypes.sizeof(ops.SplitJob) == 40
    ops.weight_pack_multi(ops.JobTable(jobs, v1.device, None))
    for a, b, nm in ((w1, W1, "wn"), (w2, W2, "transposed"), (wp, WP, "pair"), (w3, W3, "bf16"), (w9, W9, "9x9"), (i1, I1, "inv"),
                     (i2, I2, "inv t"), (i3, I3, "inv bf16"), (torch.cat([ba, bb]), BP, "bias pair")):
        assert torch.equal(a, b), nm
    assert ops.amax_value(A1) == ops.amax_value(s1[1]) and ops.amax_value(AP) == ops.amax_value(sp[1]) and ops.amax_value(A9) == m9
    S1, SP = torch.zeros_like(s1[0]), torch.zeros_like(sp[0])
    j1, k1 = ops.split_job(W1, A1, S1, 0)
    j2, _ = ops.split_job(WP, AP, SP, k1)
    ops.conv3x3_split2_weights_multi(ops.JobTable([j1, j2], v1.device, None))
    assert torch.equal(S1, s1[0]) and torch.equal(SP, sp[0])
    # a table whose wg_begin does not add up is refused before anything is launched
    bad, _ = ops.pack_job(v1, g1, W1, I1, A1, False, 0, False, 5)
    try:
        ops.weight_pack_multi(ops.JobTable([bad], v1.device, None))
        raise AssertionError("an inconsistent job table was accepted")
    except RuntimeError:
        pass
    return dict(jobs=len(jobs), workgroups=n)


def check_prepack_net(device, case_name="x8_nb4", steps=3):
    """graph.PREPACK: the packed kernels of a training step refilled in two launches from the second step on.  Three steps (every
    parameter changed in place in between, so that every pack sees new weights) with and without: the same outputs and gradients to the bit
    on the emulator (GPU: forward to the bit, gradients to reduction noise), and no per-tensor pack / split / absmax-of-weights
    call once the tables are ready."""
    from dasr_amd import graph
    case = ([c for c in DEPTHNET_CASES if c["name"] == case_name] or
            [dict(name="x2_one_depth_block", scale=2, which=[0], L=16, nb=4, B=1, H=8, W=12)])[0]    # (the emulator's size)
    old = graph.SPLIT_MIN_PIXELS, graph.SPLIT_PIECES, graph.PREPACK
    counted = ("weight_pack", "conv3x3_split2_weights", "copy_")
    orig = {k: getattr(ops, k) for k in counted}
    res, calls = {}, {}
    try:
        graph.SPLIT_MIN_PIXELS, graph.SPLIT_PIECES = 0, 2
        for pre in (True, False):
            graph.PREPACK = pre
            net, cfg = build_net(case, device)
            lq, gt, dm, mk = [t.to(device) for t in synth.closed_form_batch(0, case["B"], case["H"], case["W"], cfg["scale"])]
            res[pre], calls[pre] = [], []
            for step in range(steps):
                n = {k: 0 for k in counted}
                for k in counted:
                    def wrapped(*a, _k=k, _n=n, **kw):
                        _n[_k] += 1
                        return orig[_k](*a, **kw)
                    setattr(ops, k, wrapped)
                net.zero_grad(set_to_none=True)
                sr = net(lq, dm, mk)
                wgt = torch.cos(torch.arange(sr.numel(), dtype=torch.float32) * 0.013).reshape(sr.shape).to(device)
                (sr * wgt).sum().backward()
                grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
                res[pre].append((sr.detach().clone(), grads))
                calls[pre].append(dict(n))
                with torch.no_grad():               # (not along the gradients: on the GPU their last bits vary from run to run)
                    for i, (k, p) in enumerate(net.named_parameters()):
                        bump = torch.cos(torch.arange(p.numel(), dtype=torch.float32) * 0.37 + i + step).reshape(p.shape).to(device)
                        p.add_(bump * p.abs().max(), alpha=2e-3)
            if pre:
                plans = list(net._prepack.values())
                assert len(plans) == 1 and plans[0].ready and plans[0].tables[1] is not None
    finally:
        for k in counted:
            setattr(ops, k, orig[k])
        graph.SPLIT_MIN_PIXELS, graph.SPLIT_PIECES, graph.PREPACK = old
    for step in range(steps):
        assert torch.equal(res[True][step][0], res[False][step][0]), ("forward differs", step)
        for k, gten in res[False][step][1].items():
            if device == "cpu":
                assert torch.equal(res[True][step][1][k], gten), ("gradient differs", step, k)
            else:
                d = (res[True][step][1][k] - gten).double().norm().item()
                assert d <= 2e-5 * max(gten.double().norm().item(), 1e-30) or any(z in k for z in ZERO_GRAD_KEYS), ("gradient differs", step, k, d)
    assert calls[True][0] == calls[False][0], (calls[True][0], calls[False][0])        # the recorded step IS the per-tensor path
    assert all(c["weight_pack"] == 0 and c["conv3x3_split2_weights"] == 0 and c["copy_"] <= calls[False][0]["copy_"] - 4
               for c in calls[True][1:]), calls[True]                                   # (copy_: the bias pairs are jobs too)
    assert all(c["weight_pack"] > 10 and c["conv3x3_split2_weights"] > 5 for c in calls[False]), calls[False]
    return dict(per_tensor_calls=calls[False][0], prepacked_calls=calls[True][-1])


def check_split_conv(device, seed=5, pieces=3):
    """dasr_conv3x3_{fwd,dgrad}_split - fp32 convolutions as six bf16 MFMA products of three-piece operands - against
    torch's FLOAT64 convolution of the same fp32 operands, next to the exact-fp32 MFMA kernels they replace
    (dasr_conv2d_fwd / dasr_conv2d_dgrad): the split kernels must be as close to the float64 result as the fp32 kernels are.
    Measured on the MI355X: split error 0.75 .. 1.0x the fp32 kernel's in every case (the bf16 MFMA sums a 16-deep block before
    it rounds into the accumulator; the exact-fp32 MFMA rounds per product) - gate 1.25x + 1e-7 of the largest value.  The
    CPU emulator models every MFMA as an fp32 fma chain, six times as many roundings as the hardware's: 1.9 .. 2.4x there,
    gate 3x + 2e-7.  Ragged tiles, one to
    four channel slices' worth of rows, 64 / 128 / 256 channels, accumulating dgrad, and - second pass - one workgroup per
    XCD walking a list of items.
    pieces = 2: the fp16 x 2 scheme (three products, per-tensor power-of-two scales from dasr_absmax) under the SAME gates;
    its operands are additionally scaled far from 1 (x by 2^9 x 1.3, dy by 3e-8: a gradient-sized tensor) - the error
    measures are relative, so only the scale handling can tell."""
    gen = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=gen)
    out = {}
    fac, slack = (3.0, 2e-7) if device == "cpu" else (1.25, 1e-7)
    api = _SplitApi(pieces)
    sx_, sd_ = (665.6, 3e-8) if pieces == 2 else (1.0, 1.0)
    # (impl + 64: the first fp16 x 2 weight-gradient kernel - operands split per K-step - for every block shape; GPU only)
    # impl + 256 + 512: the alternative forms of the forward / dgrad kernel - four waves with four tile rows each at 128
    # produced channels, one barrier per chunk at 64
    # impl + 1024: swaps the two eight-wave forms - kernel-row K-steps at 128 produced channels (default there: 4 row groups x 2
    # channel halves, kernel-column K-steps), the halves form at 64
    for mode in (0, 2) + ((64, 768, 770, 1024, 1026) if pieces == 2 and device != "cpu" else ((1024,) if pieces == 2 else ())):
        one_wg = mode
        ops.set_conv_bf16_impl(mode)
        shapes = [(64, 64, 1, 17, 35), (128, 128, 1, 16, 32), (128, 64, 2, 33, 40), (64, 256, 1, 9, 70)]
        if device == "cpu":           # the emulator runs ~50 M MAC/s: a subset that still covers every code path
            shapes = {0: [(64, 64, 1, 17, 35), (128, 128, 1, 9, 20)], 2: [(64, 128, 2, 17, 33)], 768: [(64, 128, 2, 17, 33), (64, 64, 1, 17, 35)],
                      1024: [(64, 128, 2, 17, 33), (64, 64, 1, 17, 35)]}[mode]
        try:
            for (cin, cout, B, H, W) in shapes:
                x = rn(B, cin, H, W) * (1.0 + rn(B, cin, 1, 1).abs()) * sx_
                w = rn(cout, cin, 3, 3) * (1.0 / math.sqrt(9 * cin))
                bias = rn(cout) * 0.3 * sx_
                x64, w64, b64 = x.double().requires_grad_(True), w.double().requires_grad_(True), bias.double()
                ref = F.conv2d(x64, w64, b64, padding=1)
                dy = rn(B, cout, H, W) * sd_
                gx64, = torch.autograd.grad(ref, x64, dy.double())
                assert ops.conv3x3_split_supported(H, W, cin, cout)
                xd = nhwc(x).to(device)
                wp = ops.pack_hwio(w.permute(2, 3, 1, 0).contiguous().to(device))
                ws = api.weights(wp)
                bd = bias.to(device)
                y_sp = api.fwd(xd, ws, bd, cout)
                y_32 = ops.conv2d_fwd(xd, wp, bd)
                e_sp = (nchw(y_sp.cpu()).double() - ref.detach()).abs().max().item() / ref.detach().abs().max().item()
                e_32 = (nchw(y_32.cpu()).double() - ref.detach()).abs().max().item() / ref.detach().abs().max().item()
                assert e_sp <= fac * e_32 + slack, ("fwd", cin, cout, H, W, e_sp, e_32)
                dyd = nhwc(dy).to(device)
                dx_sp = api.dgrad(dyd, ws, xd.shape)
                dx_32 = ops.conv2d_dgrad(dyd, wp, xd.shape)
                g_sp = (nchw(dx_sp.cpu()).double() - gx64).abs().max().item() / gx64.abs().max().item()
                g_32 = (nchw(dx_32.cpu()).double() - gx64).abs().max().item() / gx64.abs().max().item()
                assert g_sp <= fac * g_32 + slack, ("dgrad", cin, cout, H, W, g_sp, g_32)
                # weight / bias gradient
                gw64, = torch.autograd.grad(F.conv2d(x64, w64, b64, padding=1), w64, dy.double())
                gb64 = dy.double().sum((0, 2, 3))
                dw_sp, db_sp = api.wgrad(xd, dyd)
                dw_32, db_32 = ops.conv2d_wgrad(xd, dyd, (3, 3, cin, cout))
                w_sp, w_32 = rel_max(dw_sp.permute(3, 2, 0, 1), gw64), rel_max(dw_32.permute(3, 2, 0, 1), gw64)
                assert w_sp <= fac * w_32 + slack, ("wgrad", cin, cout, H, W, w_sp, w_32)
                assert rel_max(db_sp, gb64) <= 2 * rel_max(db_32, gb64) + 1e-6, ("dbias", cin, cout)
                base = rn(B, H, W, cin) * sd_
                accd = base.to(device).clone()
                api.dgrad(dyd, ws, xd.shape, out=accd)
                want = nhwc(gx64) + base.double()
                g_acc = (accd.cpu().double() - want).abs().max().item() / want.abs().max().item()
                assert g_acc <= fac * g_32 + 2 * slack, ("dgrad accumulate", cin, cout, g_acc)
                out["%d->%d %dx%d%s" % (cin, cout, H, W, {0: "", 2: " 1wg", 64: " wgrad-v1", 128: " wgrad-v2", 768: " alt forms", 770: " alt forms 1wg", 1024: " swapped forms", 1026: " swapped forms 1wg"}[mode])] = tuple(
                    float("%.3g" % v) for v in (e_sp, e_32, g_sp, g_32, g_acc, w_sp, w_32))
            # fused epilogues and the 32-channel tile (the HR tail: classic blocks, upscale convs): act, residual, PixelShuffle(2)
            eshapes = [(32, 32, 1, True, 1, 1, 17, 35), (64, 32, 2, False, 1, 2, 9, 40), (32, 128, 2, False, 2, 1, 10, 33),
                       (64, 256, 2, False, 2, 1, 16, 32)]
            if device == "cpu":
                eshapes = eshapes[:2] if mode == 0 else eshapes[2:3]
            if mode == 64:
                eshapes = eshapes[:3]
            if mode in (768, 770, 1024, 1026):
                eshapes = eshapes[2:] if device != "cpu" else eshapes[2:3]
            for (cin, cout, act, res, ps, B, H, W) in eshapes:
                x = rn(B, cin, H, W)
                w = rn(cout, cin, 3, 3) * (1.0 / math.sqrt(9 * cin))
                bias = rn(cout) * 0.3
                r = rn(B, cout, H, W) if res else None
                ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
                if ps > 1:
                    ref = F.pixel_shuffle(ref, ps)
                if res:
                    ref = ref + r.double()
                ref = F.relu(ref) if act == 1 else (F.leaky_relu(ref, 0.2) if act == 2 else ref)
                xd = nhwc(x).to(device)
                wp = ops.pack_hwio(w.permute(2, 3, 1, 0).contiguous().to(device))
                ws = api.weights(wp)
                rd = nhwc(r).to(device) if res else None
                y_sp = api.fwd(xd, ws, bias.to(device), cout, rd, act, ps)
                y_32 = ops.conv2d_fwd(xd, wp, bias.to(device), rd, 1, 1, False, act, ps)
                assert tuple(y_sp.shape) == tuple(y_32.shape) == tuple(nhwc(ref).shape)
                e_sp = (nchw(y_sp.cpu()).double() - ref).abs().max().item() / ref.abs().max().item()
                e_32 = (nchw(y_32.cpu()).double() - ref).abs().max().item() / ref.abs().max().item()
                assert e_sp <= fac * e_32 + slack, ("fwd epilogue", cin, cout, act, res, ps, e_sp, e_32)
                dy = rn(B, cout, H, W)
                gx64, = torch.autograd.grad(F.conv2d(x.double().requires_grad_(True), w.double(), None, padding=1),
                                            [], dy.double(), allow_unused=True) if False else (None,)
                x64 = x.double().requires_grad_(True)
                gx64, = torch.autograd.grad(F.conv2d(x64, w.double(), None, padding=1), x64, dy.double())
                dx_sp = api.dgrad(nhwc(dy).to(device), ws, xd.shape)
                g_sp = (nchw(dx_sp.cpu()).double() - gx64).abs().max().item() / gx64.abs().max().item()
                assert g_sp <= 4e-6, ("dgrad 32-channel", cin, cout, g_sp)
                # weight gradient of the smaller (ci, co) blocks: 32 x 32, 64 x 32, 32 x 128 (waves split the K-steps)
                w64 = w.double().requires_grad_(True)
                gw64, = torch.autograd.grad(F.conv2d(x.double(), w64, None, padding=1), w64, dy.double())
                dw_sp, db_sp = api.wgrad(xd, nhwc(dy).to(device))
                dw_32, _ = ops.conv2d_wgrad(xd, nhwc(dy).to(device), (3, 3, cin, cout))
                w_sp, w_32 = rel_max(dw_sp.permute(3, 2, 0, 1), gw64), rel_max(dw_32.permute(3, 2, 0, 1), gw64)
                assert w_sp <= fac * w_32 + slack, ("wgrad small block", cin, cout, w_sp, w_32)
                assert rel_max(db_sp, dy.double().sum((0, 2, 3))) <= 2e-6
                out["%d->%d act%d res%d ps%d%s" % (cin, cout, act, res, ps, {0: "", 2: " 1wg", 64: " wgrad-v1", 128: " wgrad-v2", 768: " alt forms", 770: " alt forms 1wg", 1024: " swapped forms", 1026: " swapped forms 1wg"}[mode])] = (
                    float("%.3g" % e_sp), float("%.3g" % e_32), float("%.3g" % g_sp))
        finally:
            ops.set_conv_bf16_impl(0)
    return out


# =====================================================================================================================
# Mixed precision (bf16 activations; BASELINE.json configs[2..3])
# =====================================================================================================================
BF16 = torch.bfloat16


def _bf(x):
    """Round to bf16 and come back: the value a bf16 tensor holds, as float32."""
    return x.to(BF16).to(torch.float32)


def check_bf16_conv_variants(device, seed=0, impl=0):
    """dasr_conv2d_{fwd,dgrad,wgrad}_bf16 (v_mfma_f32_32x32x16_bf16 kernels, transposed LDS reads in the weight gradient)
    against torch's fp32 convolution of the SAME bf16-rounded operands: the only differences left are the fp32
    accumulation order and the final rounding of the bf16 outputs (half an ulp = 2^-9 relative), so the gates are
    max |err| <= 2^-8 * max|ref| for bf16 outputs and 1e-5 relative for the fp32 weight / bias gradients.
    Covers every (MT, NTW) block of the weight gradient, both N-tile widths of the forward, ragged tile rows /
    columns, fused bias / ReLU / LeakyReLU / residual / PixelShuffle(2, 3) epilogues and the accumulating dgrad.
    ``impl`` (ops.set_conv_bf16_impl): 0 = the persistent LDS-DMA kernel where it applies (1, 3 and 4 channel chunks, one
    to three channel slices per pixel tile, both tile widths), 2 = the same with ONE workgroup per XCD, so that every
    workgroup walks a list of items (cross-item prefetch, accumulator re-initialisation, slice changes), 1 = the first
    kernel everywhere."""
    ops.set_conv_bf16_impl(impl)
    try:
        return _check_bf16_conv_variants(device, seed, impl)
    finally:
        ops.set_conv_bf16_impl(0)


def _check_bf16_conv_variants(device, seed, impl):
    gen = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=gen)
    cases = [  # cin, cout, act, ps_r, residual, B, H, W
        (32, 32, 1, 1, True, 1, 8, 32), (64, 64, 0, 1, False, 2, 9, 33), (32, 64, 2, 1, False, 1, 16, 64),
        (64, 32, 2, 1, False, 1, 7, 40), (128, 128, 0, 1, False, 1, 8, 32), (64, 256, 2, 2, False, 1, 6, 32),
        (32, 128, 2, 2, False, 1, 10, 35), (64, 288, 2, 3, False, 1, 5, 32), (64, 64, 1, 1, True, 1, 8, 30),
        (96, 192, 1, 1, False, 2, 19, 70), (128, 256, 2, 1, True, 1, 17, 45), (256, 64, 0, 1, False, 3, 9, 66),
    ]
    n_v2 = ops.conv_bf16_v2_launches()
    if device != "cpu":        # BASELINE.json configs[2] frame size: many tile rounds, both N slices, 8 XCD-ordered waves of blocks
        cases += [(64, 64, 0, 1, False, 2, 256, 320), (128, 128, 0, 1, False, 1, 256, 320), (32, 128, 2, 2, False, 1, 512, 640)]
    worst = {}
    for (cin, cout, act, ps, res, B, H, W) in cases:
        x = _bf(rn(B, cin, H, W))
        w = _bf(rn(cout, cin, 3, 3) * (1.0 / math.sqrt(9 * cin)))
        bias = rn(cout) * 0.1
        r = _bf(rn(B, cout, H, W)) if res else None
        xt, wt, bt = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        ref = F.conv2d(xt, wt, bt, padding=1)
        if ps > 1:
            ref = F.pixel_shuffle(ref, ps)
        if res:
            ref = ref + r
        ref = F.relu(ref) if act == 1 else (F.leaky_relu(ref, 0.2) if act == 2 else ref)
        # ---- forward
        xd = nhwc(x).to(device).to(BF16)
        wp = ops.pack_hwio(w.permute(2, 3, 1, 0).contiguous().to(device).to(BF16))
        y = ops.conv2d_fwd(xd, wp, bias.to(device), nhwc(r).to(device).to(BF16) if res else None, 1, 1, False, act, ps)
        assert y.dtype == BF16 and tuple(y.shape) == tuple(nhwc(ref).shape)
        e = (nchw(y.float().cpu()) - ref.detach()).abs().max().item() / ref.detach().abs().max().item()
        assert e <= 2.0 ** -8, ("fwd", cin, cout, act, ps, res, e)
        # ---- backward of the pure convolution (epilogue off): dgrad, accumulating dgrad, wgrad + bias gradient
        dy = _bf(rn(B, cout, H, W))
        ref0 = F.conv2d(xt, wt, bt, padding=1)
        gx, gw, gb = torch.autograd.grad(ref0, (xt, wt, bt), dy)
        dyd = nhwc(dy).to(device).to(BF16)
        dx = ops.conv2d_dgrad(dyd, wp, xd.shape, out_dtype=BF16)
        e1 = (nchw(dx.float().cpu()) - gx).abs().max().item() / gx.abs().max().item()
        assert dx.dtype == BF16 and e1 <= 2.0 ** -8, ("dgrad", cin, cout, e1)
        base = _bf(rn(B, H, W, cin))
        acc = base.to(device).to(BF16).clone()
        ops.conv2d_dgrad(dyd, wp, xd.shape, out=acc)
        want = nhwc(gx) + base
        e2 = (acc.float().cpu() - want).abs().max().item() / want.abs().max().item()
        assert e2 <= 2.0 ** -7, ("dgrad accumulate", cin, cout, e2)        # dx is rounded before the add as well
        dw, db = ops.conv2d_wgrad(xd, dyd, (3, 3, cin, cout))
        assert dw.dtype == torch.float32 and db.dtype == torch.float32
        e3 = rel_max(dw.permute(3, 2, 0, 1), gw)
        e4 = rel_max(db, gb)
        assert e3 <= 1e-5 and e4 <= 1e-5, ("wgrad", cin, cout, e3, e4)
        worst[(cin, cout, ps, H)] = (e, e1, e2, e3, e4)
    n_v2 = ops.conv_bf16_v2_launches() - n_v2
    assert (n_v2 == 0) if (impl & 3) == 1 else (n_v2 >= 2 * len(cases) - 6), (impl, n_v2)   # the persistent kernel really ran
    return {"%d->%d ps%d H%d" % k: tuple(round(v, 7) for v in vs) for k, vs in worst.items()}


def check_bf16_encoder_s2d(device, seed=3):
    """The encoder's stride-2 Conv2d / ConvTranspose2d (sftmd_arch.py:745-749) in the stride-1 forms of csrc/s2d.hip
    (space-to-depth image + expanded kernel; expanded kernel + PixelShuffle(2) epilogue) against torch's own strided and
    transposed convolutions (the latter WITHOUT output_padding, chained into the next stride-2 layer as the reference
    chains layer4 into layer5) of the same bf16-rounded operands: forward to 2^-8 of the largest value, and - through the
    tape operators of graph.py - the input gradient (2^-7: two roundings) and the fp32 weight / bias gradients (2e-3: the
    LeakyReLU backward rounds dy * 0.2 to bf16 before the weight gradient sums it; measured 2e-4 .. 3e-4; a misplaced
    kernel slice would be an O(1) error).
    Odd and even frame sizes; the fp32 and the bf16 flavour of the space-to-depth input."""
    from dasr_amd import graph, tape as tp
    gen = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=gen)
    worst = {}
    for (cin, cout, B, H, W, x_bf16) in [(32, 64, 2, 12, 16, False), (64, 128, 1, 7, 9, True), (32, 32, 1, 5, 34, True)]:
        x = _bf(rn(B, cin, H, W))
        w = _bf(rn(cout, cin, 3, 3) * (1.0 / math.sqrt(9 * cin)))
        bias = rn(cout) * 0.1
        xt, wt, bt = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        ref = F.leaky_relu(F.conv2d(xt, wt, bt, stride=2, padding=1), 0.2)
        dy = _bf(rn(*ref.shape))
        gx, gw, gb = torch.autograd.grad(ref, (xt, wt, bt), dy)
        t = tp.Tape(True, BF16)
        xv = tp.Var(nhwc(x).to(device).to(BF16 if x_bf16 else torch.float32), True)
        wv = tp.Var(ops.pack_hwio(w.permute(2, 3, 1, 0).contiguous().to(device)), True)
        bv = tp.Var(bias.to(device), True)
        y = graph.conv(t, graph.space_to_depth2(t, xv), graph.expand_s2(t, wv), bv, act=ops.ACT_LRELU)
        assert y.data.dtype == BF16 and tuple(y.data.shape) == tuple(nhwc(ref).shape), (y.data.shape, ref.shape)
        e = (nchw(y.data.float().cpu()) - ref.detach()).abs().max().item() / ref.detach().abs().max().item()
        y.grad = nhwc(dy).to(device).to(BF16)
        t.backward()
        e1 = (nchw(xv.grad.float().cpu()) - gx).abs().max().item() / gx.abs().max().item()
        e2 = rel_max(wv.grad.permute(3, 2, 0, 1), gw)
        e3 = rel_max(bv.grad, gb)
        assert xv.grad.dtype == xv.data.dtype and wv.grad.dtype == torch.float32
        assert e <= 2.0 ** -8 and e1 <= 2.0 ** -7 and e2 <= 2e-3 and e3 <= 2e-3, ("s2", cin, cout, H, W, e, e1, e2, e3)
        worst["s2 %d->%d %dx%d" % (cin, cout, H, W)] = (e, e1, e2, e3)
    # Encoder layers 4 -> 5 chained, as the reference has them (sftmd_arch.py:748-749): ConvTranspose2d(3, stride=2,
    # padding=1) WITHOUT output_padding, i.e. (2H-1) x (2W-1) outputs, then LeakyReLU, then Conv2d(3, stride=2, padding=1).
    # The PixelShuffle image of the stride-1 form is 2H x 2W: its first (2H-1) x (2W-1) pixels must be the reference's
    # layer, and the next layer (space-to-depth with those VALID extents) must see zero padding in the extra row / column
    # and send no gradient into it.  Reference on the same bf16-rounded operands, the layer-4 image rounded to bf16
    # (straight-through) as the stride-1 path stores it.
    for (cin, cout, B, H, W) in [(128, 32, 1, 4, 5), (32, 64, 2, 6, 8), (64, 32, 1, 3, 3)]:
        x = _bf(rn(B, cin, H, W))
        w = _bf(rn(cin, cout, 3, 3) * (1.0 / math.sqrt(9 * cin)))          # ConvTranspose2d: [Cin][Cout][kh][kw]
        bias = rn(cout) * 0.1
        w5 = _bf(rn(cout, cout, 3, 3) * (1.0 / math.sqrt(9 * cout)))
        b5 = rn(cout) * 0.1
        xt, wt, bt, w5t, b5t = [v.clone().requires_grad_(True) for v in (x, w, bias, w5, b5)]
        ref4 = F.leaky_relu(F.conv_transpose2d(xt, wt, bt, stride=2, padding=1), 0.2)
        assert tuple(ref4.shape[2:]) == (2 * H - 1, 2 * W - 1)
        ref4r = ref4 + (_bf(ref4.detach()) - ref4.detach())
        ref5 = F.conv2d(ref4r, w5t, b5t, stride=2, padding=1)
        assert tuple(ref5.shape[2:]) == (H, W)
        dy = _bf(rn(*ref5.shape))
        gx, gw, gb, gw5, gb5 = torch.autograd.grad(ref5, (xt, wt, bt, w5t, b5t), dy)
        t = tp.Tape(True, BF16)
        xv = tp.Var(nhwc(x).to(device).to(BF16), True)
        # the packed form of a transposed kernel is HWIO of w[ci][co][kh][kw] (ops.weight_pack(transposed=True))
        wv = tp.Var(ops.pack_hwio(w.permute(2, 3, 0, 1).contiguous().to(device)), True)
        bv = tp.Var(bias.to(device), True)
        w5v = tp.Var(ops.pack_hwio(w5.permute(2, 3, 1, 0).contiguous().to(device)), True)
        b5v = tp.Var(b5.to(device), True)
        w4, b4 = graph.expand_t2(t, wv, bv)
        y4 = graph.conv(t, xv, w4, b4, act=ops.ACT_LRELU, ps_r=2)
        assert y4.data.dtype == BF16 and tuple(y4.data.shape) == (B, 2 * H, 2 * W, cout), y4.data.shape
        y5 = graph.conv(t, graph.space_to_depth2(t, y4, (2 * H - 1, 2 * W - 1)), graph.expand_s2(t, w5v), b5v)
        assert tuple(y5.data.shape) == tuple(nhwc(ref5).shape), (y5.data.shape, ref5.shape)
        y4c = nchw(y4.data.float().cpu())[:, :, :2 * H - 1, :2 * W - 1]
        e4 = (y4c - ref4.detach()).abs().max().item() / ref4.detach().abs().max().item()
        e = (nchw(y5.data.float().cpu()) - ref5.detach()).abs().max().item() / ref5.detach().abs().max().item()
        y5.grad = nhwc(dy).to(device).to(BF16)
        t.backward()
        e1 = (nchw(xv.grad.float().cpu()) - gx).abs().max().item() / gx.abs().max().item()
        e2 = max(rel_max(wv.grad.permute(2, 3, 0, 1), gw), rel_max(w5v.grad.permute(3, 2, 0, 1), gw5))
        e3 = max(rel_max(bv.grad, gb), rel_max(b5v.grad, gb5))
        # (measured on the emulator: e4 <= 2.9e-3, e <= 2.5e-3, e1 <= 2.9e-3, e2 <= 2.0e-3, e3 <= 2.5e-3; the extra row /
        # column leaking into layer 5 - the output_padding=1 operator - is an O(1) error in the last output row and column)
        assert e4 <= 2.0 ** -8 and e <= 2.0 ** -7 and e1 <= 2.0 ** -7 and e2 <= 6e-3 and e3 <= 6e-3, \
            ("t2 -> s2", cin, cout, H, W, e4, e, e1, e2, e3)
        worst["t2>s2 %d->%d %dx%d" % (cin, cout, H, W)] = (e4, e, e1, e2, e3)
    return {k: tuple(round(v, 7) for v in vs) for k, vs in worst.items()}


def check_bf16_encoder_paths_agree(device):
    """The two encoder implementations of the bf16 path - stride-1 bf16-MFMA forms (graph.ENCODER_S2D = True, the default)
    and the fp32 gather kernels (False; pinned to the reference by check_encoder_geometry / the whole-net goldens) - must
    compute the same function: e5 (layer5's output) and the region-pooled depth matrix st agree to bf16 rounding on even
    and odd frame sizes.  (The output_padding=1 operator instead of the reference's transposed layer moved e5's last row
    and column by 0.9 of the map's largest value and st by 0.81 in rel-L2.)"""
    from dasr_amd import graph, tape as tp
    out = {}
    for (H, W) in ((16, 20), (17, 21), (12, 28)):
        case = dict(name="enc", which=[0, 1], nb=4, scale=8, L=32, B=2, H=H, W=W)
        net, cfg = build_net(case, device)
        lq, gt, dm, mk = [t.to(device) for t in synth.closed_form_batch(0, 2, H, W, 8)]
        P = {k: tp.Var(p.detach(), False, name=k) for k, p in net.named_parameters()}
        res = {}
        for s2d in (True, False):
            graph.ENCODER_S2D = s2d
            try:
                t = tp.Tape(False, BF16)
                x0 = tp.Var(ops.nchw_to_nhwc(lq))
                e1, e5, st = graph.encoder_forward(t, P, cfg, x0, mk)
            finally:
                graph.ENCODER_S2D = True
            res[s2d] = (e5.data.float().cpu(), st.data.float().cpu())
        e5a, sta = res[True]
        e5b, stb = res[False]
        assert e5a.shape == e5b.shape and sta.shape == stb.shape
        d5 = (e5a - e5b).abs().max().item() / e5b.abs().max().item()
        edge = max((e5a[:, -1] - e5b[:, -1]).abs().max().item(), (e5a[:, :, -1] - e5b[:, :, -1]).abs().max().item()) / \
            e5b.abs().max().item()
        dst = ((sta - stb).pow(2).sum() / stb.pow(2).sum()).sqrt().item()
        # measured (emulator): d5 <= 6e-3, st rel-L2 <= 5e-3
        assert d5 <= 2e-2 and edge <= 2e-2 and dst <= 2e-2, ("encoder paths", H, W, d5, edge, dst)
        out["%dx%d" % (H, W)] = (round(d5, 5), round(edge, 5), round(dst, 5))
    return out


def digest_cosine(named_grads, golden, prefix, skip=()):
    dot = na = nb = 0.0
    for k, gten in named_grads:
        if any(s in k for s in skip):
            continue
        want = np.asarray(golden[prefix + k], dtype=np.float64)
        got = grad_digest(gten.cpu())
        dot += float((got * want).sum())
        na += float((got ** 2).sum())
        nb += float((want ** 2).sum())
    return dot / math.sqrt(max(na * nb, 1e-300))


# Gates of the bf16 whole-net checks, per case: (gradient rel-L2 of the linear functional vs the reference's float64 run,
# gradient rel-L2 of the harness loss vs the reference's fp32 run, minimum cosine similarity of the latter).  The rel-L2
# gates are ~3x the values measured on the emulator and the MI355X (bf16 keeps 8 significant bits per stored activation;
# the test functional's oscillating weights and the L1 loss's sign() make the parameter gradients sums with heavy
# cancellation, so 0.4 % per-element noise becomes 3..22 % of the gradient norm at these tiny frame sizes; two bf16
# implementations with identical rounding points - this one and oracle.bf16_storage() - sit 4.5..7 % apart themselves).
#   measured (emulator): x8_nb4 lin 0.106 loss 0.014 cos 0.99997;  x4_nb4 lin 0.224 loss 0.095 cos 0.9958
# (x2_nb4 - four DGBs whose instance norms see 96 pixels - is too small a frame for 8-bit activations: 37 dB, not gated)
BF16_GATES = {"x8_nb4": (0.32, 0.05, 0.999), "x4_nb4": (0.65, 0.28, 0.985)}
# PSNR of the bf16 image against the reference's fp32 image of the same case (measured 48.3 / 45.7 dB with either encoder
# implementation; the output_padding=1 encoder of round 2 sat at 41.7 / 37.3)
BF16_PSNR_MIN = {"x8_nb4": 45.0, "x4_nb4": 43.0}


def check_bf16_depthnet_case(case, device, dpsnr_tol=0.02):
    """The whole net on the bf16 path (net.set_compute_dtype(torch.bfloat16)) against the reference's golden vectors
    for the same case.  north_star's bar for reduced precision: PSNR within 0.02 dB of the reference's
    (|PSNR(out_bf16, GT) - PSNR(out_ref, GT)| <= 0.02).  Gradients: see BF16_GATES.  The same forward under
    torch.autocast(bfloat16) with the module left at float32 gives the same bits; the kernels themselves are pinned
    bit-for-bit by check_bf16_ops_vs_fp32_kernels / check_bf16_conv_variants."""
    g = load("depthnet_" + case["name"])
    lin_gate, loss_gate, cos_min = BF16_GATES[case["name"]]
    net, cfg = build_net(case, device)
    net.set_compute_dtype(BF16)
    lq, gt, dm, mk = [t.to(device) for t in synth.closed_form_batch(0, case["B"], case["H"], case["W"], cfg["scale"])]
    with torch.no_grad():
        sr0 = net(lq, dm, mk)
    assert sr0.dtype == torch.float32
    ref = torch.from_numpy(g["sr"])
    gt_c = gt.cpu()
    dpsnr = abs(O.psnr_255(sr0.cpu(), gt_c) - O.psnr_255(ref, gt_c))
    psnr_vs_ref = O.psnr_255(sr0.cpu(), ref)
    err = (sr0.cpu() - ref).abs().max().item()
    assert dpsnr <= dpsnr_tol, ("bf16 psnr", case["name"], dpsnr)
    assert psnr_vs_ref >= BF16_PSNR_MIN[case["name"]], psnr_vs_ref
    nograd = set(g["nograd"].tolist())

    def grads():
        named = []
        for k, p in net.named_parameters():
            if k in nograd:
                assert p.grad is None, k
            else:
                assert p.grad is not None and p.grad.dtype == torch.float32, k
                assert bool(torch.isfinite(p.grad).all()), k
                named.append((k, p.grad.detach().clone()))
        return named

    sr = net(lq, dm, mk)
    assert torch.equal(sr.detach(), sr0)
    wgt = torch.cos(torch.arange(sr.numel(), dtype=torch.float32) * 0.013).reshape(sr.shape).to(device)
    (sr * wgt).sum().backward()
    named = grads()
    l64, worst64 = digest_global_rel_l2(named, g, "gl64.", skip=ZERO_GRAD_KEYS)
    assert l64 <= lin_gate, ("bf16 linear-functional grads vs fp64 reference", case["name"], l64, worst64)
    net.zero_grad(set_to_none=True)
    sr = net(lq, dm, mk)
    w = torch.ones(cfg["depthRangeNum"], device=device, requires_grad=True)
    total, l_pix, l_dyn, per = O.total_loss(sr, gt, mk, w)
    total.backward()
    named = grads()
    lloss, _ = digest_global_rel_l2(named, g, "g.", skip=ZERO_GRAD_KEYS)
    cos = digest_cosine(named, g, "g.", skip=ZERO_GRAD_KEYS)
    assert lloss <= loss_gate and cos >= cos_min, ("bf16 loss grads", case["name"], lloss, cos)
    assert abs(l_pix.item() - float(g["l_pix"])) <= 1e-3 and abs(l_dyn.item() - float(g["l_dyn"])) <= 2e-3, \
        ("bf16 loss values", case["name"], l_pix.item() - float(g["l_pix"]), l_dyn.item() - float(g["l_dyn"]))
    if device != "cpu":
        net.set_compute_dtype(torch.float32)
        with torch.no_grad(), torch.autocast("cuda", dtype=BF16):
            sr_ac = net(lq, dm, mk)
        assert torch.equal(sr_ac, sr0), "autocast opt-in differs from set_compute_dtype"
    return dict(dpsnr=dpsnr, psnr_vs_ref=psnr_vs_ref, max_err=err, lin64_l2=l64, loss_l2=lloss, loss_cos=cos,
                l_pix_err=abs(l_pix.item() - float(g["l_pix"])), l_dyn_err=abs(l_dyn.item() - float(g["l_dyn"])))


def check_bf16_ops_vs_fp32_kernels(device, seed=1):
    """Every templated kernel of the bf16 path against its own fp32 instantiation (already pinned to the reference by
    the golden tests) on IDENTICAL bf16-valued inputs.  Both compute in fp32, so a bf16 output must be exactly the
    rounding of the fp32 kernel's output, and fp32 outputs (statistics, dD, parameter gradients) must agree to
    summation-order noise: SEAN forward / backward (one-hot and soft masks), instance-norm statistics, the mask
    convolution and its fused weight gradient, the 9x9 output convolution (fwd / dgrad / wgrad), the epilogue
    backward (activation, PixelShuffle), add / accumulate / casts."""
    gen = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=gen)
    dev = lambda t: t.to(device)
    h = lambda t: dev(t).to(BF16)
    out = {}
    # ---- SEAN
    B, H, W, C, K = 2, 9, 33, 64, 10
    t, gb2, res = _bf(rn(B, H, W, C)), _bf(rn(B, H, W, 2 * C)), _bf(rn(B, H, W, C))
    _, _, _, mk = synth.closed_form_batch(1, B, H, W, 1, K)
    D = rn(B, 2, 9, K, C) * 0.1
    bg, bb = rn(C) * 0.1, rn(C) * 0.1
    ag, ab = torch.full((1,), 0.7), torch.full((1,), 0.74)
    dout = _bf(rn(B, H, W, C))
    for soft in (False, True):
        mask = dev(mk if not soft else (0.7 * mk + 0.3 * torch.rand(mk.shape, generator=gen)))
        region, flag = ops.mask_compress(mask)
        mean, var = ops.instnorm_stats(dev(t))
        m16, v16 = ops.instnorm_stats(h(t))
        assert torch.equal(mean, m16) and torch.equal(var, v16), "instnorm statistics of identical values differ"
        for r in (None, res):
            a32 = (dev(t), mean, var, dev(gb2), mask, region, flag, dev(D), dev(bg), dev(bb), dev(ag), dev(ab),
                   dev(r) if r is not None else None, True)
            a16 = (h(t), mean, var, h(gb2), mask, region, flag, dev(D), dev(bg), dev(bb), dev(ag), dev(ab),
                   h(r) if r is not None else None, True)
            y32, y16 = ops.sean_fwd(*a32), ops.sean_fwd(*a16)
            assert y16.dtype == BF16 and torch.equal(y16, y32.to(BF16)), ("sean fwd", soft, r is not None)
            ops.set_conv_bf16_impl(2048)            # the 4-channels-per-lane form of the bf16 gather kernel (default: 8)
            try:
                y16b = ops.sean_fwd(*a16)
            finally:
                ops.set_conv_bf16_impl(0)
            assert torch.equal(y16b, y16), ("sean fwd, 4 channels per lane", soft, r is not None)
        # backward on the bf16-valued forward output
        y = y16.float()
        g32 = ops.sean_bwd(dev(dout), y, dev(t), mean, var, dev(gb2), mask, region, flag, dev(D), dev(bg), dev(bb), dev(ag),
                           dev(ab), True, True)
        g16 = ops.sean_bwd(h(dout), y16, h(t), mean, var, h(gb2), mask, region, flag, dev(D), dev(bg), dev(bb), dev(ag),
                           dev(ab), True, True)
        names = ("dt", "dgb2", "dD", "dbg", "dbb", "dag", "dab", "dres")
        for nm, a, b in zip(names, g32, g16):
            if nm in ("dgb2", "dres"):
                assert b.dtype == BF16 and torch.equal(b, a.to(BF16)), ("sean bwd", nm, soft)
            elif nm == "dt":      # pass B re-reads the (rounded) intermediate it stored: one extra rounding
                e = (b.float() - a).abs().max().item() / a.abs().max().item()
                assert b.dtype == BF16 and e <= 2.0 ** -7, ("sean bwd dt", soft, e)
                out["sean_dt_soft" if soft else "sean_dt"] = e
            elif nm == "dD" and not soft:
                # the one-hot bf16 kernel sums G on the bf16 matrix cores: each term carries G's bf16 rounding (2^-9
                # relative to the term), the one-hot factor and the fp32 accumulation are exact; on this random-sign
                # test data the sums cancel to ~sqrt(N) terms, so the error relative to the largest sum is ~2^-9 too
                # (measured 2.4e-3)
                assert b.dtype == torch.float32 and rel_max(b, a) <= 2.0 ** -7, ("sean bwd dD", rel_max(b, a))
                out["sean_dD_onehot"] = rel_max(b, a)
            else:
                assert b.dtype == torch.float32 and rel_max(b, a) <= 1e-5, ("sean bwd", nm, soft, rel_max(b, a))
    # ---- mask convolution (depth map fp32 -> 2C channels) and its fused weight gradient
    Bc, Hc, Wc, Co = 2, 7, 19, 128
    depth = dev(rn(Bc, Hc, Wc, 1))
    wm = ops.pack_hwio(dev(rn(3, 3, 1, Co) * 0.3))
    bm = dev(rn(Co) * 0.1)
    a32 = ops.conv2d_fwd(depth, wm, bm, act=ops.ACT_RELU)
    a16 = ops.conv2d_fwd(depth, wm, bm, act=ops.ACT_RELU, out_dtype=BF16)
    assert a16.dtype == BF16 and torch.equal(a16, a32.to(BF16))
    dy = _bf(rn(Bc, Hc, Wc, Co))
    w32 = ops.conv2d_wgrad_act(depth, dev(dy), a16.float(), (3, 3, 1, Co), ops.ACT_RELU)
    w16 = ops.conv2d_wgrad_act(depth, h(dy), a16, (3, 3, 1, Co), ops.ACT_RELU)
    assert rel_max(w16[0], w32[0]) <= 1e-5 and rel_max(w16[1], w32[1]) <= 1e-5
    # ---- 9x9 output convolution on the bf16 matrix cores: bf16 x, the fp32 kernel and dy rounded to bf16 as MFMA
    # operands, fp32 accumulation, fp32 y / dw / db.  Reference: torch's fp32 convolution of the SAME rounded operands, so
    # only the accumulation order differs (and the final rounding of the bf16 dx); ragged tile rows / columns, two batches
    # (the forward / dgrad workgroups are persistent: the second pass caps the grid at 3 workgroups - impl 2 - so that each
    # walks several tiles, incl. across the batch boundary, with the next tile's loads in flight)
    for (B9, H9, W9, impl9) in ((1, 11, 70, 0), (2, 19, 60, 0), (2, 19, 60, 2)):
        ops.set_conv_bf16_impl(impl9)
        x9 = _bf(rn(B9, H9, W9, 32))
        w9f = rn(9, 9, 32, 3) * 0.02
        w9 = ops.pack_hwio(dev(w9f))
        b9 = rn(3)
        xt = nchw(x9).clone().requires_grad_(True)
        wt = _bf(w9f).permute(3, 2, 0, 1).contiguous().requires_grad_(True)          # OIHW of the rounded kernel
        ref = F.conv2d(xt, wt, b9, padding=4)
        y16 = ops.conv2d_fwd(h(x9), w9, dev(b9), pad=4)
        assert y16.dtype == torch.float32 and tuple(y16.shape) == (B9, H9, W9, 3)
        e = rel_max(nchw(y16), ref.detach())
        assert e <= 2e-6, ("conv9 bf16 fwd", e)
        dy9 = rn(B9, H9, W9, 3)
        gx, gw = torch.autograd.grad(F.conv2d(xt, wt, None, padding=4), (xt, wt), nchw(_bf(dy9)))
        dx16 = ops.conv2d_dgrad(dev(dy9), w9, x9.shape, pad=4, out_dtype=BF16)
        e1 = (nchw(dx16.float().cpu()) - gx).abs().max().item() / gx.abs().max().item()
        assert dx16.dtype == BF16 and e1 <= 2.0 ** -8, ("conv9 bf16 dgrad", e1)
        acc = h(x9).clone()
        ops.conv2d_dgrad(dev(dy9), w9, x9.shape, pad=4, out=acc)
        want = nhwc(gx) + x9
        e2 = (acc.float().cpu() - want).abs().max().item() / want.abs().max().item()
        assert e2 <= 2.0 ** -7, ("conv9 bf16 dgrad accumulate", e2)
        dw16, db16 = ops.conv2d_wgrad(h(x9), dev(dy9), (9, 9, 32, 3), pad=4)
        e3 = rel_max(dw16.permute(3, 2, 0, 1), gw)
        e4 = rel_max(db16, dy9.sum((0, 1, 2)))                  # the bias gradient sums the fp32 dy
        assert e3 <= 1e-5 and e4 <= 1e-5, ("conv9 bf16 wgrad", e3, e4)
        out["conv9_%dx%d_impl%d" % (H9, W9, impl9)] = (e, e1, e2, e3)
    ops.set_conv_bf16_impl(0)
    # ---- epilogue backward (activation and PixelShuffle), add, accumulate, casts
    for act, ps in ((1, 1), (2, 2), (2, 3)):
        Cq = 8
        yv, dyv = _bf(rn(1, 6 * ps, 5 * ps, Cq)), _bf(rn(1, 6 * ps, 5 * ps, Cq))
        e32 = ops.conv2d_epilogue_bwd(dev(dyv), dev(yv), 6, 5, Cq * ps * ps, act, ps)
        e16 = ops.conv2d_epilogue_bwd(h(dyv), h(yv), 6, 5, Cq * ps * ps, act, ps)
        assert e16.dtype == BF16 and torch.equal(e16, e32.to(BF16)), ("epilogue bwd", act, ps)
    a, b = _bf(rn(3, 5, 8)), _bf(rn(3, 5, 8))
    assert torch.equal(ops.add(h(a), h(b)), (a + b).to(BF16).to(device))
    acc = h(a).clone()
    ops.accumulate_(acc, h(b))
    assert torch.equal(acc, (a + b).to(BF16).to(device))
    f = dev(rn(3, 5, 8))
    want = f + dev(b)
    ops.accumulate_(f, h(b))                                   # bf16 gradient into an fp32 tensor
    assert torch.equal(f, want)
    assert torch.equal(ops.cast_to_bf16(dev(a + 0.001)), (a + 0.001).to(BF16).to(device))
    assert torch.equal(ops.cast_to_f32(h(a)), dev(a))
    cp = torch.empty_like(h(a))
    assert torch.equal(ops.copy_(cp, h(a)), h(a))
    return out


def check_bf16_c3_full_frame(device="cuda", scale=4, H=256, W=320):
    """BASELINE.json configs[2] at its full frame size (x4 net, nb=16, L=256, DGBs 0..13, one 256x320 LR frame ->
    1024x1280) on the bf16 path - and, with scale=8 / 128x160, configs[3]'s per-GPU network - against the fp32 CPU oracle
    (forward: north_star's reduced-precision bar, |PSNR(out_bf16, GT) - PSNR(out_oracle, GT)| <= 0.02 dB) and against
    this repo's fp32 HIP path on the same weights (harness-loss gradients: relative L2 and cosine, gated at ~3x / well
    below the values measured on the MI355X)."""
    case = dict(name="c3", scale=scale, which=list(range(14)), L=256, nb=16, B=1, H=H, W=W)
    net, cfg = build_net(case, device)
    lq, gt, dm, mk = synth.closed_form_batch(0, 1, case["H"], case["W"], scale)
    sd = _oracle_sd(net)
    with torch.no_grad():
        ref = O.depthnet_forward(sd, cfg, lq, dm, mk)
    # CPU model of the same rounding points (oracle.bf16_storage, test infrastructure): its image AND its harness-loss
    # gradients - the bf16 kernels are gated against the model of what they are supposed to compute, not only against this
    # repo's own fp32 run
    sdm = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    wm = torch.ones(10, requires_grad=True)
    with O.bf16_storage():
        ref_model_g = O.depthnet_forward(sdm, cfg, lq, dm, mk)
        O.total_loss(ref_model_g, gt, mk, wm)[0].backward()
    ref_model = ref_model_g.detach()
    model_grads = {k: v.grad for k, v in sdm.items() if v.grad is not None}
    psnr_model = O.psnr_255(ref_model, ref)
    lqd, gtd, dmd, mkd = [t.to(device) for t in (lq, gt, dm, mk)]
    res = {}
    for dt in (torch.float32, BF16):
        net.set_compute_dtype(dt)
        net.zero_grad(set_to_none=True)
        sr = net(lqd, dmd, mkd)
        w = torch.ones(10, device=device, requires_grad=True)
        total, l_pix, l_dyn, _ = O.total_loss(sr, gtd, mkd, w)
        total.backward()
        res[dt] = (sr.detach().cpu(), float(l_pix), float(l_dyn),
                   {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
    sr32, sr16 = res[torch.float32][0], res[BF16][0]
    assert tuple(sr16.shape) == (1, 3, 1024, 1280)
    dpsnr32 = abs(O.psnr_255(sr32, gt) - O.psnr_255(ref, gt))
    dpsnr16 = abs(O.psnr_255(sr16, gt) - O.psnr_255(ref, gt))
    psnr16 = O.psnr_255(sr16, ref)
    num = den = dot = nb = 0.0
    for k, a in res[torch.float32][3].items():
        if any(z in k for z in ZERO_GRAD_KEYS):
            continue
        b = res[BF16][3][k]
        assert bool(torch.isfinite(b).all()), k
        a, b = a.double(), b.double()
        num += (a - b).pow(2).sum().item()
        den += a.pow(2).sum().item()
        dot += (a * b).sum().item()
        nb += b.pow(2).sum().item()
    rel, cos = math.sqrt(num / den), dot / math.sqrt(den * nb)
    # HIP bf16 gradients against the model's (and, for scale, the model's against the fp32 HIP run)
    def _dist(ga, gb):
        n_ = d_ = t_ = m_ = 0.0
        for k, a in ga.items():
            if any(z in k for z in ZERO_GRAD_KEYS) or k not in gb:
                continue
            a, b = a.double().cpu(), gb[k].double().cpu()
            n_ += (a - b).pow(2).sum().item(); d_ += a.pow(2).sum().item()
            t_ += (a * b).sum().item(); m_ += b.pow(2).sum().item()
        return math.sqrt(n_ / d_), t_ / math.sqrt(d_ * m_)
    rel_m, cos_m = _dist(model_grads, res[BF16][3])
    rel_mf, cos_mf = _dist(res[torch.float32][3], model_grads)
    out = dict(grad_hip_bf16_vs_model=(rel_m, cos_m), grad_model_vs_fp32=(rel_mf, cos_mf), dpsnr_fp32=dpsnr32, dpsnr_bf16=dpsnr16, psnr_bf16_vs_oracle=psnr16, psnr_cpu_bf16_model_vs_oracle=psnr_model,
               max_err_bf16=(sr16 - ref).abs().max().item(), loss_grad_rel_l2_bf16_vs_fp32=rel, loss_grad_cosine=cos,
               l_pix=(res[torch.float32][1], res[BF16][1]), l_dyn=(res[torch.float32][2], res[BF16][2]))
    print("bf16 full frame x%d %dx%d:" % (scale, H, W), out)
    assert dpsnr32 <= 1e-3, dpsnr32                 # the fp32 path: north_star's fp32 bar
    assert dpsnr16 <= 0.02, dpsnr16                 # the bf16 path: north_star's reduced-precision bar
    # 13 DGBs deep, every stored activation carries 8 significant bits and each block's two instance norms re-amplify
    # the noise of the one before: measured 32.3 dB against the fp32 oracle's image on the MI355X (45 .. 48 dB for the
    # two-block golden cases); the floor only catches a broken kernel
    assert psnr16 >= 28.0, psnr16
    # ... and it is the price of the storage format, not of these kernels: a CPU restatement that rounds at the same
    # points (oracle.bf16_storage) lands within 3 dB of the HIP path's distance to the fp32 image
    assert psnr16 >= psnr_model - 3.0, (psnr16, psnr_model)
    assert rel <= BF16_C3_GRAD_GATE[0] and cos >= BF16_C3_GRAD_GATE[1], (rel, cos)
    # against the CPU model of the same rounding points: measured on the MI355X rel-L2 0.226 / cosine 0.9747 at c3's frame,
    # 0.115 / 0.9934 at c4's - CLOSER than the model itself sits to the fp32 gradient (0.271 / 0.9626 and 0.145 / 0.9897:
    # thirteen double-instance-norm blocks amplify any two roundings apart); gates ~2.2x the measured distances
    gate_m = (0.5, 0.94) if scale == 4 else (0.3, 0.98)
    assert rel_m <= gate_m[0] and cos_m >= gate_m[1], ("bf16 gradients vs the bf16 model", rel_m, cos_m)
    return out


# harness-loss gradients of the bf16 path vs this repo's fp32 path at c3's frame size: (max rel-L2, min cosine); set from
# the values measured on the MI355X (profiles/r02_gpu_tests.log)
BF16_C3_GRAD_GATE = (0.6, 0.9)
