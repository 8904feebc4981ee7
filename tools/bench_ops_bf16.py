#!/usr/bin/env python3
"""Per-op timing of the bf16 path at BASELINE.json configs[2] shapes (x4, B=32, 256x320 LR), HIP events on the launch
stream, isolated (nothing co-running).  Usage: python tools/bench_ops_bf16.py [--batch 32] [--iters 5] [--only sean,conv]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dasr_amd  # noqa
from dasr_amd import ops, prep, synth

BF = torch.bfloat16


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--hw", default="256x320")
    ap.add_argument("--impl", default="1,16,32", help="conv3x3 forward / dgrad kernels to compare (ops.set_conv_bf16_impl "
                    "codes, comma separated): 1 first kernel, 0 persistent (auto), 16 its 4-wave form, 32 its 8-wave form")
    a = ap.parse_args()
    dev = torch.device("cuda")
    H, W = (int(v) for v in a.hw.split("x"))
    B, C, K = a.batch, 64, 10
    px = B * H * W
    torch.manual_seed(0)
    rb = lambda *s: torch.randn(*s, device=dev).to(BF)
    if not a.only or "sean" in a.only:
        t, gb2, resid = rb(B, H, W, C), rb(B, H, W, 2 * C), rb(B, H, W, C)
        dm = torch.rand(B, 1, H, W, device=dev) * 9.99 + 0.01
        mk = prep.depth_to_masks(dm, K)
        region = mk._dasr_region
        D = torch.randn(B, 2, 9, K, C, device=dev) * 0.1
        bg, bb = torch.randn(C, device=dev), torch.randn(C, device=dev)
        ag, ab = torch.full((1,), 0.7, device=dev), torch.full((1,), 0.74, device=dev)
        us = timeit(lambda: ops.instnorm_stats(t), a.iters)
        print("instnorm_stats bf16       %8.1f us  %7.1f GB/s" % (us, px * C * 2 / us / 1e3))
        mean, var = ops.instnorm_stats(t)
        for impl, tag in ((0, ""), (2048, " 4ch/lane")):        # default: 8 channels (16 bytes) per lane; + 2048: the first form
            ops.set_conv_bf16_impl(impl)
            for name, r in (("sean_fwd bf16", None), ("sean_fwd bf16 +res", resid)):
                us = timeit(lambda: ops.sean_fwd(t, mean, var, gb2, mk, region, None, D, bg, bb, ag, ab, r, True), a.iters)
                nbytes = px * (2 * (4 * C + (C if r is not None else 0)) + 1)
                print("%-25s %8.1f us  %7.1f GB/s (%.1f%% of 8 TB/s)" % (name + tag, us, nbytes / us / 1e3, nbytes / us / 1e3 / 80))
        ops.set_conv_bf16_impl(0)
        out = ops.sean_fwd(t, mean, var, gb2, mk, region, None, D, bg, bb, ag, ab, resid, True)
        dout = rb(B, H, W, C)
        us = timeit(lambda: ops.sean_bwd(dout, out, t, mean, var, gb2, mk, region, None, D, bg, bb, ag, ab, True, True),
                    a.iters)
        print("sean_bwd bf16             %8.1f us  %7.1f GB/s (12C bf16/px)" % (us, px * 12 * C * 2 / us / 1e3))
        del t, gb2, resid, out, dout, mk
    if not a.only or "c1" in a.only:
        depth = torch.randn(B, H, W, 1, device=dev)
        wm = ops.pack_hwio(torch.randn(3, 3, 1, 128, device=dev) * 0.3)
        bm = torch.randn(128, device=dev)
        us = timeit(lambda: ops.conv2d_fwd(depth, wm, bm, act=1, out_dtype=BF), a.iters)
        print("mask conv 1->128 fwd bf16 %8.1f us  %7.1f GB/s" % (us, px * 128 * 2 / us / 1e3))
        y = ops.conv2d_fwd(depth, wm, bm, act=1, out_dtype=BF)
        dy = rb(B, H, W, 128)
        us = timeit(lambda: ops.conv2d_wgrad_act(depth, dy, y, (3, 3, 1, 128), 1), a.iters)
        print("mask conv wgrad bf16      %8.1f us  %7.1f GB/s" % (us, px * 128 * 2 * 2 / us / 1e3))
        del y, dy
    if not a.only or "conv" in a.only:
        shapes = [(H, W, 64, 64), (H, W, 128, 128), (H, W, 32, 64), (H, W, 64, 128), (2 * H, 2 * W, 32, 32),
                  (2 * H, 2 * W, 32, 128)]
        for (h, w, ci, co) in shapes:
            x = rb(B, h, w, ci)
            wt = ops.pack_hwio((torch.randn(3, 3, ci, co, device=dev) * 0.05).to(BF))
            bias = torch.randn(co, device=dev)
            y = ops.conv2d_fwd(x, wt, bias)
            fl = 2.0 * 9 * ci * co * B * h * w
            nb = B * h * w * (ci + co) * 2
            us_w = timeit(lambda: ops.conv2d_wgrad(x, y, tuple(wt.shape[1:])), a.iters)
            impls = tuple(int(v) for v in a.impl.split(","))
            res = {i: [] for i in impls}
            for rnd in range(3):                  # interleaved rounds in one process (cdna_hip_programming.md rule 24)
                for i in impls:
                    ops.set_conv_bf16_impl(i)
                    res[i].append((timeit(lambda: ops.conv2d_fwd(x, wt, bias), a.iters),
                                   timeit(lambda: ops.conv2d_dgrad(y, wt, x.shape, out_dtype=BF), a.iters)))
            ops.set_conv_bf16_impl(0)
            for i in impls:
                us = sorted(r[0] for r in res[i])[1]
                us_d = sorted(r[1] for r in res[i])[1]
                print("conv3x3 bf16 %3dx%3d %3d->%3d impl %2d fwd %8.1f us %6.1f TF %5.2f TB/s | dgrad %8.1f us %6.1f TF | wgrad %8.1f us %6.1f TF"
                      % (h, w, ci, co, i, us, fl / us / 1e6, nb / us / 1e6, us_d, fl / us_d / 1e6, us_w, fl / us_w / 1e6))
            del x, y
    if not a.only or "conv9" in a.only:
        x = rb(B, 4 * H, 4 * W, 32)
        wt = ops.pack_hwio(torch.randn(9, 9, 32, 3, device=dev) * 0.02)
        bias = torch.randn(3, device=dev)
        y = ops.conv2d_fwd(x, wt, bias, pad=4)
        fl = 2.0 * 81 * 32 * 3 * B * 16 * H * W
        us = timeit(lambda: ops.conv2d_fwd(x, wt, bias, pad=4), a.iters)
        us_d = timeit(lambda: ops.conv2d_dgrad(y, wt, x.shape, pad=4, out_dtype=BF), a.iters)
        us_w = timeit(lambda: ops.conv2d_wgrad(x, y, tuple(wt.shape[1:]), pad=4), a.iters)
        print("conv9x9 (bf16 x) 32->3    fwd %8.1f us %6.1f TF | dgrad %8.1f us %6.1f TF | wgrad %8.1f us %6.1f TF"
              % (us, fl / us / 1e6, us_d, fl / us_d / 1e6, us_w, fl / us_w / 1e6))


if __name__ == "__main__":
    main()
