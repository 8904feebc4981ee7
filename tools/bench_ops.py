#!/usr/bin/env python3
"""Per-op timing on the GPU (HIP events on the launch stream): SEAN fwd/bwd and the hot conv shapes.
Usage: python tools/bench_ops.py [--batch 16] [--iters 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dasr_amd  # noqa
from dasr_amd import ops, synth


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
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--amax", action="store_true", help="SEAN forward: the instantiation that also leaves max |out| behind")
    a = ap.parse_args()
    dev = torch.device("cuda")
    B, H, W, C, K = a.batch, 128, 160, 64, 10
    torch.manual_seed(0)
    res = {}
    if not a.only or "sean" in a.only:
        t = torch.randn(B, H, W, C, device=dev)
        gb2 = torch.randn(B, H, W, 2 * C, device=dev)
        _, _, _, mk = synth.seeded_batch(0, B, H, W, 1, K)
        mk = mk.to(dev)
        region, flag = ops.mask_compress(mk)
        flag = None if int(flag.item()) == 0 else flag   # what graph.MaskPack does
        D = torch.randn(B, 2, 9, K, C, device=dev) * 0.1
        bg, bb = torch.randn(C, device=dev), torch.randn(C, device=dev)
        ag, ab = torch.full((1,), 0.7, device=dev), torch.full((1,), 0.74, device=dev)
        resid = torch.randn(B, H, W, C, device=dev)
        mean, var = ops.instnorm_stats(t)
        px = B * H * W
        us = timeit(lambda: ops.instnorm_stats(t), a.iters)
        print("instnorm_stats            %8.1f us  %7.1f GB/s" % (us, px * C * 4 / us / 1e3))
        for name, r in (("sean_fwd", None), ("sean_fwd+res", resid)):
            am = ops.amax_buffer(t) if a.amax else None      # (--amax: the instantiation that also keeps max |out|)
            us = timeit(lambda: ops.sean_fwd(t, mean, var, gb2, mk, region, flag, D, bg, bb, ag, ab, r, True, amax=am), a.iters)
            nbytes = px * (4 * (4 * C + (C if r is not None else 0)) + 4 * K)
            print("%-25s %8.1f us  %7.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (name, us, nbytes / us / 1e3,
                                                                                 nbytes / us / 1e3 / 80))
        us = timeit(lambda: ops.sean_fwd(t, mean, var, gb2, mk, None, None, D, bg, bb, ag, ab, None, True), a.iters)
        print("sean_fwd general path     %8.1f us" % us)
        out = ops.sean_fwd(t, mean, var, gb2, mk, region, flag, D, bg, bb, ag, ab, resid, True)
        dout = torch.randn_like(out)
        us = timeit(lambda: ops.sean_bwd(dout, out, t, mean, var, gb2, mk, region, flag, D, bg, bb, ag, ab, True, True),
                    a.iters)
        print("sean_bwd (fast)           %8.1f us  %7.1f GB/s (12C floats/px)" % (us, px * 12 * C * 4 / us / 1e3))
    if not a.only or "dynk" in a.only:
        L = 256
        st = torch.randn(B, K, L, device=dev)
        A_w, A_b = torch.randn(K, K, 1, 1, device=dev) * 0.3, torch.randn(K, device=dev) * 0.1
        Wg, Wb = torch.randn(C, L, 3, 3, device=dev) * 0.05, torch.randn(C, L, 3, 3, device=dev) * 0.05
        stp, Dk = ops.dynk_fwd(st, A_w, A_b, Wg, Wb)
        dD = torch.randn_like(Dk)
        dst = torch.zeros_like(st)
        us = timeit(lambda: ops.dynk_fwd(st, A_w, A_b, Wg, Wb), a.iters)
        print("dynk_fwd (stp + D)        %8.1f us" % us)
        us = timeit(lambda: ops.dynk_bwd(dD, st, stp, A_w, Wg, Wb, dst), a.iters)
        print("dynk_bwd (dW, dstp, dA+dst: 3 launches) %8.1f us" % us)
    if not a.only or "c1" in a.only:
        px = B * H * W
        depth = torch.randn(B, H, W, 1, device=dev)
        wm = ops.pack_hwio(torch.randn(3, 3, 1, 128, device=dev) * 0.3)
        bm = torch.randn(128, device=dev)
        us = timeit(lambda: ops.conv2d_fwd(depth, wm, bm, act=1), a.iters)
        print("mask conv 1->128 fwd      %8.1f us  %7.1f GB/s" % (us, px * 128 * 4 / us / 1e3))
        y = ops.conv2d_fwd(depth, wm, bm, act=1)
        dy = torch.randn_like(y)
        us = timeit(lambda: ops.conv2d_wgrad_act(depth, dy, y, (3, 3, 1, 128), 1), a.iters)
        print("mask conv wgrad           %8.1f us  %7.1f GB/s" % (us, px * 128 * 4 * 2 / us / 1e3))
        del y, dy
    if not a.only or "conv" in a.only:
        shapes = [(128, 160, 64, 64), (128, 160, 128, 128), (128, 160, 32, 64), (128, 160, 64, 256),
                  (256, 320, 64, 32), (256, 320, 32, 32), (256, 320, 32, 128), (512, 640, 32, 32), (512, 640, 32, 128)]
        for (h, w, ci, co) in shapes:
            x = torch.randn(B, h, w, ci, device=dev)
            wt = ops.pack_hwio(torch.randn(3, 3, ci, co, device=dev) * 0.05)
            bias = torch.randn(co, device=dev)
            y = ops.conv2d_fwd(x, wt, bias)
            fl = 2.0 * 9 * ci * co * B * h * w
            us = timeit(lambda: ops.conv2d_fwd(x, wt, bias), a.iters)
            us_d = timeit(lambda: ops.conv2d_dgrad(y, wt, x.shape), a.iters)
            us_w = timeit(lambda: ops.conv2d_wgrad(x, y, tuple(wt.shape[1:])), a.iters)
            print("conv3x3 %3dx%3d %3d->%3d  fwd %8.1f us %6.1f TF | dgrad %8.1f us %6.1f TF | wgrad(+bias) %8.1f us %6.1f TF"
                  % (h, w, ci, co, us, fl / us / 1e6, us_d, fl / us_d / 1e6, us_w, fl / us_w / 1e6))
            del x, y
        x = torch.randn(B, 1024, 1280, 32, device=dev)
        wt = ops.pack_hwio(torch.randn(9, 9, 32, 3, device=dev) * 0.02)
        bias = torch.randn(3, device=dev)
        y = ops.conv2d_fwd(x, wt, bias, pad=4)
        fl = 2.0 * 81 * 32 * 3 * B * 1024 * 1280
        us = timeit(lambda: ops.conv2d_fwd(x, wt, bias, pad=4), a.iters)
        us_d = timeit(lambda: ops.conv2d_dgrad(y, wt, x.shape, pad=4), a.iters)
        us_w = timeit(lambda: ops.conv2d_wgrad(x, y, tuple(wt.shape[1:]), pad=4), a.iters)
        print("conv9x9 1024x1280 32->3   fwd %8.1f us %6.1f TF | dgrad %8.1f us %6.1f TF | wgrad(+bias) %8.1f us %6.1f TF"
              % (us, fl / us / 1e6, us_d, fl / us_d / 1e6, us_w, fl / us_w / 1e6))


if __name__ == "__main__":
    main()
