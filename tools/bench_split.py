#!/usr/bin/env python3
"""fp32 3x3 convolutions at the x8 bench shapes: exact-fp32 MFMA kernels vs the split-bf16 kernels (forward, dgrad).
HIP events, isolated, interleaved rounds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dasr_amd  # noqa
from dasr_amd import ops


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    H0, W0 = 128, 160
    shapes = ((128, 128, 1), (64, 64, 1), (64, 256, 1), (64, 32, 2), (32, 128, 2), (32, 32, 4), (32, 128, 4))
    for ci, co, up in shapes:
        H, W = H0 * up, W0 * up
        x = torch.randn(B, H, W, ci, device=dev)
        wp = ops.pack_hwio(torch.randn(3, 3, ci, co, device=dev) * 0.05)
        bias = torch.randn(co, device=dev)
        ws = ops.conv3x3_split_weights(wp)
        ws2 = ops.conv3x3_split2_weights(wp)
        y = ops.conv2d_fwd(x, wp, bias)
        xm, ym = ops.absmax(x), ops.absmax(y)
        fl = 2.0 * 9 * ci * co * B * H * W
        keys = ("f32 fwd", "bf16x3 fwd", "fp16x2 fwd", "f32 dgrad", "bf16x3 dgrad", "fp16x2 dgrad", "f32 wgrad", "bf16x3 wgrad",
                "fp16x2 wgrad", "fp16x2v1 wgrad", "fp16x2alt fwd", "fp16x2alt dgrad", "fp16x2h fwd", "fp16x2h dgrad", "bf16x3 weights", "fp16x2 weights", "absmax x", "absmax y")
        r = {k: [] for k in keys}
        for _ in range(3):
            r["f32 fwd"].append(timeit(lambda: ops.conv2d_fwd(x, wp, bias)))
            r["bf16x3 fwd"].append(timeit(lambda: ops.conv3x3_fwd_split(x, ws, bias, co)))
            r["fp16x2 fwd"].append(timeit(lambda: ops.conv3x3_fwd_split2(x, xm, ws2, bias, co)))
            r["f32 dgrad"].append(timeit(lambda: ops.conv2d_dgrad(y, wp, x.shape)))
            r["bf16x3 dgrad"].append(timeit(lambda: ops.conv3x3_dgrad_split(y, ws, x.shape)))
            r["fp16x2 dgrad"].append(timeit(lambda: ops.conv3x3_dgrad_split2(y, ym, ws2, x.shape)))
            r["f32 wgrad"].append(timeit(lambda: ops.conv2d_wgrad(x, y, (3, 3, ci, co))))
            r["bf16x3 wgrad"].append(timeit(lambda: ops.conv3x3_wgrad_split(x, y)))
            r["fp16x2 wgrad"].append(timeit(lambda: ops.conv3x3_wgrad_split2(x, xm, y, ym)))
            ops.set_conv_bf16_impl(256 + 512)   # the four-wave form (128 produced channels) / the chunk form (64) of the fp16 x 2 kernel
            r["fp16x2alt fwd"].append(timeit(lambda: ops.conv3x3_fwd_split2(x, xm, ws2, bias, co)))
            r["fp16x2alt dgrad"].append(timeit(lambda: ops.conv3x3_dgrad_split2(y, ym, ws2, x.shape)))
            ops.set_conv_bf16_impl(1024)        # the other eight-wave form (128 produced channels: kernel-row steps; 64: channel halves)
            r["fp16x2h fwd"].append(timeit(lambda: ops.conv3x3_fwd_split2(x, xm, ws2, bias, co)))
            r["fp16x2h dgrad"].append(timeit(lambda: ops.conv3x3_dgrad_split2(y, ym, ws2, x.shape)))
            ops.set_conv_bf16_impl(64)      # the first fp16 x 2 weight-gradient kernel (operands split per K-step)
            r["fp16x2v1 wgrad"].append(timeit(lambda: ops.conv3x3_wgrad_split2(x, xm, y, ym)))
            ops.set_conv_bf16_impl(0)
            r["bf16x3 weights"].append(timeit(lambda: ops.conv3x3_split_weights(wp)))
            r["fp16x2 weights"].append(timeit(lambda: ops.conv3x3_split2_weights(wp)))
            r["absmax x"].append(timeit(lambda: ops.absmax(x)))
            r["absmax y"].append(timeit(lambda: ops.absmax(y)))
        for k, v in r.items():
            us = sorted(v)[1]
            if "absmax" in k:
                n = (x if k.endswith("x") else y).numel() * 4
                print("B=%d %3d->%3d @%dx%d %-15s %8.1f us  %6.2f TB/s" % (B, ci, co, H, W, k, us, n / us / 1e6))
                continue
            nprod = 6 if "bf16x3" in k else (3 if "fp16x2" in k else 0)
            print("B=%d %3d->%3d @%dx%d %-15s %8.1f us  %6.1f TF (fp32-equivalent)  %7.1f TF of 16-bit MFMA work"
                  % (B, ci, co, H, W, k, us, fl / us / 1e6, nprod * fl / us / 1e6))
        del x, y


if __name__ == "__main__":
    main()
