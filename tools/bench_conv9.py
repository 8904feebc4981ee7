#!/usr/bin/env python3
"""The 9x9 output convolution at the x8 bench shape (B frames of 1024 x 1280, 32 -> 3): exact-fp32 MFMA kernels vs the
fp16 x 2 split kernels.  HIP events, isolated, median of 3 rounds of 3."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dasr_amd  # noqa
from dasr_amd import ops


def timeit(fn, iters=3):
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
    H, W, ci, co = 1024, 1280, 32, 3
    x = torch.randn(B, H, W, ci, device=dev)
    wp = ops.pack_hwio(torch.randn(9, 9, ci, co, device=dev) * 0.02)
    bias = torch.randn(co, device=dev)
    dy = torch.randn(B, H, W, co, device=dev)
    wm, xm, dm = ops.absmax(wp[0]), ops.absmax(x), ops.absmax(dy)
    fl = 2.0 * 81 * ci * co * B * H * W

    def lin(fn):        # the same launch with the linear tile order (default: 4 x 8 tile blocks per XCD)
        ops.set_conv_bf16_impl(4096)
        try:
            return fn()
        finally:
            ops.set_conv_bf16_impl(0)
    r = {}
    for _ in range(3):
        for k, fn in (("f32 fwd", lambda: ops.conv2d_fwd(x, wp, bias, pad=4)),
                      ("fp16x2 fwd", lambda: ops.conv9_fwd_split2(x, xm, wp, wm, bias)),
                      ("fp16x2 fwd linear", lambda: lin(lambda: ops.conv9_fwd_split2(x, xm, wp, wm, bias))),
                      ("f32 dgrad", lambda: ops.conv2d_dgrad(dy, wp, x.shape, pad=4)),
                      ("fp16x2 dgrad", lambda: ops.conv9_dgrad_split2(dy, dm, wp, wm, x.shape)),
                      ("f32 wgrad", lambda: ops.conv2d_wgrad(x, dy, (9, 9, ci, co), pad=4)),
                      ("fp16x2 wgrad", lambda: ops.conv9_wgrad_split2(x, xm, dy, dm)),
                      ("absmax x", lambda: ops.absmax(x))):
            r.setdefault(k, []).append(timeit(fn))
    for k, v in r.items():
        us = sorted(v)[1]
        print("B=%d 9x9 32->3 @%dx%d %-18s %9.1f us  %6.1f TF (useful fp32-equivalent)" % (B, H, W, k, us, fl / us / 1e6))


if __name__ == "__main__":
    main()
