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
    H, W = 128, 160
    for ci, co in ((128, 128), (64, 64)):
        x = torch.randn(B, H, W, ci, device=dev)
        wp = ops.pack_hwio(torch.randn(3, 3, ci, co, device=dev) * 0.05)
        bias = torch.randn(co, device=dev)
        ws = ops.conv3x3_split_weights(wp)
        y = ops.conv2d_fwd(x, wp, bias)
        fl = 2.0 * 9 * ci * co * B * H * W
        r = {k: [] for k in ("f32 fwd", "split fwd", "f32 dgrad", "split dgrad", "f32 wgrad", "split wgrad", "split weights")}
        for _ in range(3):
            r["f32 fwd"].append(timeit(lambda: ops.conv2d_fwd(x, wp, bias)))
            r["split fwd"].append(timeit(lambda: ops.conv3x3_fwd_split(x, ws, bias, co)))
            r["f32 dgrad"].append(timeit(lambda: ops.conv2d_dgrad(y, wp, x.shape)))
            r["split dgrad"].append(timeit(lambda: ops.conv3x3_dgrad_split(y, ws, x.shape)))
            r["f32 wgrad"].append(timeit(lambda: ops.conv2d_wgrad(x, y, (3, 3, ci, co))))
            r["split wgrad"].append(timeit(lambda: ops.conv3x3_wgrad_split(x, y)))
            r["split weights"].append(timeit(lambda: ops.conv3x3_split_weights(wp)))
        for k, v in r.items():
            us = sorted(v)[1]
            print("B=%d %3d->%3d %-14s %8.1f us  %6.1f TF (fp32-equivalent)  %7.1f TF of bf16 MFMA work"
                  % (B, ci, co, k, us, fl / us / 1e6, (6 if "split" in k else 0) * fl / us / 1e6))


if __name__ == "__main__":
    main()
