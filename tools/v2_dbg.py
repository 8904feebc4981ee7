#!/usr/bin/env python3
"""Timing experiments on the persistent bf16 convolution (csrc/conv_bf16_v2.hip, library built with -DDASR_V2_DEBUG): the
forward kernel with parts compiled out (WRONG results, timing only) - what each part costs.  HIP events, isolated."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dasr_amd  # noqa
from dasr_amd import ops

BF = torch.bfloat16
NAMES = {0: "full", 1: "no W dma", 2: "no halo dma", 3: "no dma", 4: "no mfma", 8: "no stores", 16: "no wait/barrier",
         32: "no lds reads", 36: "no lds reads, no mfma", 19: "no dma, no barrier", 12: "no mfma, no stores",
         64: "W contiguous", 128: "halo contiguous", 192: "W + halo contiguous", 100: "W contiguous, no lds/mfma",
         228: "W+halo contig, no lds/mfma"}


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
    form = int(sys.argv[1]) if len(sys.argv) > 1 else 32      # 16: 4-wave form, 32: 8-wave form
    dev = torch.device("cuda")
    B, H, W = 32, 256, 320
    for ci, co in ((128, 128), (64, 64)):
        x = torch.randn(B, H, W, ci, device=dev).to(BF)
        wt = ops.pack_hwio((torch.randn(3, 3, ci, co, device=dev) * 0.05).to(BF))
        bias = torch.randn(co, device=dev)
        fl = 2.0 * 9 * ci * co * B * H * W
        for rnd in range(2):
            for dbg in sorted(NAMES):
                ops.set_conv_bf16_impl(form | (dbg << 13))
                us = timeit(lambda: ops.conv2d_fwd(x, wt, bias))
                if rnd == 1:
                    print("form %d %3d->%3d %-24s %8.1f us  %7.1f TF-equivalent" % (form, ci, co, NAMES[dbg], us, fl / us / 1e6))
        ops.set_conv_bf16_impl(0)


if __name__ == "__main__":
    main()
