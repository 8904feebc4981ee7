#!/usr/bin/env python3
"""mlp_mask (Cin = 1) conv: forward and fused-ReLU wgrad timings at the bench shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dasr_amd  # noqa
from dasr_amd import ops

dev = "cuda"
B, H, W, C = 16, 128, 160, 128
x = torch.randn(B, H, W, 1, device=dev)
dy = torch.randn(B, H, W, C, device=dev)
y = torch.relu(torch.randn(B, H, W, C, device=dev))


def timeit(fn, iters=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


us = timeit(lambda: ops.conv2d_wgrad_act(x, dy, y, (3, 3, 1, C), 1))
print("c1 wgrad(+relu bwd) %.1f us  %.0f GB/s" % (us, 2 * dy.numel() * 4 / us / 1e3))
