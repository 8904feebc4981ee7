#!/usr/bin/env python3
"""One bf16 trunk convolution shape in a loop (for rocprofv3 --pmc runs).  Usage: bench_one_bf16.py [fwd|dgrad|wgrad] [Cin Cout H W B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dasr_amd  # noqa
from dasr_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
ci, co, H, W, B = (int(v) for v in sys.argv[2:7]) if len(sys.argv) > 6 else (128, 128, 256, 320, 32)
dev, BF = torch.device("cuda"), torch.bfloat16
x = torch.randn(B, H, W, ci, device=dev).to(BF)
wt = ops.pack_hwio((torch.randn(3, 3, ci, co, device=dev) * 0.05).to(BF))
bias = torch.randn(co, device=dev)
y = ops.conv2d_fwd(x, wt, bias)
for _ in range(5):
    if mode == "fwd":
        ops.conv2d_fwd(x, wt, bias)
    elif mode == "dgrad":
        ops.conv2d_dgrad(y, wt, x.shape, out_dtype=BF)
    else:
        ops.conv2d_wgrad(x, y, tuple(wt.shape[1:]))
torch.cuda.synchronize()
