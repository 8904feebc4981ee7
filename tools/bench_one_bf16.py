#!/usr/bin/env python3
"""One trunk convolution shape in a loop (for rocprofv3 --pmc runs).
Usage: bench_one_bf16.py [fwd|dgrad|wgrad|split|splitw] [Cin Cout H W B [impl]]
  fwd / dgrad / wgrad: the bf16 kernels (impl: ops.set_conv_bf16_impl code - 1 first kernel, 16 / 32 persistent 4- / 8-wave)
  split / splitw: the fp32 split forward / weight gradient in the bf16 x 3 scheme; split2 / split2w: in the fp16 x 2 scheme"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dasr_amd  # noqa
from dasr_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
ci, co, H, W, B = (int(v) for v in sys.argv[2:7]) if len(sys.argv) > 6 else (128, 128, 256, 320, 32)
dev, BF = torch.device("cuda"), torch.bfloat16
if len(sys.argv) > 7:
    ops.set_conv_bf16_impl(int(sys.argv[7]))
if mode in ("split", "splitw"):
    x = torch.randn(B, H, W, ci, device=dev)
    wp = ops.pack_hwio(torch.randn(3, 3, ci, co, device=dev) * 0.05)
    bias = torch.randn(co, device=dev)
    ws = ops.conv3x3_split_weights(wp)
    y = ops.conv3x3_fwd_split(x, ws, bias, co)
    for _ in range(5):
        if mode == "split":
            ops.conv3x3_fwd_split(x, ws, bias, co)
        else:
            ops.conv3x3_wgrad_split(x, y)
    torch.cuda.synchronize()
    sys.exit(0)
if mode in ("split2", "split2w"):
    x = torch.randn(B, H, W, ci, device=dev)
    wp = ops.pack_hwio(torch.randn(3, 3, ci, co, device=dev) * 0.05)
    bias = torch.randn(co, device=dev)
    ws = ops.conv3x3_split2_weights(wp)
    xm = ops.absmax(x)
    y = ops.conv3x3_fwd_split2(x, xm, ws, bias, co)
    ym = ops.absmax(y)
    for _ in range(5):
        if mode == "split2":
            ops.conv3x3_fwd_split2(x, xm, ws, bias, co)
        else:
            ops.conv3x3_wgrad_split2(x, xm, y, ym)
    torch.cuda.synchronize()
    sys.exit(0)
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
