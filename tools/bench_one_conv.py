#!/usr/bin/env python3
"""Run one conv shape a few times (for rocprofv3 --pmc). Usage: bench_one_conv.py H W Cin Cout [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dasr_amd  # noqa
from dasr_amd import ops
h, w, ci, co = (int(v) for v in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 16
dev = torch.device("cuda")
torch.manual_seed(0)
x = torch.randn(B, h, w, ci, device=dev)
wt = ops.pack_hwio(torch.randn(3, 3, ci, co, device=dev) * 0.05)
bias = torch.randn(co, device=dev)
for _ in range(3):
    y = ops.conv2d_fwd(x, wt, bias)
    dx = ops.conv2d_dgrad(y, wt, x.shape)
    dw, db = ops.conv2d_wgrad(x, y, tuple(wt.shape[1:]))
torch.cuda.synchronize()
print("done")
