#!/usr/bin/env python3
"""Where do the cycles of k_conv3x3_mfma go?  Builds a private copy of the library with -DDASR_CONV_TIMING
(s_memtime stamps between the phases of the forward kernel, summed over waves) and prints the shares.
Usage: python tools/conv_phase_timing.py [Cin Cout [H W [B]]]"""
import ctypes
import glob
import os
import subprocess
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out", "libdasr_timing.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = sorted(glob.glob(os.path.join(root, "depth-aware-endoscopy-sr_amd", "csrc", "*.hip")))
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-ffp-contract=on"] + (os.environ.get("DASR_EXTRA_DEFS", "-DDASR_CONV_TIMING").split()) + ["-o", out] + srcs)
lib = ctypes.CDLL(out)
args = [int(v) for v in sys.argv[1:]]
Cin, Cout = (args + [128, 128])[:2] if len(args) >= 2 else (128, 128)
H, W = args[2:4] if len(args) >= 4 else (128, 160)
B = args[4] if len(args) >= 5 else 16
dev = torch.device("cuda")
x = torch.randn(B, H, W, Cin, device=dev)
w = torch.randn(2 * 9 * Cin * Cout, device=dev) * 0.05
y = torch.empty(B, H, W, Cout, device=dev)
P = ctypes.c_void_p
lib.dasr_conv2d_fwd.argtypes = [P, P, P, P, P, P] + [ctypes.c_int] * 14 + [P]
lib.dasr_conv2d_fwd.restype = ctypes.c_int


def run():
    rc = lib.dasr_conv2d_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), None, B, H, W, Cin, H, W, Cout, 3, 3, 1, 1,
                             0, 0, 1, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


buf = (ctypes.c_ulonglong * 8)()
run()
torch.cuda.synchronize()
if hasattr(lib, "dasr_debug_conv_phase_read"):
    lib.dasr_debug_conv_phase_read(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
print("conv3x3 %dx%d %d->%d B=%d: %.1f us / launch  [%s]" % (H, W, Cin, Cout, B, e0.elapsed_time(e1) * 100, os.environ.get("DASR_EXTRA_DEFS", "-DDASR_CONV_TIMING")))
if not hasattr(lib, "dasr_debug_conv_phase_read"):
    sys.exit(0)
lib.dasr_debug_conv_phase_read(buf, 1)
names = ["prologue prefetch", "barrier 1 (chunk done)", "commit (vmcnt wait + ds_write)", "barrier 2", "issue next prefetch",
         "MFMA loop", "epilogue", "-"]
tot = float(sum(buf))
print("conv3x3 %dx%d %d->%d B=%d: %.1f us / launch" % (H, W, Cin, Cout, B, e0.elapsed_time(e1) * 100))
for n, v in zip(names, buf):
    print("  %-34s %6.2f %%" % (n, 100.0 * v / tot))
