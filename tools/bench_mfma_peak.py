#!/usr/bin/env python3
"""Build and run tools/mfma_peak.hip (measured fp32 MFMA ceiling, quoted next to the 157.3 TFLOP/s spec number)."""
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
exe = os.path.join(here, "..", "gpurun_out", "mfma_peak")
os.makedirs(os.path.dirname(exe), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(here, "mfma_peak.hip")])
rc = 0
for w in (sys.argv[1:] or ["8", "4", "2", "1"]):
    rc |= subprocess.call([exe, w])
sys.exit(rc)
