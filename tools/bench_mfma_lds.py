#!/usr/bin/env python3
"""Build and run tools/mfma_lds.hip (MFMA stream fed from LDS like the conv kernels' fragment loop)."""
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
exe = os.path.join(here, "..", "gpurun_out", "mfma_lds")
os.makedirs(os.path.dirname(exe), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-o", exe, os.path.join(here, "mfma_lds.hip")])
sys.exit(subprocess.call([exe]))
