#!/usr/bin/env python3
"""Where the host time of a training step goes: cProfile over a few eager steps of the x8 bench configuration (the GPU runs
behind; what is measured is Python + ctypes + torch allocator work per launch).  Usage: python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dasr_amd  # noqa
from dasr_amd import harness, networks, prep, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda")
opt = {"network_G": dict(networks.X8_NETWORK_G, upscale=8), "datasets": {"train": {"depthMaskNum": 10}}}
net = networks.define_G(opt).to(dev)
tr = harness.Trainer(net)
lq, gt, dm, mk = [t.to(dev) for t in synth.seeded_batch(0, 16, 128, 160, 8)]
mk = prep.depth_to_masks(dm, 10)
for _ in range(2):
    tr.optimize_parameters(lq, gt, dm, mk)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    tr.optimize_parameters(lq, gt, dm, mk)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
