#!/usr/bin/env python3
"""Which torch ops issue the small elementwise / copy launches of a step (torch.profiler, one c4-shaped step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dasr_amd  # noqa
from dasr_amd import harness, networks, prep, synth
dev = torch.device("cuda")
opt = {"network_G": dict(networks.X8_NETWORK_G, upscale=8), "datasets": {"train": {"depthMaskNum": 10}}}
net = networks.define_G(opt)
synth.closed_form_fill_(net.state_dict().items())
net = net.to(dev)
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    net.set_compute_dtype(torch.bfloat16)
tr = harness.Trainer(net, 10)
lq, gt, dm, _ = synth.seeded_batch(0, 4, 128, 160, 8, 10)
lq, gt, dm = lq.to(dev), gt.to(dev), dm.to(dev)
mk = prep.depth_to_masks(dm, 10)
for _ in range(2):
    tr.optimize_parameters(lq, gt, dm, mk)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as p:
    tr.optimize_parameters(lq, gt, dm, mk)
    torch.cuda.synchronize()
rows = sorted(p.key_averages(), key=lambda e: -e.count)
for e in rows[:40]:
    print("%-60s count %5d  cpu %8.1f us  cuda %8.1f us" % (e.key[:60], e.count, e.cpu_time_total, getattr(e, "device_time_total", 0.0)))
