#!/usr/bin/env python3
"""What would an fp32 residual stream buy the bf16 path?  CPU study on the oracle's model of the bf16 rounding points
(oracle.bf16_storage: the HIP bf16 path lands within 0.05 dB of it at c3's frame size): one x4 frame of BASELINE configs[2]
(256x320 LR, nb=16, L=256), image PSNR against the fp32 oracle and rel-L2 / cosine of the harness-loss gradient against the
fp32 gradient, for the shipping rounding points and with some tensors kept in fp32.  Usage: bf16_model_study.py [H W scale]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dasr_amd  # noqa
from dasr_amd import synth
from oracle import depthnet_oracle as O


def run(sd, cfg, lq, gt, dm, mk):
    for v in sd.values():
        v.grad = None
    w = torch.ones(10, requires_grad=True)
    sr = O.depthnet_forward(sd, cfg, lq, dm, mk)
    total, _, _, _ = O.total_loss(sr, gt, mk, w)
    total.backward()
    g = torch.cat([v.grad.reshape(-1) for k, v in sd.items() if v.grad is not None and
                   not (k.endswith("conv1.0.bias") or k.endswith("conv2.0.bias"))]).double()
    return sr.detach(), g


def main():
    H, W, scale = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 320, 4)
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = O.make_cfg(scale=scale)
    sd = O.new_state_dict(cfg)
    synth.closed_form_fill_(sd.items())
    for v in sd.values():
        v.requires_grad_(True)
    lq, gt, dm, mk = synth.seeded_batch(0, 1, H, W, scale, 10)
    t0 = time.time()
    ref, gref = run(sd, cfg, lq, gt, dm, mk)
    print("fp32 reference: %.0f s" % (time.time() - t0), flush=True)
    for keep in ([], ["block_out"], ["conv_out"], ["block_out", "conv_out"], ["block_out", "conv_out", "gb2"]):
        O.BF16_KEEP_FP32 = set(keep)
        with O.bf16_storage():
            out, g = run(sd, cfg, lq, gt, dm, mk)
        O.BF16_KEEP_FP32 = set()
        rel = float((g - gref).norm() / gref.norm())
        cos = float((g @ gref) / (g.norm() * gref.norm()))
        print("fp32-kept %-32s PSNR vs fp32 image %6.2f dB   loss-gradient rel-L2 %.4f  cosine %.5f"
              % ("+".join(keep) or "(none: shipping)", O.psnr_255(out, ref), rel, cos), flush=True)


if __name__ == "__main__":
    main()
