import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dasr_amd
from dasr_amd import ops, synth
BF = torch.bfloat16
dev = "cuda"
gen = torch.Generator().manual_seed(1)
rn = lambda *s: torch.randn(*s, generator=gen)
bf = lambda x: x.to(BF).float()
B, H, W, C, K = 2, 9, 33, 64, 10
t, gb2, res = bf(rn(B, H, W, C)), bf(rn(B, H, W, 2 * C)), bf(rn(B, H, W, C))
_, _, _, mk = synth.closed_form_batch(1, B, H, W, 1, K)
D = rn(B, 2, 9, K, C) * 0.1
bg, bb = rn(C) * 0.1, rn(C) * 0.1
ag, ab = torch.full((1,), 0.7), torch.full((1,), 0.74)
dout = bf(rn(B, H, W, C))
d = lambda x: x.to(dev)
h = lambda x: x.to(dev).to(BF)
mask = d(mk)
region, flag = ops.mask_compress(mask)
mean, var = ops.instnorm_stats(d(t))
y16 = ops.sean_fwd(h(t), mean, var, h(gb2), mask, region, flag, d(D), d(bg), d(bb), d(ag), d(ab), h(res), True)
y = y16.float()
g32 = ops.sean_bwd(d(dout), y, d(t), mean, var, d(gb2), mask, region, flag, d(D), d(bg), d(bb), d(ag), d(ab), True, True)
g16 = ops.sean_bwd(h(dout), y16, h(t), mean, var, h(gb2), mask, region, flag, d(D), d(bg), d(bb), d(ag), d(ab), True, True)
a, b = g32[2].cpu(), g16[2].cpu()     # [B,2,9,K,C]
print("max |dD32|", a.abs().max().item())
for s in range(2):
    for tap in range(9):
        e = (a[:, s, tap] - b[:, s, tap]).abs().max().item()
        print("s %d tap %d  maxerr %.4f   per-16ch-group:" % (s, tap, e),
              ["%.3f" % (a[:, s, tap, :, 16 * q:16 * q + 16] - b[:, s, tap, :, 16 * q:16 * q + 16]).abs().max().item() for q in range(4)])
print("per region k (tap 4, s 0):", [(a[:, 0, 4, k] - b[:, 0, 4, k]).abs().max().item() for k in range(K)])
