#!/usr/bin/env python3
"""What leaving max |.| behind costs each producer kernel (x8 bench shapes, B = 16): the op with and without its amax argument,
against the dasr_absmax pass it replaces.  HIP events, isolated, median of 3 rounds of 5."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dasr_amd  # noqa
from dasr_amd import ops, prep


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def med(fn):
    return sorted(timeit(fn) for _ in range(3))[1]


def main():
    dev = torch.device("cuda")
    B, H, W, C, K = 16, 128, 160, 64, 10
    z = lambda: torch.zeros(1, device=dev)
    t, gb2, res = torch.randn(B, H, W, C, device=dev), torch.randn(B, H, W, 2 * C, device=dev), torch.randn(B, H, W, C, device=dev)
    dm = torch.rand(B, 1, H, W, device=dev) * 9.99 + 0.01
    mk = prep.depth_to_masks(dm, K)
    region = mk._dasr_region
    D = torch.randn(B, 2, 9, K, C, device=dev) * 0.1
    bg, bb = torch.randn(C, device=dev), torch.randn(C, device=dev)
    ag, ab = torch.full((1,), 0.7, device=dev), torch.full((1,), 0.74, device=dev)
    mean, var = ops.instnorm_stats(t)
    slots = [z() for _ in range(64)]
    it = [0]

    def slot():
        it[0] = (it[0] + 1) % 64
        return slots[it[0]]

    for name, r in (("sean_fwd", None), ("sean_fwd +res", res)):
        a = med(lambda: ops.sean_fwd(t, mean, var, gb2, mk, region, None, D, bg, bb, ag, ab, r, True))
        b = med(lambda: ops.sean_fwd(t, mean, var, gb2, mk, region, None, D, bg, bb, ag, ab, r, True, amax=slot()))
        print("%-28s %8.1f us   with amax %8.1f us" % (name, a, b))
    out = ops.sean_fwd(t, mean, var, gb2, mk, region, None, D, bg, bb, ag, ab, res, True)
    dout = torch.randn(B, H, W, C, device=dev)
    a = med(lambda: ops.sean_bwd(dout, out, t, mean, var, gb2, mk, region, None, D, bg, bb, ag, ab, True, True))
    b = med(lambda: ops.sean_bwd(dout, out, t, mean, var, gb2, mk, region, None, D, bg, bb, ag, ab, True, True, dt_amax=slot(), dgb2_amax=slot()))
    print("%-28s %8.1f us   with amax %8.1f us" % ("sean_bwd (dt, dgb2)", a, b))
    depth = torch.randn(B, H, W, 1, device=dev)
    wm = ops.pack_hwio(torch.randn(3, 3, 1, 128, device=dev) * 0.3)
    bm = torch.randn(128, device=dev)
    a = med(lambda: ops.conv2d_fwd(depth, wm, bm, act=1))
    b = med(lambda: ops.conv2d_fwd(depth, wm, bm, act=1, amax=slot()))
    print("%-28s %8.1f us   with amax %8.1f us" % ("mask conv 1->128", a, b))
    for (ci, co, up, ps, act) in ((128, 128, 1, 1, 0), (64, 64, 1, 1, 0), (32, 128, 4, 2, 2), (32, 32, 4, 1, 2)):
        x = torch.randn(B, H * up, W * up, ci, device=dev)
        ws2 = ops.conv3x3_split2_weights(ops.pack_hwio(torch.randn(3, 3, ci, co, device=dev) * 0.05))
        bias = torch.randn(co, device=dev)
        xm = ops.absmax(x)
        a = med(lambda: ops.conv3x3_fwd_split2(x, xm, ws2, bias, co, None, act, ps))
        b = med(lambda: ops.conv3x3_fwd_split2(x, xm, ws2, bias, co, None, act, ps, amax=slot()))
        y = ops.conv3x3_fwd_split2(x, xm, ws2, bias, co, None, act, ps)
        c = med(lambda: ops.absmax(y))
        print("%-28s %8.1f us   with amax %8.1f us   (absmax pass over y: %.1f us)" % ("split2 %d->%d @%dx%d ps%d" % (ci, co, H * up, W * up, ps), a, b, c))
        if act or ps > 1:
            dy = torch.randn_like(y)
            a = med(lambda: ops.conv2d_epilogue_bwd(dy, y, H * up, W * up, co, act, ps))
            b = med(lambda: ops.conv2d_epilogue_bwd(dy, y, H * up, W * up, co, act, ps, amax=slot()))
            print("%-28s %8.1f us   with amax %8.1f us" % ("  its epilogue backward", a, b))
        del x, y


if __name__ == "__main__":
    main()
