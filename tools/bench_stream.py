#!/usr/bin/env python3
"""Measured HBM ceilings next to the spec-sheet 8 TB/s: plain copy (1R:1W) and a 3R:1W elementwise kernel on
arrays of the SEAN forward's size (torch's own elementwise kernels; HIP events on the current stream)."""
import argparse
import torch


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    n = a.batch * 128 * 160 * 64
    dev = torch.device("cuda")
    x, y, z, o = (torch.randn(n, device=dev) for _ in range(4))
    big = torch.randn(4 * n, device=dev)
    bigo = torch.empty_like(big)
    us = timeit(lambda: o.copy_(x), a.iters)
    print("copy 1R:1W   (%4d MB)  %7.1f us  %7.1f GB/s" % (8 * n // 2**20, us, 8 * n / us / 1e3))
    us = timeit(lambda: bigo.copy_(big), a.iters)
    print("copy 1R:1W   (%4d MB)  %7.1f us  %7.1f GB/s" % (32 * n // 2**20, us, 32 * n / us / 1e3))
    us = timeit(lambda: torch.addcmul(x, y, z, out=o), a.iters)
    print("addcmul 3R:1W (%4d MB)  %7.1f us  %7.1f GB/s" % (16 * n // 2**20, us, 16 * n / us / 1e3))
    us = timeit(lambda: torch.add(x, y, out=o), a.iters)
    print("add 2R:1W    (%4d MB)  %7.1f us  %7.1f GB/s" % (12 * n // 2**20, us, 12 * n / us / 1e3))
    us = timeit(lambda: x.sum(), a.iters)
    print("sum 1R:0W    (%4d MB)  %7.1f us  %7.1f GB/s" % (4 * n // 2**20, us, 4 * n / us / 1e3))
    us = timeit(lambda: o.fill_(1.0), a.iters)
    print("fill 0R:1W   (%4d MB)  %7.1f us  %7.1f GB/s" % (4 * n // 2**20, us, 4 * n / us / 1e3))


if __name__ == "__main__":
    main()
