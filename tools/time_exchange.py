#!/usr/bin/env python3
"""Where the time of the data-parallel gradient exchange goes on ONE GPU (one-rank RCCL group): pack, all-reduce, scale,
unpack, for the x8 net's parameter list in one and in four buckets.  HIP events + host wall time per phase."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import dasr_amd  # noqa
from dasr_amd import networks


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    opt = {"network_G": dict(networks.X8_NETWORK_G), "datasets": {"train": {"depthMaskNum": 10}}}
    net = networks.define_G(opt).cuda()
    grads = [torch.randn_like(p) for p in net.parameters()]
    print("%d tensors, %.1f M floats" % (len(grads), sum(g.numel() for g in grads) / 1e6))

    def phase(name, fn, n=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            out = fn()
        e1.record()
        host = (time.perf_counter() - t0) / n * 1e3
        torch.cuda.synchronize()
        print("%-28s host %.2f ms   device %.2f ms" % (name, host, e0.elapsed_time(e1) / n))
        return out

    for nb in (1, 4):
        k = (len(grads) + nb - 1) // nb
        buckets = [grads[i:i + k] for i in range(0, len(grads), k)]
        print("---- %d bucket(s)" % nb)
        flats = phase("cat", lambda: [torch.cat([g.reshape(-1) for g in b]) for b in buckets])
        phase("all_reduce (sync)", lambda: [dist.all_reduce(f) for f in flats])
        def asyncs():
            ws = [dist.all_reduce(f, async_op=True) for f in flats]
            for w in ws:
                w.wait()
        phase("all_reduce (async + wait)", asyncs)
        phase("div_", lambda: [f.div_(2) for f in flats])
        def unpack():
            for f, b in zip(flats, buckets):
                views, off = [], 0
                for g in b:
                    views.append(f[off:off + g.numel()].view_as(g))
                    off += g.numel()
                torch._foreach_copy_(b, views)
        phase("views + _foreach_copy_", unpack)
    dist.destroy_process_group()


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def trainer_steps():
    """c4's per-GPU step with the exchange off / on (one-rank group, world pretended 2), host time inside the hooks."""
    from dasr_amd import harness, prep, synth
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29578")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    opt = {"network_G": dict(networks.X8_NETWORK_G), "datasets": {"train": {"depthMaskNum": 10}}}
    net = networks.define_G(opt)
    synth.closed_form_fill_(net.state_dict().items())
    net = net.cuda().set_compute_dtype(torch.bfloat16)
    lq, gt, dm, _ = synth.seeded_batch(0, 16, 128, 160, 8)
    lq, gt, dm = lq.cuda(), gt.cuda(), dm.cuda()
    mk = prep.depth_to_masks(dm, 10)
    for world in (1, 2, 1, 2):
        tr = harness.Trainer(net)
        tr.group = dist.group.WORLD
        acc = {"hook": 0.0, "finish": 0.0}
        if world > 1:
            tr._enable_dp(world)
            h0, f0 = tr._on_grad_bucket, tr._finish_allreduce

            def hook(pv, h0=h0):
                t = time.perf_counter()
                h0(pv)
                acc["hook"] += time.perf_counter() - t

            def fin(f0=f0):
                t = time.perf_counter()
                f0()
                acc["finish"] += time.perf_counter() - t
            object.__setattr__(net, "_grad_bucket_hook", hook)
            tr._finish_allreduce = fin
        else:
            tr.world = 1
            object.__setattr__(net, "_grad_bucket_hook", None)
        for _ in range(3):
            tr.optimize_parameters(lq, gt, dm, mk)
        torch.cuda.synchronize()
        acc["hook"] = acc["finish"] = 0.0
        t0 = time.perf_counter()
        n = 6
        for _ in range(n):
            tr.optimize_parameters(lq, gt, dm, mk)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        print("world %d: %.2f ms/step (host enqueue %.2f), in bucket hooks %.2f ms, in finish %.2f ms"
              % (world, tot / n * 1e3, host / n * 1e3, acc["hook"] / n * 1e3, acc["finish"] / n * 1e3))
    dist.destroy_process_group()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "steps":
    trainer_steps()
