#!/usr/bin/env python3
"""bench.py — LR frames/s of the DepthNet hot path (fwd + losses + bwd + Adam) at x8, 128x160 LR.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; weak scaling (16 frames per GPU, BASELINE.json configs[1] at N=1).  Prints ONE JSON
line on rank 0.  `value` = frames all ranks processed / max-over-ranks wall time of exactly K steps, inputs
resident in HBM.  Extra objects: `roofline` (the DGB dynamic-conv + DFN modulation forward kernel,
dasr_sean_fwd, timed with HIP events on its own stream inside the timed steps) and `cpu_baseline` (the
CPU oracle on the host cores, rank 0 at N=1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

import dasr_amd  # noqa: F401
from dasr_amd import harness, networks, ops, synth

LR_H, LR_W, SCALE, K_REGIONS = 128, 160, 8, 10
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def sean_algorithmic_bytes(B, H, W, C, K, residual):
    """3 activation reads (t, gamma2, beta2) + 1 write of C floats + K mask floats per LR pixel
    per SEAN call = 4*(4C) + 4K = 1064 B/px at C=64, K=10 (+ one more C-read for the residual variant)."""
    per_px = 4 * (4 * C + (C if residual else 0)) + 4 * K
    return per_px * B * H * W


class SeanTimer:
    """Wraps ops.sean_fwd with HIP events recorded on the launch stream (torch's current stream)."""

    def __init__(self):
        self.pairs = []
        self.enabled = False
        self._orig = ops.sean_fwd

    def install(self):
        orig = self._orig

        def timed(t, mean, var, gb2, mask, region, flag, D, bias_g, bias_b, alpha_g, alpha_b, residual, relu):
            if not self.enabled:
                return orig(t, mean, var, gb2, mask, region, flag, D, bias_g, bias_b, alpha_g, alpha_b, residual, relu)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(t, mean, var, gb2, mask, region, flag, D, bias_g, bias_b, alpha_g, alpha_b, residual, relu)
            e1.record()
            B, H, W, C = t.shape
            self.pairs.append((e0, e1, sean_algorithmic_bytes(B, H, W, C, mask.shape[1], residual is not None)))
            return out

        ops.sean_fwd = timed

    def summary(self):
        if not self.pairs:
            return None
        ms = [a.elapsed_time(b) for a, b, _ in self.pairs]
        nbytes = [n for _, _, n in self.pairs]
        avg_ms = sum(ms) / len(ms)
        avg_bytes = sum(nbytes) / len(nbytes)
        return avg_ms, avg_bytes, len(ms)


def host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def cpu_baseline(frames=4, steps=2):
    """CPU oracle (as-written PyTorch restatement of the reference, oracle/depthnet_oracle.py) on the host
    cores: forward + losses + backward + Adam on `frames` frame(s) of the same x8 workload."""
    from oracle import depthnet_oracle as O
    cores = min(host_cores(), 32)
    torch.set_num_threads(cores)
    cfg = O.make_cfg()
    sd = O.new_state_dict(cfg)
    synth.closed_form_fill_(sd.items())
    for v in sd.values():
        v.requires_grad_(True)
    w = torch.ones(K_REGIONS, requires_grad=True)
    optim = torch.optim.Adam(list(sd.values()) + [w], lr=1e-3, betas=(0.9, 0.99))
    lq, gt, dm, mk = synth.seeded_batch(0, frames, LR_H, LR_W, SCALE, K_REGIONS)
    times = []
    fwd_times = []
    for i in range(steps + 1):
        print("[bench] cpu baseline step %d/%d (%d threads)" % (i, steps, cores), file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        optim.zero_grad(set_to_none=True)
        sr = O.depthnet_forward(sd, cfg, lq, dm, mk)
        t1 = time.perf_counter()
        print("[bench]   forward %.1f s" % (t1 - t0), file=sys.stderr, flush=True)
        total, _, _, _ = O.total_loss(sr, gt, mk, w)
        total.backward()
        optim.step()
        t2 = time.perf_counter()
        print("[bench]   loss+backward+Adam %.1f s" % (t2 - t1), file=sys.stderr, flush=True)
        if i > 0:                      # first step is the warm-up
            times.append(t2 - t0)
            fwd_times.append(t1 - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(frames / med, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d frame(s) x8 128x160, 1 warm-up + %d timed fwd+loss+bwd+Adam steps of the CPU oracle "
                      "(median); forward-only %.3f frames/s" % (frames, steps, frames / (sum(fwd_times) / len(fwd_times)))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="frames per GPU (configs[1]: 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)"
                             % args.gpus)
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        group = dist.group.WORLD

    opt = {"network_G": dict(networks.X8_NETWORK_G), "datasets": {"train": {"depthMaskNum": K_REGIONS}}}
    net = networks.define_G(opt)
    synth.closed_form_fill_(net.state_dict().items())
    net = net.to(dev)
    trainer = harness.Trainer(net, K_REGIONS, group=group)
    B = args.batch
    lq, gt, dm, mk = (t.to(dev) for t in synth.seeded_batch(rank * B, B, LR_H, LR_W, SCALE, K_REGIONS))

    timer = SeanTimer()
    timer.install()

    def barrier():
        if world > 1:
            dist.barrier()

    def note(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    note("inputs resident; %d warm-up step(s)" % args.warmup)
    for i in range(args.warmup):
        trainer.optimize_parameters(lq, gt, dm, mk)
        torch.cuda.synchronize()
        note("warm-up step %d done" % (i + 1))
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.optimize_parameters(lq, gt, dm, mk)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = el.item()
    loss = float(trainer.log["l_all"])

    # Roofline of the dynamic-conv forward kernel.  In the timed steps the kernel shares the GPU with the side
    # stream's convolutions (graph.SIDE_STREAM), which stretches its duration and says nothing about the kernel, so
    # ONE extra, untimed step is run with the side stream off and the same HIP-event timers: that is `achieved`;
    # the duration seen inside the timed (overlapped) steps is reported next to it.
    overlapped = timer.summary()
    timer.pairs = []
    from dasr_amd import graph as _graph
    _side = _graph.SIDE_STREAM
    _graph.SIDE_STREAM = False
    timer.enabled = True
    trainer.optimize_parameters(lq, gt, dm, mk)
    torch.cuda.synchronize()
    timer.enabled = False
    _graph.SIDE_STREAM = _side
    roof = None
    s = timer.summary()
    if s is not None:
        avg_ms, avg_bytes, n = s
        achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "k_sean_fwd_onehot (dasr_sean_fwd: DGB dynamic conv + DFN modulation, forward)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "launches_timed": n,
                "avg_launch_us": round(avg_ms * 1e3, 2), "algorithmic_bytes_per_launch": int(avg_bytes),
                "measured": "HIP events on the launch stream, one extra un-overlapped step after the timed region"}
        if overlapped is not None:
            o_ms, o_bytes, o_n = overlapped
            roof["in_timed_region"] = {"avg_launch_us": round(o_ms * 1e3, 2), "launches": o_n,
                                       "achieved": round(o_bytes / (o_ms * 1e-3) / 1e9, 1),
                                       "note": "co-running with the side-stream convolutions"}
        pmc = os.path.join(ROOT, "profiles", "sean_fwd_pmc.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("B") == B:          # counters were collected at this launch size
                    roof["traffic"] = int(rec.get("hbm_bytes_per_launch"))
            except Exception:
                pass

    if rank == 0:
        out = {
            "metric": "LR frames/sec fwd+bwd at x8 (128x160 LR)", "value": round(world * B * args.steps / elapsed, 3),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Kvasir x8 synthetic, batch=%d per GPU, fp32, DepthNet nb=16 nf=64 L=256 K=10 "
                                   "(BASELINE.json configs[1]); step = fwd + L1 + dynamic loss + bwd + Adam" % B,
                       "global_batch": world * B, "lr_hw": [LR_H, LR_W], "scale": SCALE,
                       "parallelism": "dp%d" % world},
            "loss": round(loss, 6),
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            note("GPU part done (%.1f ms/step); timing the CPU oracle baseline" % (1e3 * elapsed / args.steps))
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
