"""Data-parallel harness on 2 CPU ranks (gloo): gradients after the flat all-reduce equal the single-process
gradients on the concatenated batch, and the dynamic loss is the GLOBAL ratio.

The generator itself is communication-free, so the DP logic is exercised with a small stand-in net (pure torch,
CPU); the HIP DepthNet under the same Trainer is covered by bench.py --gpus N on the GPU node."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from dasr_amd import harness, synth


class _TinyNet(torch.nn.Module):
    """forward(input, depthMap, depthMask) -> x8 image in [0,1]; one parameter is never used (like block nb-2)."""

    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3 + 1 + 10, 3 * 64, 3, padding=1)
        self.unused = torch.nn.Parameter(torch.ones(5))
        with torch.no_grad():
            synth.closed_form_fill_(self.state_dict().items())

    def forward(self, x, d, m):
        y = F.pixel_shuffle(self.conv(torch.cat([x, d, m], 1)), 8)
        return torch.clamp(0.5 + 0.2 * y, 0, 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _single_process_reference(B, H, W):
    torch.manual_seed(0)
    net = _TinyNet()
    tr = harness.Trainer(net, 10)
    lq, gt, dm, mk = synth.seeded_batch(0, B, H, W, 8)
    tr.optimize_parameters(lq, gt, dm, mk)
    return {k: v.detach().clone() for k, v in net.state_dict().items()}, \
        tr.dynamic_loss.trainable_weight.detach().clone(), float(tr.log["l_all"])


def _worker(rank, world, port, B, H, W, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    torch.set_num_threads(1)
    net = _TinyNet()
    tr = harness.Trainer(net, 10, group=dist.group.WORLD)
    per = B // world
    lq, gt, dm, mk = synth.seeded_batch(rank * per, per, H, W, 8)
    tr.optimize_parameters(lq, gt, dm, mk)
    if rank == 0:
        torch.save(({k: v.detach().clone() for k, v in net.state_dict().items()},
                    tr.dynamic_loss.trainable_weight.detach().clone(), float(tr.log["l_all"])), q)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_step(tmp_path):
    B, H, W = 4, 8, 10
    ref_sd, ref_w, ref_loss = _single_process_reference(B, H, W)
    ctx = mp.get_context("spawn")
    q = str(tmp_path / "rank0.pt")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, H, W, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    sd, w, loss = torch.load(q)
    assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))       # global dynamic-loss ratio, mean L1
    for k in ref_sd:
        assert torch.allclose(sd[k], ref_sd[k], rtol=1e-4, atol=1e-6), k  # same Adam step => same weights
    assert torch.allclose(w, ref_w, rtol=1e-4, atol=1e-6)
    assert torch.equal(sd["unused"], ref_sd["unused"])      # no gradient anywhere: skipped by the flat all-reduce


def test_lr_schedule_matches_reference_samples(golden_dir):
    import json
    lrs = json.load(open(os.path.join(golden_dir, "lr_schedule.json")))
    for step, lr in lrs.items():
        assert abs(harness.cosine_restart_lr(int(step)) - lr) <= 1e-9 + 1e-6 * lr, step


def test_dynamic_loss_matches_oracle():
    from oracle import depthnet_oracle as O
    torch.manual_seed(1)
    lq, gt, dm, mk = synth.seeded_batch(3, 2, 8, 10, 8)
    sr = torch.rand_like(gt).requires_grad_(True)
    mod = harness.DynamicMaskLoss(10, 10.0)
    per, wl, l_dyn, sm = mod(sr, gt, mk)
    w = torch.ones(10, requires_grad=True)
    per_o, l_dyn_o, sm_o = O.dynamic_mask_loss(sr, gt, mk, w, 10.0)
    assert abs(l_dyn.item() - l_dyn_o.item()) <= 1e-6
    g1, = torch.autograd.grad(l_dyn, sr, retain_graph=True)
    g2, = torch.autograd.grad(l_dyn_o, sr)
    assert torch.allclose(g1, g2, rtol=1e-5, atol=1e-9)
