"""Data-parallel harness on 2 CPU ranks (gloo): gradients after the flat all-reduce equal the single-process
gradients on the concatenated batch, and the dynamic loss is the GLOBAL ratio.

The generator itself is communication-free.  The DP logic is exercised twice: with a small stand-in net (pure torch),
and with the HIP DepthNet itself - its kernels run by the CPU kernel emulator in each rank - so that the tape's
side-stream bookkeeping and the no-clone gradient hand-off meet the real flat all-reduce."""
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


def _worker(rank, world, port, B, H, W, q, pass_group=True):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    torch.set_num_threads(1)
    net = _TinyNet()
    # group=None while torch.distributed is initialised must mean the WORLD group everywhere (gradient average AND
    # the global dynamic-loss ratio), not "average the gradients but use the local ratio"
    tr = harness.Trainer(net, 10, group=dist.group.WORLD) if pass_group else harness.Trainer(net, 10)
    assert tr.world == world and tr.group is not None
    per = B // world
    lq, gt, dm, mk = synth.seeded_batch(rank * per, per, H, W, 8)
    tr.optimize_parameters(lq, gt, dm, mk)
    if rank == 0:
        torch.save(({k: v.detach().clone() for k, v in net.state_dict().items()},
                    tr.dynamic_loss.trainable_weight.detach().clone(), float(tr.log["l_all"])), q)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pass_group", [True, False], ids=["group_given", "group_none"])
def test_two_rank_step_equals_single_process_step(tmp_path, pass_group):
    B, H, W = 4, 8, 10
    ref_sd, ref_w, ref_loss = _single_process_reference(B, H, W)
    ctx = mp.get_context("spawn")
    q = str(tmp_path / "rank0.pt")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, H, W, q, pass_group)) for r in range(2)]
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


# ---- the HIP DepthNet (kernels on the CPU emulator) under the same Trainer, 2 gloo ranks ---------------------------
_HIP_CASE = dict(name="dp_hip", scale=2, which=[0, 1], L=16, nb=4, B=2, H=8, W=12)
_ZERO_GRAD_KEYS = (".conv1.0.bias", ".conv2.0.bias")      # mathematically zero gradients: pure rounding noise


def _hip_step(emu_lib, rank, world, group):
    """One Trainer step of the HIP DepthNet on this rank's share of the batch; returns (loss, {name: grad})."""
    os.environ["DASR_HIPEMU_LIB"] = emu_lib
    from dasr_amd import _lib
    from dasr_amd.depthnet import DepthNet
    _lib.reset_for_tests()
    c = _HIP_CASE
    net = DepthNet(which_ResBlk_depth=c["which"], nb=c["nb"], scale=c["scale"], depth_latent_ch=c["L"])
    synth.closed_form_fill_(net.state_dict().items())
    tr = harness.Trainer(net, 10, group=group)
    per = c["B"] // world
    lq, gt, dm, mk = synth.closed_form_batch(rank * per, per, c["H"], c["W"], c["scale"])
    sizes = []
    orig = tr._submit_bucket
    tr._submit_bucket = lambda idx, grads: (sizes.append(len(idx)), orig(idx, grads))[1]
    log = tr.optimize_parameters(lq, gt, dm, mk)
    if world > 1:        # the tape's bucket boundaries fired inside the backward; the last bucket carries the loss weights
        assert len(sizes) >= 3 and sizes[0] > 0 and sizes[-1] > 0, sizes
    else:
        assert sizes == [], sizes
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    grads["loss_w"] = tr.dynamic_loss.trainable_weight.grad.detach().clone()
    return float(log["l_all"]), grads


def _hip_worker(rank, world, port, emu_lib, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    loss, grads = _hip_step(emu_lib, rank, world, dist.group.WORLD)
    if rank == 0:
        torch.save((loss, grads), q)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_hip_depthnet_step_equals_single_process_step(tmp_path):
    from dasr_amd import build
    emu_lib = build.build_emu()
    prev = os.environ.get("DASR_HIPEMU_LIB")
    try:
        ref_loss, ref_grads = _hip_step(emu_lib, 0, 1, None)
    finally:
        from dasr_amd import _lib
        if prev is None:
            os.environ.pop("DASR_HIPEMU_LIB", None)
        else:
            os.environ["DASR_HIPEMU_LIB"] = prev
        _lib.reset_for_tests()
    ctx = mp.get_context("spawn")
    q = str(tmp_path / "hip_rank0.pt")
    port = _free_port()
    procs = [ctx.Process(target=_hip_worker, args=(r, 2, port, emu_lib, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    loss, grads = torch.load(q)
    assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (loss, ref_loss)
    assert set(grads) == set(ref_grads)                 # the never-called block is skipped on every rank alike
    assert not any(k.startswith("depth-residual2.") for k in grads)
    num = den = 0.0
    for k, g in ref_grads.items():
        if any(z in k for z in _ZERO_GRAD_KEYS):
            continue
        num += (grads[k].double() - g.double()).pow(2).sum().item()
        den += g.double().pow(2).sum().item()
    rel = (num / den) ** 0.5
    print("2-rank HIP DepthNet step: loss %.6f vs %.6f, gradient rel L2 %.3g" % (loss, ref_loss, rel))
    assert rel <= 2e-5, rel


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
