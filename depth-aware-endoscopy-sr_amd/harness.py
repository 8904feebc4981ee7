"""Training / test harness around DepthNet — the counterpart of the reference's model wrapper.

Reproduces what ``F_Model_depthCond`` does for the shipped ymls
(codes/models/F_model_depthCond.py:87-122,146-192,228-234; codes/train.py:179-199):
L1 pixel loss (weight 1) + dynamic depth-aware smooth-L1 loss (weight 10, 10 trainable region
weights, codes/models/modules/mask_loss.py:44-90), Adam(lr 1e-3, betas (0.9, 0.99), wd 0) over the
generator's parameters plus the loss weights, CosineAnnealingLR_Restart stepped BEFORE the optimiser
(codes/models/lr_scheduler.py:34-62).  Losses and optimiser stay PyTorch (north_star); the generator
forward/backward is the HIP path.

Data parallel: one process per GPU (torchrun), ``torch.distributed`` backend ``nccl`` (= RCCL over
xGMI) on the GPU box, ``gloo`` in CPU tests.  The net itself is communication-free (instance norm is
per-sample).  Per step: ONE small all-reduce of the packed loss scalars (K numerators | K denominators |
sum|sr-hr|: the dynamic loss is then the GLOBAL ratio the reference's default nn.DataParallel path computes
on GPU0, SURVEY.md §8e, and the logged L1 is the global-batch value), and the gradient all-reduce (14.8 M
fp32 = 59 MB for the x8 net, plus the 10 loss weights) in FOUR flat buckets issued asynchronously from the
backward tape as each bucket's gradients complete (HR tail, later LR blocks, earlier LR blocks, then encoder /
head / depth branch / loss weights), waited for once before ``optimizer.step()``.
"""
import math

import torch
import torch.nn.functional as F


def cosine_restart_lr(step, base_lr=1e-3, T_period=(20000, 20000, 20000, 20000),
                      restarts=(20000, 40000, 60000), weights=(1, 1, 1), eta_min=1e-7):
    """Learning rate after ``step`` calls of CosineAnnealingLR_Restart.step()
    (closed form of the recursion in codes/models/lr_scheduler.py:46-62)."""
    last_restart, T, w = 0, T_period[0], 1.0
    for j, r in enumerate(restarts):
        if step >= r:
            last_restart, T, w = r, T_period[j + 1], weights[j]
    return eta_min + (base_lr * w - eta_min) * (1 + math.cos(math.pi * (step - last_restart) / T)) / 2


def _world(group=None):
    """Number of data-parallel ranks: derived from torch.distributed itself, so that a caller who passes no group
    while a process group is initialised still gets the GLOBAL dynamic-loss ratio (group None == the WORLD group)."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_world_size(group)
    return 1


class _RegionSums(torch.autograd.Function):
    """sums = (K smooth-L1 numerators | K areas | sum|sr-hr|) in ONE pass over sr, hr (dasr_loss_sums); the
    backward is one elementwise pass (dasr_loss_bwd).  One-hot masks only (region bytes from dasr_mask_compress)."""

    @staticmethod
    def forward(ctx, sr, hr, region, K):
        from . import ops
        sr_c, hr_c = sr.contiguous(), hr.contiguous()
        ctx.save_for_backward(sr_c, hr_c, region)
        ctx.K = K
        return ops.loss_sums(sr_c, hr_c, region, K)

    @staticmethod
    def backward(ctx, dsums):
        from . import ops
        sr, hr, region = ctx.saved_tensors
        return ops.loss_bwd(sr, hr, region, dsums.contiguous(), ctx.K), None, None, None


def fused_losses(sr, hr, mask_list, trainable_weight, pixel_weight, dynamic_weight, group=None):
    """l_pix and l_dynamic from one pass over (sr, hr) on the GPU when the masks are one-hot, or None when they are
    not (the caller then uses the PyTorch formulation).  Same values and gradients as nn.L1Loss +
    dynamic_weight_mask_loss('smoothl1'); with a process group the region sums are made global first."""
    from . import ops
    if not sr.is_cuda or sr.dtype != torch.float32 or mask_list.shape[2] == 0:
        return None
    H, W = sr.shape[2:]
    h, w = mask_list.shape[2:]
    if H % h or W % w or H // h != W // w:
        return None
    from . import graph, prep
    if not prep.attach_region(mask_list):     # no-op for masks from prep.depth_to_masks / already checked this step
        return None
    region = graph.attached_region(mask_list)
    K = mask_list.shape[1]
    sums = _RegionSums.apply(sr, hr, region, K)
    num, den, l1 = sums[:K], sums[K:2 * K].detach(), sums[2 * K]
    world = _world(group)
    l_pix = pixel_weight * l1 / sr.numel()
    l_pix_global = l_pix.detach()
    if world > 1:
        # ONE collective for everything the step needs from the other ranks' losses: K numerators | K denominators | L1 sum
        g = sums.detach().clone()
        torch.distributed.all_reduce(g, group=group)
        gnum, gden = g[:K], g[K:2 * K]
        local = num / gden * world
        per = gnum / gden + (local - local.detach())
        l_pix_global = pixel_weight * g[2 * K] / (sr.numel() * world)
    else:
        per = num / den
    sm = F.softmax(trainable_weight, dim=0)
    l_dyn = (sm * per).sum() * dynamic_weight
    return l_pix, l_dyn, per, sm, l_pix_global


class DynamicMaskLoss(torch.nn.Module):
    """dynamic_weight_mask_loss with the 'smoothl1' criterion (mask_loss.py:44-90).

    ``smooth_l1(m*sr, m*hr)`` only sees the difference ``m*(sr-hr)``, so each region costs one masked
    pass over ``sr-hr``.  With a process group the region sums are made global before the division.
    """

    def __init__(self, num_regions=10, weight=10.0):
        super().__init__()
        self.l_mask_w = weight
        self.trainable_weight = torch.nn.Parameter(torch.ones(num_regions))

    def forward(self, sr, hr, mask_list, group=None, piggyback=None):
        """``piggyback``: a detached scalar summed over the ranks in the same collective as the region sums (the
        harness passes its local L1 value); its global sum is left in ``self.piggyback_sum``."""
        K = mask_list.shape[1]
        assert K == self.trainable_weight.numel(), "dynamic loss: %d trainable region weights but %d mask channels" % (self.trainable_weight.numel(), K)
        sm = F.softmax(self.trainable_weight, dim=0)
        diff = sr - hr
        nums, dens = [], []
        for i in range(K):
            m = F.interpolate(mask_list[:, i:i + 1], size=sr.shape[2:], mode="nearest")
            nums.append(F.smooth_l1_loss(m * diff, torch.zeros_like(diff), reduction="none").sum())
            dens.append(3.0 * m.sum())
        num, den = torch.stack(nums), torch.stack(dens)
        world = _world(group)
        self.piggyback_sum = piggyback
        if world > 1:
            parts = [num.detach(), den.detach()] + ([piggyback.detach().reshape(1)] if piggyback is not None else [])
            g = torch.cat(parts)
            torch.distributed.all_reduce(g, group=group)          # one collective: numerators | denominators | piggyback
            gnum, gden = g[:K], g[K:2 * K]
            if piggyback is not None:
                self.piggyback_sum = g[2 * K]
            # value = global ratio; gradient: this rank's numerator over the global denominator, times world so
            # that the later gradient AVERAGE over ranks equals the gradient of the global loss
            local = num / gden * world
            per = gnum / gden
            per = per + (local - local.detach())
        else:
            per = num / den
        weighted = sm * per
        return list(per.unbind(0)), list(weighted.unbind(0)), weighted.sum() * self.l_mask_w, sm


class Trainer:
    """feed_data / optimize_parameters / test of the reference wrapper, for one DepthNet."""

    def __init__(self, net, num_regions=10, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.0, pixel_weight=1.0,
                 dynamic_weight=10.0, group=None, T_period=(20000,) * 4, restarts=(20000, 40000, 60000),
                 restart_weights=(1, 1, 1), eta_min=1e-7, use_graph=False):
        """``use_graph``: capture the whole step (zero_grad, forward, losses, backward, Adam) into ONE hipGraph after three
        eager steps and replay it afterwards - for steps whose ~2000 launches cost the host more than the GPU needs to run
        them (x8 / 16 frames / bf16: 22-30 ms of enqueue for a 35 ms step).  Single-rank only (the gradient exchange stays
        eager); same-shaped inputs (the loader's tensors are copied into the captured ones); Adam runs in its
        ``capturable`` form with the learning rate in a device tensor that the scheduler refreshes before every replay."""
        self.net = net
        if group is None and torch.distributed.is_available() and torch.distributed.is_initialized():
            group = torch.distributed.group.WORLD
        self.group = group
        self.world = _world(group)
        device = next(net.parameters()).device
        self.dynamic_loss = DynamicMaskLoss(num_regions, dynamic_weight).to(device)
        self.l_pix_w = pixel_weight
        self.params = [p for p in net.parameters() if p.requires_grad] + list(self.dynamic_loss.parameters())
        self.base_lr = lr
        self.use_graph = bool(use_graph) and device.type == "cuda" and self.world == 1
        self._graph = None
        self._static = None
        self._eager_steps = 0
        if self.use_graph:
            self._lr_t = torch.tensor(float(lr), device=device)
            self.optimizer = torch.optim.Adam(self.params, lr=self._lr_t, weight_decay=weight_decay, betas=betas,
                                              capturable=True)
        else:
            # on the GPU the multi-tensor "fused" form: one pass over the ~250 parameter tensors in a few launches instead
            # of the default's ~75 (1.3 ms of GPU time per step at x8; 64.3 -> 63.8 ms/step); same update rule, same state_dict
            self.optimizer = torch.optim.Adam(self.params, lr=lr, weight_decay=weight_decay, betas=betas,
                                              fused=device.type == "cuda")
        self.sched = dict(T_period=T_period, restarts=restarts, weights=restart_weights, eta_min=eta_min)
        self.step_count = 0
        self.log = {}
        self._buckets = []           # [(flat tensor, [parameter indices], async work)] of the step in flight
        self._submitted = set()
        self._pindex = {}
        if self.world > 1:
            self._enable_dp(self.world)

    def _enable_dp(self, world):
        """Switch the gradient exchange on for ``world`` ranks (the constructor does this when the group has more than one
        rank; the single-GPU RCCL test calls it with a pretended world of 2 to exercise the bucketed path)."""
        self.world = world
        net = self.net
        names = getattr(net, "_param_names", None)
        if names is not None:        # the HIP DepthNet: its backward hands over completed gradients per tape bucket
            byname = dict(net.named_parameters())
            slot = {id(p): i for i, p in enumerate(self.params)}
            self._pindex = {n: slot[id(byname[n])] for n in names if id(byname[n]) in slot}
            object.__setattr__(net, "_grad_bucket_hook", self._on_grad_bucket)

    def update_learning_rate(self):
        self.step_count += 1                       # scheduler.step() comes first (codes/train.py:194)
        lr = cosine_restart_lr(self.step_count, self.base_lr, **self.sched)
        if self.use_graph:
            self._lr_t.fill_(lr)                   # the captured Adam reads the tensor
        else:
            for g in self.optimizer.param_groups:
                g["lr"] = lr
        return lr

    def _submit_bucket(self, idx, grads):
        """Pack ``grads`` (of parameters ``idx``) with one concatenation and start their all-reduce asynchronously."""
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        work = torch.distributed.all_reduce(flat, group=self.group, async_op=True)
        self._buckets.append((flat, idx, work))
        self._submitted.update(idx)

    def _on_grad_bucket(self, pvars):
        """Called from the backward tape at each bucket boundary (tape.mark) with the net's parameter Vars: whatever has a
        gradient by now and has not been sent yet is complete (every parameter is packed exactly once) and goes out.
        Parameters that never receive a gradient (the constructed-but-unused block ``depth-residual{nb-2}``,
        SURVEY.md §8a row 1) never show up, on every rank alike, so the bucket layouts are identical everywhere."""
        idx, grads = [], []
        cur = torch.cuda.current_stream() if pvars and pvars[0].data.is_cuda else None
        for v in pvars:
            i = self._pindex.get(v.name)
            if i is None or v.grad is None or i in self._submitted:
                continue
            if cur is not None and v.grad_event is not None:   # produced on a side stream: order the pack after it
                cur.wait_event(v.grad_event)
            idx.append(i)
            grads.append(v.grad)
        self._submit_bucket(idx, grads)

    def _finish_allreduce(self):
        """The last bucket (everything the tape buckets did not cover, incl. the loss weights), then wait for all of them
        and write the averaged gradients back with one multi-tensor copy per bucket."""
        if self.world <= 1:
            return
        idx = [i for i, p in enumerate(self.params) if p.grad is not None and i not in self._submitted]
        self._submit_bucket(idx, [self.params[i].grad for i in idx])
        for flat, idx, work in self._buckets:
            work.wait()
            flat.div_(self.world)
            grads = [self.params[i].grad for i in idx]
            views, off = [], 0
            for g in grads:
                k = g.numel()
                views.append(flat[off:off + k].view_as(g))
                off += k
            torch._foreach_copy_(grads, views)
        self._buckets = []
        self._submitted = set()

    def optimize_parameters(self, lq, gt, depth, masks):
        if self.use_graph:
            return self._graphed_step(lq, gt, depth, masks)
        return self._eager_step(lq, gt, depth, masks)

    def _graphed_step(self, lq, gt, depth, masks):
        """Three eager steps (allocator pools, Adam state, side streams come into being), then capture, then replays."""
        if self._graph is None and self._eager_steps < 3:
            self._eager_steps += 1
            return self._eager_step(lq, gt, depth, masks)
        self.update_learning_rate()
        if self._graph is None:
            from . import prep
            prep.attach_region(masks)              # any host read-back happens before the capture, not inside it
            self._static = (lq, gt, depth, masks)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step_body(lq, gt, depth, masks)
            self._graph = g
            self._static_log = self.log
        else:
            if self._static[3] is not masks:       # region bytes travel with the mask tensor they were derived from
                from . import graph as _g, ops as _ops, prep as _prep
                _prep.attach_region(masks)
                r_new, r_old = _g.attached_region(masks), getattr(self._static[3], "_dasr_region", None)
                if r_new is None or r_old is None:
                    raise ValueError("Trainer(use_graph=True): masks must be one-hot (prep.depth_to_masks / attach_region)")
                r_old.copy_(r_new)
            for dst, src in zip(self._static, (lq, gt, depth, masks)):
                if dst is not src:
                    if dst.shape != src.shape:
                        raise ValueError("Trainer(use_graph=True): the captured step has inputs of shape %s, got %s"
                                         % (tuple(dst.shape), tuple(src.shape)))
                    dst.copy_(src)
            if self._static[3] is not masks:       # the copy bumped the captured mask tensor's version: re-stamp its bytes
                self._static[3]._dasr_version = _ops.tensor_version(self._static[3])
        self._graph.replay()
        self.log = self._static_log
        return self.log

    def _eager_step(self, lq, gt, depth, masks):
        self.update_learning_rate()
        return self._step_body(lq, gt, depth, masks)

    def _step_body(self, lq, gt, depth, masks):
        self.optimizer.zero_grad(set_to_none=True)
        if masks.is_cuda:
            from . import prep
            prep.attach_region(masks)          # before the step's work is queued; free for prep.depth_to_masks output
        sr = self.net(lq, depth, masks)
        grp = self.group if self.world > 1 else None
        fused = fused_losses(sr, gt, masks, self.dynamic_loss.trainable_weight, self.l_pix_w,
                             self.dynamic_loss.l_mask_w, grp)
        if fused is not None:
            l_pix, l_dyn, per, sm, l_pix_log = fused
        else:
            l_pix = self.l_pix_w * F.l1_loss(sr, gt)
            per, weighted, l_dyn, sm = self.dynamic_loss(sr, gt, masks, grp, piggyback=l_pix.detach())
            l_pix_log = self.dynamic_loss.piggyback_sum / self.world     # the global-batch value
        total = l_pix + l_dyn
        self._buckets, self._submitted = [], set()
        total.backward()
        self._finish_allreduce()
        self.optimizer.step()
        self.log = {"l_all": l_pix_log + l_dyn.detach(), "l_pix": l_pix_log, "l_dynamic": l_dyn.detach()}
        return self.log

    @torch.no_grad()
    def test(self, lq, depth, masks):
        self.net.eval()
        sr = self.net(lq, depth, masks)
        self.net.train()
        return sr


# ---- checkpoint interop (codes/models/base_model.py:77-119) --------------------------------
def save_network(net, path):
    """base_model.save_network: the module's state_dict (reference key names) with CPU tensors."""
    net = getattr(net, "module", net)
    torch.save({k: v.detach().cpu() for k, v in net.state_dict().items()}, path)


def load_network(path, net, strict=True):
    """base_model.load_network: accepts the authors' `*_G.pth` files (a leading 'module.' from DataParallel
    checkpoints is stripped); strict key matching as in the reference."""
    net = getattr(net, "module", net)
    loaded = torch.load(path, map_location="cpu")
    clean = {(k[7:] if k.startswith("module.") else k): v for k, v in loaded.items()}
    return net.load_state_dict(clean, strict=strict)


def _save_training_state(trainer, path, epoch=0):
    """base_model.save_training_state: {'epoch','iter','schedulers','optimizers'}; the scheduler here is stateless
    given the step, so its entry holds the step and its constants.  'loss_weights' is an addition: the reference keeps
    the 10 dynamic-loss weights outside netG and does not checkpoint them at all."""
    state = {"epoch": epoch, "iter": trainer.step_count,
             "schedulers": [dict(last_epoch=trainer.step_count, base_lr=trainer.base_lr, **trainer.sched)],
             "optimizers": [trainer.optimizer.state_dict()],
             "loss_weights": trainer.dynamic_loss.trainable_weight.detach().cpu()}
    torch.save(state, path)


def _resume_training(trainer, state):
    """base_model.resume_training."""
    if isinstance(state, str):
        state = torch.load(state, map_location="cpu")
    assert len(state["optimizers"]) == 1, "Wrong lengths of optimizers"
    assert len(state["schedulers"]) == 1, "Wrong lengths of schedulers"
    trainer.optimizer.load_state_dict(state["optimizers"][0])
    trainer.step_count = int(state["schedulers"][0]["last_epoch"])
    if "loss_weights" in state:
        with torch.no_grad():
            trainer.dynamic_loss.trainable_weight.copy_(state["loss_weights"].to(trainer.dynamic_loss.trainable_weight.device))
    return state.get("epoch", 0), state.get("iter", trainer.step_count)


Trainer.save_training_state = _save_training_state
Trainer.resume_training = _resume_training


def calculate_psnr(img1, img2):
    """PSNR on [0,255] images (codes/utils/util.py:646-653)."""
    mse = torch.mean((img1.double() - img2.double()) ** 2).item()
    if mse == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(mse))
