"""``DepthNet`` — drop-in for the reference generator, running on the HIP kernels.

Same constructor keywords, same ``forward(input, depthMap, depthMask)``, same ``state_dict`` keys
(incl. ``weight_g``/``weight_v`` and the hyphenated block names registered with ``add_module``)
as ``codes/models/modules/sftmd_arch.py:837-950`` of the reference, so that
``codes/models/networks.py:41-49`` (``define_G``) can construct it and
``F_Model_depthCond`` (``codes/models/F_model_depthCond.py:161,232``) can call it unchanged.

The sub-modules here only own parameters (with PyTorch's default initialisation); they have no
``forward`` of their own.  ``DepthNet.forward`` hands the parameters to ``graph.depthnet_forward``
which launches HIP kernels through the C ABI; gradients come from the tape's hand-written
backward, bridged into ``torch.autograd`` by one ``autograd.Function`` so that the reference's
losses, ``loss.backward()`` and ``torch.optim.Adam`` work as before.
"""
import contextlib
import math

import torch
import torch.nn as nn

from . import graph
from .tape import Tape, Var


# ---------------------------------------------------------------------------------------------
# parameter holders
# ---------------------------------------------------------------------------------------------
def _default_conv_init_(weight, bias, fan_in):
    bound = 1.0 / math.sqrt(fan_in)          # kaiming_uniform_(a=sqrt(5)) as in nn.Conv2d.reset_parameters
    with torch.no_grad():
        weight.uniform_(-bound, bound)
        if bias is not None:
            bias.uniform_(-bound, bound)


class _PlainConv(nn.Module):
    """Parameters of an ``nn.Conv2d`` (keys ``weight``, ``bias``)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        _default_conv_init_(self.weight, self.bias, cin * k * k)

    def extra_repr(self):
        return "%d, %d, kernel_size=%d" % (self.cin, self.cout, self.k)

    def forward(self, *a, **k):
        raise RuntimeError("parameter holder: DepthNet.forward drives the HIP kernels, sub-modules are not callable")


class _WNConv(nn.Module):
    """Parameters of ``weight_norm(nn.Conv2d | nn.ConvTranspose2d)`` (keys ``bias, weight_g, weight_v``)."""

    def __init__(self, cin, cout, k=3, transposed=False):
        super().__init__()
        self.cin, self.cout, self.k, self.transposed = cin, cout, k, transposed
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        v = torch.empty(shape)
        bias = torch.empty(cout)
        _default_conv_init_(v, bias, shape[1] * k * k)     # torch's fan_in = size(1) * receptive field
        self.bias = nn.Parameter(bias)
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).reshape(-1, 1, 1, 1))
        self.weight_v = nn.Parameter(v)

    def extra_repr(self):
        return "%d, %d, kernel_size=%d, weight_norm%s" % (self.cin, self.cout, self.k,
                                                          ", transposed" if self.transposed else "")

    forward = _PlainConv.forward


class _SEAN(nn.Module):
    """Parameters of the reference's SEAN / DFN (normalization.py:8-49), registration order kept."""

    def __init__(self, label_nc, norm_nc, len_latent, use_trainable_params):
        super().__init__()
        if use_trainable_params:
            self.alpha_beta = nn.Parameter(torch.rand(1))
            self.alpha_gamma = nn.Parameter(torch.rand(1))
        self.A_i_j = _PlainConv(label_nc, label_nc, 1)
        self.mlp_gamma_s = _PlainConv(len_latent, norm_nc, 3)
        self.mlp_beta_s = _PlainConv(len_latent, norm_nc, 3)
        self.mlp_mask = nn.Sequential(_PlainConv(1, 2 * norm_nc, 3), nn.ReLU())
        self.mlp_gamma_o = _PlainConv(2 * norm_nc, norm_nc, 3)
        self.mlp_beta_o = _PlainConv(2 * norm_nc, norm_nc, 3)

    forward = _PlainConv.forward


class _DepthBlock(nn.Module):
    """Parameters of Depth_Residual_Block_Mask (sftmd_arch.py:808-824)."""

    def __init__(self, nf, latent, K, use_trainable_params):
        super().__init__()
        self.norm1 = _SEAN(K, nf, latent, use_trainable_params)
        self.actv1 = nn.ReLU(True)
        self.norm2 = _SEAN(K, nf, latent, use_trainable_params)
        self.conv1 = nn.Sequential(_PlainConv(nf, nf, 3), nn.InstanceNorm2d(nf, affine=False))
        self.conv2 = nn.Sequential(_PlainConv(nf, nf, 3), nn.InstanceNorm2d(nf, affine=False))

    forward = _PlainConv.forward


class _ClassicBlock(nn.Module):
    """Parameters of Classic_Residual_Block, weight-norm variant (sftmd_arch.py:128-146)."""

    def __init__(self, nf):
        super().__init__()
        self.block = nn.Sequential(_WNConv(nf, nf), nn.ReLU(True), _WNConv(nf, nf))

    forward = _PlainConv.forward


class _Encoder(nn.Module):
    """Parameters of Encoder, weight-norm variant (sftmd_arch.py:735-750)."""

    def __init__(self, in_nc, latent):
        super().__init__()
        self.layer1 = _WNConv(in_nc, 32)
        self.layer2 = _WNConv(32, 64)
        self.layer3 = _WNConv(64, 128)
        self.layer4 = _WNConv(128, latent, transposed=True)
        self.layer5 = _WNConv(latent, latent)

    forward = _PlainConv.forward


# ---------------------------------------------------------------------------------------------
# autograd bridge
# ---------------------------------------------------------------------------------------------
def _device_guard(t):
    """Kernels launch on the current stream of the CURRENT device: make that the tensors' device (a net on cuda:1
    called while cuda:0 is current, nn.DataParallel worker threads, autograd's backward threads)."""
    return torch.cuda.device(t.device) if t.is_cuda else contextlib.nullcontext()


class _DepthNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, act_dtype, inp, depth_map, depth_mask, region, *params):
        tape = Tape(enabled=True, act_dtype=act_dtype)
        tape.prepack = net._prepack_for(inp.device, act_dtype)
        pvars = [Var(p.detach(), p.requires_grad, name) for name, p in zip(net._param_names, params)]
        P = {v.name: v for v in pvars}
        out = graph.depthnet_forward(tape, P, net.cfg, net._consts(inp.device), inp.detach().contiguous(),
                                     depth_map.detach().contiguous(), depth_mask.detach().contiguous(), region)
        ctx.tape, ctx.pvars, ctx.out = tape, pvars, out
        ctx.hook = getattr(net, "_grad_bucket_hook", None)
        return out.data

    @staticmethod
    def backward(ctx, dout):
        tape, pvars, out = ctx.tape, ctx.pvars, ctx.out
        if tape is None:
            raise RuntimeError("DepthNet: backward called twice on the same forward")
        out.grad = dout.contiguous()
        if ctx.hook is not None:
            # data-parallel harness: parameter gradients completed so far, handed over at every bucket boundary of the
            # tape so that their all-reduce overlaps the rest of the backward (harness.Trainer._on_grad_bucket)
            hook = ctx.hook
            tape.on_mark = lambda: hook(pvars)
        with _device_guard(dout):
            tape.backward()
        grads = tuple(v.grad for v in pvars)
        # drop every other reference to the gradient tensors: autograd's AccumulateGrad then adopts them as
        # `param.grad` instead of cloning each one (~300 extra copy kernels per step otherwise)
        for v in pvars:
            v.grad = None
        ctx.tape = ctx.pvars = ctx.out = ctx.hook = None
        return (None, None, None, None, None, None) + grads


class DepthNet(nn.Module):
    def __init__(self, which_ResBlk_depth=[], in_nc=3, out_nc=3, nf=64, nb=16, scale=4, input_para=10, min=0.0,
                 max=1.0, depth_latent_ch=256, depthRangeNum=10, norm_type='weight_norm', use_trainable_params=True,
                 norm_gamma=0.1, norm_beta=0.1, ablate_depth_matrix=False, ablate_depth_block=False):
        super().__init__()
        if norm_type != 'weight_norm':
            raise NotImplementedError("dasr_amd.DepthNet: only norm_type='weight_norm' (all shipped ymls) is built")
        if ablate_depth_matrix or ablate_depth_block:
            raise NotImplementedError("dasr_amd.DepthNet: the ablate_* variants are outside the hot path (SURVEY §8b)")
        if scale not in (2, 3, 4, 8):
            raise NotImplementedError("dasr_amd.DepthNet: scale must be 2, 3, 4 or 8")
        self.scale, self.min, self.max, self.para = scale, min, max, input_para
        self.num_blocks = nb
        self.which_ResBlk_depth = list(which_ResBlk_depth)
        self.isBaseline = len(self.which_ResBlk_depth) == 0
        self.ablate_depth_matrix, self.ablate_depth_block = False, False
        self.cfg = dict(which_ResBlk_depth=self.which_ResBlk_depth, in_nc=in_nc, out_nc=out_nc, nf=nf, nb=nb,
                        scale=scale, depth_latent_ch=depth_latent_ch, depthRangeNum=depthRangeNum,
                        use_trainable_params=use_trainable_params, norm_gamma=norm_gamma, norm_beta=norm_beta,
                        out_min=float(min), out_max=float(max))

        self.encoder = _Encoder(in_nc, depth_latent_ch)
        self.head = nn.Sequential(_WNConv(32, 64), nn.LeakyReLU(0.2), _WNConv(64, 64), nn.LeakyReLU(0.2))
        for name, kind, ch in graph.block_plan(self.cfg):
            if kind == "depth":
                self.add_module(name, _DepthBlock(ch, depth_latent_ch, depthRangeNum, use_trainable_params))
            else:
                self.add_module(name, _ClassicBlock(ch))
        ch_last2_upscale = 64 if scale == 4 else 32
        ch_last_upscale = 64 if scale < 4 else 32
        self.upscale1 = nn.Sequential(_WNConv(64, 64 * 4), nn.PixelShuffle(2), nn.LeakyReLU(0.2, inplace=True),
                                      _WNConv(64, 32), nn.LeakyReLU(0.2, inplace=True))
        self.upscale2 = nn.Sequential(_WNConv(ch_last2_upscale, 32 * 4), nn.PixelShuffle(2),
                                      nn.LeakyReLU(0.2, inplace=True), _WNConv(32, 32),
                                      nn.LeakyReLU(0.2, inplace=True))
        final_scale = 3 if scale == 3 else 2
        self.upscale3 = nn.Sequential(_WNConv(ch_last_upscale, 32 * final_scale ** 2), nn.PixelShuffle(final_scale),
                                      nn.LeakyReLU(0.2, inplace=True))
        self.conv_output = _PlainConv(32, out_nc, 9)
        self._const_cache = {}
        # Parameter names in registration order, fixed here.  forward() resolves them through the module tree's
        # ATTRIBUTES, not named_parameters(): an nn.DataParallel replica keeps its (broadcast, non-leaf) weights as
        # plain attributes and exposes no parameters at all (torch >= 1.5, torch/nn/parallel/replicate.py).
        self._param_names = tuple(name for name, _ in self.named_parameters())
        self._param_paths = tuple(tuple(name.split(".")) for name in self._param_names)
        self.compute_dtype = torch.float32

    def set_compute_dtype(self, dtype):
        """Storage type of the trunk's activations: ``torch.float32`` (default, the reference's precision) or
        ``torch.bfloat16`` (BASELINE.json configs[2..3]: bf16 activations and bf16-MFMA trunk convolutions; parameters,
        their gradients, instance-norm statistics, the dynamic kernels and every accumulator stay fp32; inputs and the
        returned image are fp32).  Never switched silently: either this call, or the caller's own
        ``torch.autocast('cuda', dtype=torch.bfloat16)`` around ``forward`` - the way a reference checkout opts in."""
        if isinstance(dtype, str):
            dtype = {"f32": torch.float32, "fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
                     "bfloat16": torch.bfloat16}[dtype]
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("DepthNet: compute dtype must be float32 or bfloat16")
        self.compute_dtype = dtype
        return self

    def _act_dtype(self, device):
        if self.compute_dtype == torch.bfloat16:
            return torch.bfloat16
        if device.type == "cuda" and torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16:
            return torch.bfloat16
        return torch.float32

    # constants used when the blend weights are not trainable (normalization.py:33-35)
    def _consts(self, device):
        key = str(device)
        if key not in self._const_cache:
            self._const_cache[key] = {
                "alpha_gamma": Var(torch.full((1,), float(self.cfg["norm_gamma"]), device=device)),
                "alpha_beta": Var(torch.full((1,), float(self.cfg["norm_beta"]), device=device)),
            }
        return self._const_cache[key]

    def _prepack_for(self, device, act_dtype):
        """This module's graph.Prepack for (device, activation dtype): the packed kernels of a training step, refilled in
        two launches (the buffers live with the module; an nn.DataParallel replica gets its own)."""
        if not hasattr(self, "_prepack"):
            object.__setattr__(self, "_prepack", {})
        key = (str(device), str(act_dtype))
        if key not in self._prepack:
            self._prepack[key] = graph.Prepack()
        return self._prepack[key]

    def _resolve_params(self):
        """This module's (or this replica's) weight tensors, in ``_param_names`` order."""
        out = []
        for path in self._param_paths:
            mod = self
            for part in path[:-1]:
                mod = mod._modules[part]
            out.append(getattr(mod, path[-1]))
        return out

    def forward(self, input, depthMap, depthMask):
        with _device_guard(input):
            return self._forward(input, depthMap, depthMask)

    def _forward(self, input, depthMap, depthMask):
        params = self._resolve_params()
        for t, nm in ((depthMap, "depthMap"), (depthMask, "depthMask")):
            if t.device != input.device:
                raise ValueError("DepthNet: %s is on %s, input on %s" % (nm, t.device, input.device))
        if params and params[0].device != input.device:
            raise ValueError("DepthNet: parameters on %s, input on %s" % (params[0].device, input.device))
        for t, nm in ((input, "input"), (depthMap, "depthMap"), (depthMask, "depthMask")):
            if t.dtype != torch.float32:
                raise TypeError("DepthNet: %s must be float32" % nm)
        if input.dim() != 4 or depthMap.dim() != 4 or depthMask.dim() != 4:
            raise ValueError("DepthNet: expected 4-D NCHW tensors")
        # masks prepared on the device (dasr_amd.prep.depth_to_masks) carry their region bytes: no compression pass,
        # no host read-back of the one-hot flag
        region = graph.attached_region(depthMask)
        act_dtype = self._act_dtype(input.device)
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _DepthNetFunction.apply(self, act_dtype, input, depthMap, depthMask, region, *params)
        tape = Tape(enabled=False, act_dtype=act_dtype)
        if not self.training:                 # netG.eval() + no_grad (F_model_depthCond.test): fold weights once
            if not hasattr(self, "_fold_cache"):
                object.__setattr__(self, "_fold_cache", {})
            tape.fold_cache = self._fold_cache
        P = {name: Var(p.detach(), False, name) for name, p in zip(self._param_names, params)}
        out = graph.depthnet_forward(tape, P, self.cfg, self._consts(input.device), input.detach().contiguous(),
                                     depthMap.detach().contiguous(), depthMask.detach().contiguous(), region)
        return out.data
