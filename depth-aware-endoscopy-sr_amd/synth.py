"""Synthetic frames and closed-form parameter fill.

Host-side data contract of the hot path.  The reference feeds the generator with
``LQ [3,H,W]``, ``GT [3,sH,sW]``, ``Depth [1,h,w]`` and ``DepthMaskList [K,h,w]``
float32 tensors (reference: codes/data/LQGTker_Depth_dataset.py:187-199); the
masks come from ``getDepthMask`` (same file, :204-225): K equal-width half-open
bins over the depth map's own min/max.  There is no dataset offline, so the
bench, smoke test and parity tests all draw frames from here.

Nothing in this file touches the GPU; tensors are created on the CPU and moved
by the caller.
"""
import math
import zlib

import torch
import torch.nn.functional as F

__all__ = [
    "depth_to_masks",
    "seeded_frame",
    "seeded_batch",
    "closed_form_frame",
    "closed_form_batch",
    "closed_form_fill_",
]


_MASK64 = (1 << 64) - 1


def hash_uniform(n: int, key: str) -> torch.Tensor:
    """``n`` reproducible pseudo-random float64 values in [0,1): splitmix64 of ``(crc32(key) << 32) + i``.
    Pure integer arithmetic, so every machine regenerates identical values (no RNG state involved)."""
    import numpy as np
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64((zlib.crc32(key.encode()) << 32) & _MASK64))
        z = (z + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return torch.from_numpy((z >> np.uint64(11)).astype(np.float64) / float(1 << 53))


def depth_to_masks(depth: torch.Tensor, num_masks: int = 10, fixed_range: bool = False) -> torch.Tensor:
    """Binary depth-range masks ``[K,h,w]`` from one depth map ``[1,h,w]`` (or ``[h,w]``).

    Restates the binning rule of the reference's ``getDepthMask``
    (codes/data/LQGTker_Depth_dataset.py:204-225): bin i is
    ``[min + i*delta, min + (i+1)*delta)`` with ``delta = (max-min)/K`` evaluated in
    float32; a pixel equal to the map's maximum falls into no bin.
    """
    d = depth.reshape(depth.shape[-2], depth.shape[-1]).to(torch.float32)
    if fixed_range:
        # the reference sets min/max to the Python ints 0 and 1: edges are Python doubles, and torch compares the
        # float32 map against each of them rounded to float32
        interval = (1 - 0) / num_masks
        edges = torch.tensor([0 + interval * i for i in range(num_masks + 1)], dtype=torch.float64).to(torch.float32)
        starts, ends = edges[:-1].view(-1, 1, 1), edges[1:].view(-1, 1, 1)
    else:
        lo = d.min()
        hi = d.max()
        delta = (hi - lo) / num_masks
        idx = torch.arange(num_masks, dtype=torch.float32)
        starts = (lo + delta * idx).view(-1, 1, 1)
        ends = (lo + delta * (idx + 1.0)).view(-1, 1, 1)
    return ((d.unsqueeze(0) >= starts) & (d.unsqueeze(0) < ends)).to(torch.float32)


def _all_bins_populated(masks: torch.Tensor) -> bool:
    return bool((masks.flatten(1).sum(1) > 0).all())


def seeded_frame(frame_idx: int, H: int, W: int, scale: int, num_masks: int = 10):
    """One synthetic frame, seeded by its global index.

    Returns ``(LQ [3,H,W], GT [3,sH,sW], Depth [1,H,W], Masks [K,H,W])``; every
    depth bin is guaranteed non-empty (the reference's dynamic loss divides by the
    region area, codes/models/modules/mask_loss.py:81-83).
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(1234 + int(frame_idx))
    lq = torch.rand(3, H, W, generator=g)
    z = torch.randn(1, 1, H, W, generator=g)
    k = 15
    blur = F.avg_pool2d(z, k, stride=1, padding=k // 2, count_include_pad=False)
    blur = blur * float(k)  # back to ~unit variance
    depth = (0.01 + 9.99 * torch.sigmoid(2.0 * blur)).reshape(1, H, W)
    masks = depth_to_masks(depth, num_masks)
    if not _all_bins_populated(masks):
        ramp = torch.linspace(0.0, 1.0, H * W).reshape(1, H, W)
        depth = 0.01 + 9.99 * (0.9 * ramp + 0.1 * torch.rand(1, H, W, generator=g))
        masks = depth_to_masks(depth, num_masks)
    gt = torch.rand(3, H * scale, W * scale, generator=g)
    return lq, gt, depth, masks


def seeded_batch(first_idx: int, batch: int, H: int, W: int, scale: int, num_masks: int = 10):
    """Stack ``batch`` seeded frames: ``LQ [B,3,H,W], GT, Depth [B,1,H,W], Masks [B,K,H,W]``."""
    frames = [seeded_frame(first_idx + i, H, W, scale, num_masks) for i in range(batch)]
    return tuple(torch.stack([f[j] for f in frames]) for j in range(4))


def closed_form_frame(frame_idx: int, H: int, W: int, scale: int, num_masks: int = 10):
    """RNG-free frame used by the golden fixtures (integer-hash noise for LQ/GT, a smooth sine field for
    the depth map so that the depth regions are contiguous), so that the fixture files only have to
    store expected outputs."""
    y = torch.arange(H, dtype=torch.float32).view(1, H, 1)
    x = torch.arange(W, dtype=torch.float32).view(1, 1, W)
    c = torch.arange(3, dtype=torch.float32).view(3, 1, 1)
    f = float(frame_idx)
    lq = hash_uniform(3 * H * W, "lq%d" % frame_idx).reshape(3, H, W).to(torch.float32)
    depth = 0.01 + 9.99 * (0.5 + 0.5 * torch.sin(0.23 * x + 0.31 + 0.2 * f) * torch.cos(0.19 * y + 0.1 * f)
                           * torch.cos(0.05 * x * y / max(H, W) + 0.3))
    depth = depth.reshape(1, H, W)
    masks = depth_to_masks(depth, num_masks)
    ys = torch.arange(H * scale, dtype=torch.float32).view(1, -1, 1)
    xs = torch.arange(W * scale, dtype=torch.float32).view(1, 1, -1)
    gt = (0.25 * (1.0 + torch.sin(0.11 * xs + 0.07 * ys + 1.1 * c + 0.9 * f) * torch.cos(0.05 * xs - 0.03 * ys))
          + 0.5 * hash_uniform(3 * H * W * scale * scale, "gt%d" % frame_idx).reshape(3, H * scale, W * scale)
          ).to(torch.float32)
    return lq, gt, depth, masks


def closed_form_batch(first_idx: int, batch: int, H: int, W: int, scale: int, num_masks: int = 10):
    frames = [closed_form_frame(first_idx + i, H, W, scale, num_masks) for i in range(batch)]
    return tuple(torch.stack([f[j] for f in frames]) for j in range(4))


def _phase(name: str) -> float:
    return 2.0 * math.pi * float(zlib.crc32(name.encode()) % 4096) / 4096.0


@torch.no_grad()
def closed_form_fill_(named_tensors, gain: float = 1.0):
    """Fill parameters in place with RNG-free integer-hash noise of role-dependent amplitude (SURVEY.md §8d
    asks for a closed-form fill both sides can regenerate; a pure sine fill makes some instance-norm
    channels nearly constant, which turns the parity check into a test of rounding noise).

    ``named_tensors`` is an iterable of ``(name, tensor)`` (``state_dict().items()`` or
    ``named_parameters()``).  Amplitudes follow the tensor's role so that a randomly
    "initialised" net produces outputs spread over (0,1) instead of saturating the
    final clamp: conv weights ~ 1/sqrt(fan_in), weight-norm gains near 1, small biases,
    ``alpha_gamma = 0.7``, ``alpha_beta = 0.74``, output bias 0.5.
    """
    for name, t in named_tensors:
        n = t.numel()
        wave = (2.0 * hash_uniform(n, name) - 1.0)        # uniform in [-1, 1)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "alpha_gamma":
            vals = torch.full((n,), 0.7, dtype=torch.float64)
        elif leaf == "alpha_beta":
            vals = torch.full((n,), 0.74, dtype=torch.float64)
        elif leaf == "weight_g":
            vals = 1.0 + 0.25 * wave
        elif leaf == "bias":
            vals = 0.05 * wave
            if name.startswith("conv_output"):
                vals = 0.5 + 0.55 * wave
        elif leaf == "trainable_weight":
            vals = 1.0 + 0.1 * wave
        else:  # convolution kernels (weight / weight_v)
            fan_in = max(1, n // t.shape[0])
            amp = gain * math.sqrt(6.0) / math.sqrt(fan_in)   # uniform(-a,a) with variance 2/fan_in
            if name.startswith("conv_output"):
                amp *= 0.75
            vals = amp * wave
        t.copy_(vals.to(t.dtype).reshape(t.shape))
    return named_tensors
