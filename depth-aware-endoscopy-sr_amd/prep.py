"""Device-side input preparation: the depth map is the only per-frame side input that has to
cross PCIe; the K depth-range planes and the region bytes the one-hot kernels read are derived from it on the GPU.

Reference counterparts: ``LQGTker_Depth_dataset.getDepthMask`` (codes/data/LQGTker_Depth_dataset.py:204-225), the
``_disp.npy`` reader (``:152-154``) and the tensor packing (``:187-199``)."""
import numpy as np
import torch

from . import ops


def load_disp_npy(path):
    """``name_disp.npy`` as monodepth2 wrote it, ``[1,1,h,w]`` -> float32 ``[1,h,w]`` (LQGTker_Depth_dataset.py:152-154)."""
    d = np.load(path)
    if d.ndim == 4:
        d = d.squeeze(1)
    return torch.from_numpy(np.ascontiguousarray(d)).float()


def fixed_range_edges(num_masks, device):
    """Edges of the ``depthFixedRange: true`` mode: the reference evaluates ``0 + ((1-0)/K)*i`` in Python doubles
    and torch compares the float32 map against each edge rounded to float32."""
    interval = (1 - 0) / num_masks
    return torch.tensor([0 + interval * i for i in range(num_masks + 1)], dtype=torch.float64).to(torch.float32).to(device)


def depth_to_masks(depth, num_masks=10, fixed_range=False):
    """``depth`` ``[B,1,h,w]`` (or ``[B,h,w]``) float32 on the GPU -> ``DepthMaskList`` ``[B,K,h,w]`` float32, the
    tensor ``DepthNet.forward`` / the losses take.  The returned tensor carries the region bytes of the same masks
    (``_dasr_region``), which ``DepthNet`` and ``harness.fused_losses`` pick up instead of compressing the planes
    again (and without reading the one-hot flag back to the host)."""
    if depth.dtype != torch.float32:
        raise TypeError("depth_to_masks: depth must be float32")
    edges = fixed_range_edges(num_masks, depth.device) if fixed_range else None
    planes, region = ops.depth_to_masks(depth, num_masks, edges, want_planes=True)
    planes._dasr_region = region
    planes._dasr_version = ops.tensor_version(planes)     # the shortcut is dropped if the tensor is edited in place afterwards
    return planes


def depth_to_region(depth, num_masks=10, fixed_range=False):
    """Region bytes only (``[B,h,w]`` uint8; ``num_masks`` = "no bin")."""
    edges = fixed_range_edges(num_masks, depth.device) if fixed_range else None
    return ops.depth_to_masks(depth, num_masks, edges, want_planes=False)[1]


def attach_region(masks):
    """Give a mask tensor that did NOT come from ``depth_to_masks`` (e.g. the reference's CPU dataloader output moved to
    the GPU) its region bytes, so that the generator and the fused loss take the one-hot kernels without ever reading
    a flag back mid-step.  Costs one compression pass and ONE 4-byte read-back, here, before any of the step's work is
    queued.  Returns True when the masks are one-hot (region attached), False for soft / overlapping masks."""
    from . import graph
    if graph.attached_region(masks) is not None:
        return True
    if not masks.is_cuda or masks.dtype != torch.float32 or not masks.is_contiguous():
        return False
    region, flag = ops.mask_compress(masks)
    if int(flag.item()) != 0:
        return False
    masks._dasr_region = region
    masks._dasr_version = ops.tensor_version(masks)
    return True
