"""Validation path: the reference's `test()` + `tensor2img` + cropped PSNR + SSIM loop
(codes/train.py:219-271, models/F_model_depthCond.py:228-234, utils/util.py:566-590,646-653,
pytorch_ssim/__init__.py:7-37,65-73) for one DepthNet.  The forward is the inference plan of `DepthNet`
(`net.eval()` + `torch.no_grad()`: no tape, weights folded once); the metrics are computed where the images are."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def tensor2img(tensor, out_type=np.uint8, min_max=(0, 1)):
    """utils/util.py:566-590 for 3-D (C,H,W) and 2-D (H,W) tensors: clamp, rescale to [0,1], RGB->BGR, HWC,
    `round(x*255)` for uint8.  (The 4-D branch builds a grid with torchvision, which the validation loop never
    reaches: `visuals['SR']` is one image.)"""
    tensor = tensor.squeeze().float().cpu().clamp(*min_max)
    tensor = (tensor - min_max[0]) / (min_max[1] - min_max[0])
    if tensor.dim() == 3:
        img = np.transpose(tensor.numpy()[[2, 1, 0], :, :], (1, 2, 0))
    elif tensor.dim() == 2:
        img = tensor.numpy()
    else:
        raise TypeError("tensor2img: only 3-D and 2-D tensors are supported here, got %d-D" % tensor.dim())
    if out_type == np.uint8:
        img = (img * 255.0).round()
    return img.astype(out_type)


def calculate_psnr(img1, img2):
    """utils/util.py:646-653, images in [0,255] (numpy arrays or tensors)."""
    a = np.asarray(img1.cpu() if torch.is_tensor(img1) else img1, dtype=np.float64)
    b = np.asarray(img2.cpu() if torch.is_tensor(img2) else img2, dtype=np.float64)
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(mse))


def _window_1d(window_size=11, sigma=1.5):
    """The reference's 1-D Gaussian, built the way pytorch_ssim builds it (fp32 tensor, normalised in fp32)."""
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)],
                     dtype=torch.float32)
    return g / g.sum()


def ssim(img1, img2, window_size=11, size_average=True):
    """pytorch_ssim.ssim (pytorch_ssim/__init__.py:17-37,65-73): 11x11 Gaussian (sigma 1.5) windows, zero padding,
    C1 = 0.01^2, C2 = 0.03^2 on [0,1] images `[B,C,H,W]` - one HIP kernel pass over both images (dasr_ssim: the five
    windowed moments, the SSIM ratio and its mean), no torch compute ops."""
    import ctypes
    from . import _lib, ops
    if window_size != 11:
        raise NotImplementedError("dasr_amd.validate.ssim: the reference's 11 x 11 window only")
    a, b = img1.contiguous().float(), img2.contiguous().float()
    B, C, H, W = a.shape
    out = torch.empty((B,), dtype=torch.float32, device=a.device)
    nbytes = int(_lib.get().dasr_ssim_workspace(B, C, H, W))
    ws = torch.empty((max(1, nbytes // 4),), dtype=torch.float32, device=a.device)
    g = (ctypes.c_float * 11)(*[float(v) for v in _window_1d()])
    ops._call("dasr_ssim", ops._p(a), ops._p(b), ctypes.cast(g, ctypes.c_void_p), ops._p(out), ops._p(ws), nbytes, B, C, H, W)
    return out.mean() if size_average else out


@torch.no_grad()
def test(net, lq, depth, masks):
    """F_Model_depthCond.test (F_model_depthCond.py:228-234): eval-mode forward without autograd, back to train()."""
    was_training = net.training
    net.eval()
    try:
        return net(lq, depth, masks)
    finally:
        if was_training:
            net.train()


def validate(net, frames, scale):
    """The validation loop of codes/train.py:219-262 over an iterable of `(LQ, GT, Depth, DepthMaskList)` frames
    (`[1,3,h,w]`, `[1,3,sh,sw]`, `[1,1,h,w]`, `[1,K,h,w]`): SSIM of the [0,1] tensors (pytorch_ssim), PSNR of the
    uint8 images with a `scale`-pixel border cropped.  Returns (avg_psnr, avg_ssim, n)."""
    psnr_sum = ssim_sum = 0.0
    n = 0
    for lq, gt, depth, masks in frames:
        sr = test(net, lq, depth, masks)
        ssim_sum += float(ssim(sr[:1].detach().float(), gt[:1].to(sr.device).float()))
        sr_img = tensor2img(sr[0]) / 255.0
        gt_img = tensor2img(gt[0]) / 255.0
        c = scale
        psnr_sum += calculate_psnr(sr_img[c:-c, c:-c, :] * 255, gt_img[c:-c, c:-c, :] * 255)
        n += 1
    return psnr_sum / max(n, 1), ssim_sum / max(n, 1), n
