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


def _gaussian_window(window_size, channel, like):
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)],
                     dtype=torch.float32)
    g = (g / g.sum()).unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous().to(like.device).type_as(like)


def ssim(img1, img2, window_size=11, size_average=True):
    """pytorch_ssim.ssim (pytorch_ssim/__init__.py:17-37,65-73): 11x11 Gaussian (sigma 1.5) windows, zero padding,
    C1 = 0.01^2, C2 = 0.03^2 on [0,1] images; `[B,C,H,W]` tensors on any device."""
    channel = img1.shape[1]
    window = _gaussian_window(window_size, channel, img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    if size_average:
        return ssim_map.mean()
    return ssim_map.mean(1).mean(1).mean(1)


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
