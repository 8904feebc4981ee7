"""CPU oracle for the DepthNet hot path — TEST INFRASTRUCTURE, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file; the product path (``dasr_amd``) never does.

What it is: a plain-PyTorch (CPU, fp32 or fp64) restatement of the reference
generator **with the op graph as the reference writes it** — 256-channel style
map built by expand/permute/matmul, 3x3 convolutions over it, two instance norms
in a row, weight-norm recomputed each forward, a Python loop over depth regions
in the pooling and in the loss.  It is functional: it takes a ``state_dict`` with
the reference's key names and a config dict, no ``nn.Module``.

Parity status: the reference repository holds no tests, golden vectors or
fixtures for this path, so the oracle is pinned by outputs of
the reference itself, run in the build container: ``oracle/make_golden.py``
imports ``/root/reference/codes/models/modules/{sftmd_arch,normalization,mask_loss}.py``
and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file
against those vectors on every run.  Primitive arithmetic (conv2d, instance_norm,
interpolate, pixel_shuffle, ...) is PyTorch's, as it is in the reference
(requirements.txt:41 pins torch==1.6.0; torch 2.10 CPU kernels are what run here).

Each function cites the reference lines it follows.
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------

DEFAULT_CFG = dict(
    which_ResBlk_depth=list(range(0, 14)),  # options/train/train_depthNet_SEAN_depthMask_x8.yml:54
    in_nc=3, out_nc=3, nf=64, nb=16, scale=8,
    depth_latent_ch=256, depthRangeNum=10,
    use_trainable_params=True, norm_gamma=0.1, norm_beta=0.1,
    out_min=0.0, out_max=1.0,
)


def make_cfg(**kw):
    cfg = dict(DEFAULT_CFG)
    cfg.update(kw)
    return cfg


def block_plan(cfg):
    """Channel plan and block kinds of DepthNet.__init__ (sftmd_arch.py:879-889).

    Returns a list of ``(module_name, kind, channels)`` for blocks ``i = 0..nb-1``.
    """
    nb, scale = cfg["nb"], cfg["scale"]
    num_last_block = 1 if scale == 3 else int(math.log(scale, 2))
    plan = []
    for i in range(nb):
        ch = cfg["nf"]
        if i > nb - num_last_block:
            ch = 32
        if i in cfg["which_ResBlk_depth"]:
            plan.append(("depth-residual%d" % (i + 1), "depth", ch))
        else:
            plan.append(("classic-residual%d" % (i + 1), "classic", ch))
    return plan


# ----------------------------------------------------------------------------------------
# optional bf16 storage model (checker for the product's mixed-precision path)
# ----------------------------------------------------------------------------------------
# The reference itself is fp32.  Run under ``torch.autocast(bfloat16)`` it would keep its activations in bf16; the HIP
# path's bf16 mode stores exactly these tensors in bf16 (and rounds the trunk kernels once) while computing in fp32.
# ``with bf16_storage():`` makes this restatement round at the same points - forward values AND the gradients that
# flow back through them - so that the bf16 kernels can be compared against it tightly instead of only through a
# PSNR budget.  Off by default: every golden-vector test runs the plain fp32 / fp64 graph.
_BF16_STORAGE = False
# study switches (tools/bf16_model_study.py): tensors the model keeps in fp32 although the HIP path stores them in bf16 -
# "block_out" (the residual stream: every block's output), "conv_out" (the DGB conv outputs that feed the instance norms),
# "gb2" (gamma_o / beta_o).  Empty = the shipping configuration.
BF16_KEEP_FP32 = set()


class _RoundBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _r(x, tag=None):
    """An activation tensor that the bf16 path keeps in HBM."""
    if tag is not None and tag in BF16_KEEP_FP32:
        return x
    return _RoundBf16.apply(x) if _BF16_STORAGE else x


def _rw(w):
    """A trunk kernel: rounded once when packed; the weight gradient stays fp32."""
    return w + (w.to(torch.bfloat16).to(w.dtype) - w).detach() if _BF16_STORAGE else w


class bf16_storage:
    def __enter__(self):
        global _BF16_STORAGE
        self.prev = _BF16_STORAGE
        _BF16_STORAGE = True

    def __exit__(self, *exc):
        global _BF16_STORAGE
        _BF16_STORAGE = self.prev


# ----------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------

def wn_weight(sd, prefix):
    """``torch.nn.utils.weight_norm`` (dim=0): ``w = g * v / ||v||`` over all dims but 0
    (call sites sftmd_arch.py:741,851; for ConvTranspose2d dim 0 is the IN-channel axis)."""
    return torch._weight_norm(sd[prefix + ".weight_v"], sd[prefix + ".weight_g"], 0)


def wn_conv(sd, prefix, x, stride=1, padding=1, trunk=False):
    w = wn_weight(sd, prefix)
    return F.conv2d(x, _rw(w) if trunk else w, sd[prefix + ".bias"], stride=stride, padding=padding)


def plain_conv(sd, prefix, x, padding=1, trunk=False):
    w = sd[prefix + ".weight"]
    return F.conv2d(x, _rw(w) if trunk else w, sd[prefix + ".bias"], stride=1, padding=padding)


def region_avg_pool(feature_map, mask):
    """RegionWiseAvgPooling.forward (sftmd_arch.py:714-733)."""
    if mask.size(2) != feature_map.size(2) or mask.size(3) != feature_map.size(3):
        mask = F.interpolate(mask, size=(feature_map.size(2), feature_map.size(3)), mode="bilinear",
                             align_corners=True)
        mask = (mask >= 0.5).type_as(mask)
    out = []
    for i in range(mask.size(1)):
        region = torch.cat([mask[:, i].unsqueeze(1)] * feature_map.size(1), dim=1)
        s_feat = torch.sum(region * feature_map, dim=(2, 3))
        s_mask = torch.sum(region, dim=(2, 3))
        out.append((s_feat / (s_mask + 1e-10)).unsqueeze(1))
    return torch.cat(out, dim=1)


def encoder(sd, x, depth_mask, is_baseline=False):
    """Encoder.forward (sftmd_arch.py:771-783)."""
    act = lambda t: F.leaky_relu(t, 0.2)
    out = wn_conv(sd, "encoder.layer1", x)
    feat = out
    if is_baseline:
        return act(feat), None
    out = wn_conv(sd, "encoder.layer2", act(out), stride=2)
    out = wn_conv(sd, "encoder.layer3", act(out), stride=2)
    out = F.conv_transpose2d(act(out), wn_weight(sd, "encoder.layer4"), sd["encoder.layer4.bias"],
                             stride=2, padding=1)
    out = wn_conv(sd, "encoder.layer5", act(out), stride=2)
    return act(feat), region_avg_pool(out, depth_mask)


def sean(sd, prefix, x, depth_map, depth_mask, st, cfg):
    """SEAN.forward, as written (normalization.py:52-92), non-ablated branch."""
    assert cfg["depth_latent_ch"] == st.size(2) and st.size(1) == depth_mask.size(1)
    normalized = F.instance_norm(x, eps=1e-5)                      # :56 (param_free_norm)
    depth_map = F.interpolate(depth_map, size=x.size()[2:], mode="nearest")    # :58
    depth_mask = F.interpolate(depth_mask, size=x.size()[2:], mode="nearest")  # :59
    actv = _r(F.relu(plain_conv(sd, prefix + ".mlp_mask.0", depth_map)))       # :61
    beta_o = _r(plain_conv(sd, prefix + ".mlp_beta_o", actv, trunk=True), "gb2")      # :73
    gamma_o = _r(plain_conv(sd, prefix + ".mlp_gamma_o", actv, trunk=True), "gb2")    # :74
    st = F.conv2d(st.unsqueeze(3), sd[prefix + ".A_i_j.weight"], sd[prefix + ".A_i_j.bias"])  # :80
    st = st.expand(st.size(0), st.size(1), st.size(2), depth_mask.size(3)).permute(0, 3, 2, 1)  # :81
    style_map = st.matmul(depth_mask.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)                   # :82
    beta_s = plain_conv(sd, prefix + ".mlp_beta_s", style_map)                 # :84
    gamma_s = plain_conv(sd, prefix + ".mlp_gamma_s", style_map)               # :85
    if cfg["use_trainable_params"]:
        a_g, a_b = sd[prefix + ".alpha_gamma"], sd[prefix + ".alpha_beta"]
    else:
        a_g, a_b = cfg["norm_gamma"], cfg["norm_beta"]
    gamma = a_g * gamma_s + (1.0 - a_g) * gamma_o                              # :87
    beta = a_b * beta_s + (1.0 - a_b) * beta_o                                 # :88
    return normalized * (1 + gamma) + beta                                     # :89


def depth_block(sd, name, x, depth_map, depth_mask, st, cfg):
    """Depth_Residual_Block_Mask.forward (sftmd_arch.py:826-834); conv1/conv2 are
    ``Sequential(Conv2d, InstanceNorm2d(affine=False))`` (:811-820)."""
    t = F.instance_norm(_r(plain_conv(sd, name + ".conv1.0", x, trunk=True), "conv_out"), eps=1e-5)
    a = _r(F.relu(sean(sd, name + ".norm1", t, depth_map, depth_mask, st, cfg)))
    t = F.instance_norm(_r(plain_conv(sd, name + ".conv2.0", a, trunk=True), "conv_out"), eps=1e-5)
    return _r(F.relu(x + sean(sd, name + ".norm2", t, depth_map, depth_mask, st, cfg)), "block_out")


def classic_block(sd, name, x):
    """Classic_Residual_Block.forward, weight-norm variant (sftmd_arch.py:131-151)."""
    f = wn_conv(sd, name + ".block.2", _r(F.relu(wn_conv(sd, name + ".block.0", x, trunk=True))), trunk=True)
    return _r(F.relu(x + f), "block_out")


def upscale(sd, name, x, r, second_conv):
    """upscale1/2/3 (sftmd_arch.py:891-908): wn conv -> PixelShuffle(r) -> LeakyReLU(0.2)
    [-> wn conv -> LeakyReLU(0.2)]."""
    x = _r(F.leaky_relu(F.pixel_shuffle(wn_conv(sd, name + ".0", x, trunk=True), r), 0.2))
    if second_conv:
        x = _r(F.leaky_relu(wn_conv(sd, name + ".3", x, trunk=True), 0.2))
    return x


def depthnet_forward(sd, cfg, inp, depth_map, depth_mask):
    """DepthNet.forward (sftmd_arch.py:912-950)."""
    plan = block_plan(cfg)
    nb, scale = cfg["nb"], cfg["scale"]
    is_baseline = len(cfg["which_ResBlk_depth"]) == 0
    feat, st = encoder(sd, inp, depth_mask, is_baseline)
    feat = _r(feat)               # fp32 encoder -> bf16 trunk boundary
    fea_bef = _r(F.leaky_relu(wn_conv(sd, "head.2", _r(F.leaky_relu(wn_conv(sd, "head.0", feat, trunk=True), 0.2)),
                                      trunk=True), 0.2))

    def run_block(i, x):
        name, kind, _ = plan[i]
        if kind == "depth":
            return depth_block(sd, name, x, depth_map, depth_mask, st, cfg)
        return classic_block(sd, name, x)

    fea = fea_bef
    for i in range(nb - 3):                      # :923 (block index nb-3 is never called)
        fea = run_block(i, fea)
    fea = _r(fea + fea_bef)                      # :931
    if scale == 8:
        fea = upscale(sd, "upscale1", fea, 2, True)
    fea = run_block(nb - 2, fea)                 # :934-937
    if scale >= 4:
        fea = upscale(sd, "upscale2", fea, 2, True)
    fea = run_block(nb - 1, fea)                 # :941-944
    fea = upscale(sd, "upscale3", fea, 3 if scale == 3 else 2, False)
    out = F.conv2d(fea, sd["conv_output.weight"], sd["conv_output.bias"], padding=4)   # :948
    return torch.clamp(out, min=cfg["out_min"], max=cfg["out_max"])                    # :950


# ----------------------------------------------------------------------------------------
# parameter shapes (for building a state_dict without the reference)
# ----------------------------------------------------------------------------------------

def param_shapes(cfg):
    """Ordered ``{key: shape}`` of DepthNet's state_dict (reference: sftmd_arch.py:838-910,
    normalization.py:8-49; key list dumped in SURVEY.md §8b)."""
    L, K, nf = cfg["depth_latent_ch"], cfg["depthRangeNum"], cfg["nf"]
    scale = cfg["scale"]
    shapes = {}

    def wn(prefix, cout, cin, k=3, transposed=False):
        shapes[prefix + ".bias"] = (cout,)
        if transposed:
            shapes[prefix + ".weight_g"] = (cin, 1, 1, 1)
            shapes[prefix + ".weight_v"] = (cin, cout, k, k)
        else:
            shapes[prefix + ".weight_g"] = (cout, 1, 1, 1)
            shapes[prefix + ".weight_v"] = (cout, cin, k, k)

    def plain(prefix, cout, cin, k=3):
        shapes[prefix + ".weight"] = (cout, cin, k, k)
        shapes[prefix + ".bias"] = (cout,)

    wn("encoder.layer1", 32, cfg["in_nc"])
    wn("encoder.layer2", 64, 32)
    wn("encoder.layer3", 128, 64)
    wn("encoder.layer4", L, 128, transposed=True)
    wn("encoder.layer5", L, L)
    wn("head.0", 64, 32)
    wn("head.2", 64, 64)
    for name, kind, ch in block_plan(cfg):
        if kind == "depth":
            for norm in ("norm1", "norm2"):
                p = "%s.%s" % (name, norm)
                if cfg["use_trainable_params"]:
                    shapes[p + ".alpha_beta"] = (1,)
                    shapes[p + ".alpha_gamma"] = (1,)
                plain(p + ".A_i_j", K, K, 1)
                plain(p + ".mlp_gamma_s", ch, L)
                plain(p + ".mlp_beta_s", ch, L)
                plain(p + ".mlp_mask.0", 2 * ch, 1)
                plain(p + ".mlp_gamma_o", ch, 2 * ch)
                plain(p + ".mlp_beta_o", ch, 2 * ch)
            plain(name + ".conv1.0", ch, ch)
            plain(name + ".conv2.0", ch, ch)
        else:
            wn(name + ".block.0", ch, ch)
            wn(name + ".block.2", ch, ch)
    ch_last2 = 64 if scale == 4 else 32
    ch_last = 64 if scale < 4 else 32
    fs = 3 if scale == 3 else 2
    wn("upscale1.0", 256, 64)
    wn("upscale1.3", 32, 64)
    wn("upscale2.0", 128, ch_last2)
    wn("upscale2.3", 32, 32)
    wn("upscale3.0", 32 * fs * fs, ch_last)
    plain("conv_output", cfg["out_nc"], 32, 9)
    return shapes


def new_state_dict(cfg, dtype=torch.float32):
    return {k: torch.zeros(s, dtype=dtype) for k, s in param_shapes(cfg).items()}


# ----------------------------------------------------------------------------------------
# losses, metric, optimiser schedule (harness rows 11-12 of SURVEY.md §8a)
# ----------------------------------------------------------------------------------------

def dynamic_mask_loss(sr, hr, mask_list, trainable_weight, dynamic_weight=10.0):
    """dynamic_weight_mask_loss.forward with the 'smoothl1' criterion
    (codes/models/modules/mask_loss.py:64-90). Returns (per_region_losses, weighted_total, softmax_w)."""
    K = mask_list.shape[1]
    sm = F.softmax(trainable_weight, dim=0)
    losses, weighted = [], []
    for i in range(K):
        m = F.interpolate(mask_list[:, i].unsqueeze(1), size=sr.size()[2:], mode="nearest")
        m = torch.cat([m, m, m], dim=1)
        l = F.smooth_l1_loss(m * sr, m * hr, reduction="none").sum() / m.sum()
        losses.append(l)
        weighted.append(sm[i] * l)
    return losses, sum(weighted) * dynamic_weight, sm


def total_loss(sr, hr, mask_list, trainable_weight, pixel_weight=1.0, dynamic_weight=10.0):
    """l_pix + l_dynamic as in F_Model_depthCond.optimize_parameters
    (codes/models/F_model_depthCond.py:163-190) with the shipped yml switches
    (options/train/train_depthNet_SEAN_depthMask_x8.yml:84-115: L1 pixel loss weight 1,
    dynamic smooth-L1 loss weight 10, all other losses off)."""
    l_pix = pixel_weight * F.l1_loss(sr, hr)
    per, l_dyn, sm = dynamic_mask_loss(sr, hr, mask_list, trainable_weight, dynamic_weight)
    return l_pix + l_dyn, l_pix, l_dyn, per


def cosine_restart_lr(step, base_lr=1e-3, T_period=(20000, 20000, 20000, 20000),
                      restarts=(20000, 40000, 60000), weights=(1, 1, 1), eta_min=1e-7):
    """Closed form of CosineAnnealingLR_Restart (codes/models/lr_scheduler.py:34-62) after
    ``step`` calls of ``scheduler.step()`` (the recursion there telescopes to this)."""
    last_restart, T, w = 0, T_period[0], 1.0
    for j, r in enumerate(restarts):
        if step >= r:
            last_restart, T, w = r, T_period[j + 1], weights[j]
    return eta_min + (base_lr * w - eta_min) * (1 + math.cos(math.pi * (step - last_restart) / T)) / 2


def psnr_255(a, b):
    """calculate_psnr (codes/utils/util.py:646-653) on [0,1] tensors scaled to [0,255]."""
    mse = torch.mean((a.double() * 255.0 - b.double() * 255.0) ** 2).item()
    if mse == 0:
        return float("inf")
    return 20.0 * math.log10(255.0 / math.sqrt(mse))


# ---- SSIM (validation metric): CPU restatement of pytorch_ssim/__init__.py:7-37 as written (2-D window = g g^T, five grouped
# convolutions), the checker for dasr_ssim; pinned by tests/golden/ssim.npz (generated from the reference's own module)
def ssim_ref(img1, img2, window_size=11, size_average=True):
    import math as _math
    g = torch.tensor([_math.exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)],
                     dtype=torch.float32)
    g = (g / g.sum()).unsqueeze(1)
    channel = img1.shape[1]
    window = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(channel, 1, window_size, window_size).contiguous()
    window = window.to(img1.device).type_as(img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)
