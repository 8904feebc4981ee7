"""Tensor-level wrappers of the C ABI (include/dasr.h).

PyTorch is used here for device memory (``torch.empty``) and the current stream only; every
computation is a call into libdasr_hip.so.  Activations are NHWC ``[B,H,W,C]``, kernels are HWIO
``[KH,KW,Cin,Cout]``.  fp32 is the default; a bf16 activation tensor selects the ``*_bf16`` entry
points (mixed-precision path: bf16 activations and trunk kernels, fp32 parameters / statistics /
accumulators) - the dtype of the tensors passed in decides, nothing is converted silently.
"""
import ctypes

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
IN_EPS = 1e-5  # nn.InstanceNorm2d default eps (sftmd_arch.py:813, normalization.py:17)


BF16 = torch.bfloat16


def _p(t, allow_none=False):
    return _lib.ptr(t, allow_none)


def _pa(t, allow_none=False):
    """Pointer of an activation tensor: float32 or bfloat16."""
    if t is None:
        return _lib.ptr(t, allow_none)
    return _lib.ptr(t, allow_none, dtype=t.dtype if t.dtype in (torch.float32, BF16) else torch.float32)


def _is_bf(*ts):
    """True when the tensors that are given are bfloat16 (mixing is a caller bug)."""
    kinds = {t.dtype for t in ts if t is not None}
    if len(kinds) > 1:
        raise TypeError("dasr_amd: mixed activation dtypes %s" % sorted(str(k) for k in kinds))
    return kinds == {BF16}


def _call(name, *args):
    fn = getattr(_lib.get(), name)
    _lib.check(fn(*args, _lib.stream()), name)


def empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


# ---- edge ----------------------------------------------------------------------------------
def nchw_to_nhwc(x):
    B, C, H, W = x.shape
    out = empty((B, H, W, C), x)
    _call("dasr_nchw_to_nhwc", _p(x), _p(out), B, C, H, W)
    return out


def nhwc_to_nchw(x):
    B, H, W, C = x.shape
    out = empty((B, C, H, W), x)
    _call("dasr_nhwc_to_nchw", _p(x), _p(out), B, C, H, W)
    return out


def clamp_to_nchw(y, lo, hi):
    B, H, W, C = y.shape
    out = empty((B, C, H, W), y)
    _call("dasr_clamp_to_nchw", _p(y), _p(out), B, C, H, W, float(lo), float(hi))
    return out


def clamp_to_nchw_bwd(dout, y, lo, hi):
    B, H, W, C = y.shape
    dy = empty(y.shape, y)
    _call("dasr_clamp_to_nchw_bwd", _p(dout), _p(y), _p(dy), B, C, H, W, float(lo), float(hi))
    return dy


def resize_nearest_nchw(x, H, W):
    B, C, h, w = x.shape
    if (h, w) == (H, W):
        return x
    out = empty((B, C, H, W), x)
    _call("dasr_resize_nearest_nchw", _p(x), _p(out), B * C, h, w, H, W)
    return out


def resize_nearest_u8(region, H, W):
    B, h, w = region.shape
    if (h, w) == (H, W):
        return region
    out = torch.empty((B, H, W), dtype=torch.uint8, device=region.device)
    _call("dasr_resize_nearest_u8", _lib.ptr(region, dtype=torch.uint8), _lib.ptr(out, dtype=torch.uint8), B, h, w, H, W)
    return out


# ---- max |.| of a tensor, carried on the tensor (fp16 x 2 split convolutions) ---------------------------------------
# A producer kernel that is handed an amax buffer (dasr.h: DASR_AMAX_FLOATS floats - a count and one partial maximum per
# workgroup) fills it while it stores its output; the buffer then travels with the tensor as an attribute.  Every op that
# writes into an existing tensor drops it.
AMAX_FLOATS = 4097


def amax_buffer(like):
    """An (uninitialised) amax buffer on ``like``'s device: nothing to clear, the producer writes every word it declares."""
    return torch.empty((AMAX_FLOATS,), dtype=torch.float32, device=like.device)


def amax_value(buf):
    """Host-side read of an amax buffer (tests and tools only: the kernels never need it on the host)."""
    h = buf.detach().cpu()
    n = int(h[:1].view(torch.int32).item())
    assert 1 <= n < AMAX_FLOATS, n
    return float(h[1:1 + n].max().item())


def set_amax(t, amax):
    if amax is not None:
        t._dasr_amax = amax
    return t


def get_amax(t):
    return getattr(t, "_dasr_amax", None)


def drop_amax(t):
    if t is not None and getattr(t, "_dasr_amax", None) is not None:
        t._dasr_amax = None


def add(a, b):
    out = torch.empty_like(a)
    if _is_bf(a, b):
        _call("dasr_add_bf16", _pa(a), _pa(b), _pa(out), a.numel())
    else:
        _call("dasr_add", _p(a), _p(b), _p(out), a.numel())
    return out


def accumulate_(dst, src):
    assert dst.shape == src.shape
    drop_amax(dst)
    if dst.dtype == torch.float32 and src.dtype == BF16:       # bf16 gradient arriving at an fp32 tensor
        _call("dasr_cast_bf16_to_f32", _pa(src), _p(dst), 1, dst.numel())
    elif _is_bf(dst, src):
        _call("dasr_accumulate_bf16", _pa(dst), _pa(src), dst.numel())
    else:
        _call("dasr_accumulate", _p(dst), _p(src), dst.numel())
    return dst


def copy_(dst, src):
    assert dst.numel() == src.numel() and dst.dtype == src.dtype
    drop_amax(dst)
    if dst.dtype == BF16:
        assert dst.numel() % 2 == 0
        _call("dasr_copy", _pa(dst), _pa(src), dst.numel() // 2)      # a byte copy: two bf16 per float
    else:
        _call("dasr_copy", _p(dst), _p(src), dst.numel())
    return dst


def cast_to_bf16(x):
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    _call("dasr_cast_f32_to_bf16", _p(x), _pa(out), x.numel())
    return out


def cast_to_f32(x):
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _call("dasr_cast_bf16_to_f32", _pa(x), _p(out), 0, x.numel())
    return out


# ---- the encoder's stride-2 layers in stride-1 form (csrc/s2d.hip; bf16 path only) ----------------------------------
def set_conv_bf16_impl(impl):
    """0 (default): the persistent LDS-DMA kernel runs the bf16 trunk forward / dgrad where it applies; 1: the first
    (register-staged) kernel everywhere - for A/B measurements and tests."""
    _call("dasr_set_conv_bf16_impl", int(impl))


def conv_bf16_v2_launches():
    return int(_lib.get().dasr_conv_bf16_v2_launches())


def tensor_version(t):
    """torch's in-place edit counter of ``t``; tensors created under ``torch.inference_mode()`` do not track one
    (reading it raises) and are reported as version -1: they are treated as immutable."""
    try:
        return t._version
    except RuntimeError:
        return -1


def space_to_depth2(x, valid_hw=None):
    """[B,H,W,C] fp32 or bf16 -> bf16 [B,ceil(H/2),ceil(W/2),4C], channel (2py+px)C + c = x[2i+py][2j+px][c].
    ``valid_hw`` = (Hv, Wv): rows / columns of x at or beyond them are read as zeros (the PixelShuffle image of the
    encoder's transposed layer carries one row and one column the reference's layer does not have)."""
    B, H, W, C = x.shape
    Hv, Wv = valid_hw if valid_hw is not None else (H, W)
    out = torch.empty((B, (H + 1) // 2, (W + 1) // 2, 4 * C), dtype=BF16, device=x.device)
    _call("dasr_space_to_depth2_bf16", _pa(x), 1 if x.dtype == BF16 else 0, _pa(out), B, H, W, C, Hv, Wv)
    return out


def depth_to_space2_bwd(dy, x_shape, dtype, out=None, valid_hw=None):
    """Adjoint of space_to_depth2: bf16 gradient [B,Hs,Ws,4C] -> gradient of x (``dtype``), zero beyond ``valid_hw``;
    ``out``: accumulate into it."""
    B, H, W, C = x_shape
    Hv, Wv = valid_hw if valid_hw is not None else (H, W)
    acc = out is not None
    if out is None:
        out = torch.empty(tuple(x_shape), dtype=dtype, device=dy.device)
    drop_amax(out)
    _call("dasr_depth_to_space2_bwd_bf16", _pa(dy), _pa(out), 1 if out.dtype == BF16 else 0, 1 if acc else 0, B, H, W, C,
          Hv, Wv)
    return out


def weight_expand_s2(w_packed):
    """fp32 packed kernel (2,3,3,Cin,Cout) of a stride-2 conv -> bf16 packed kernel (2,3,3,4Cin,Cout) of its stride-1 form."""
    _, KH, KW, Cin, Cout = w_packed.shape
    assert (KH, KW) == (3, 3)
    out = torch.empty((2, 3, 3, 4 * Cin, Cout), dtype=BF16, device=w_packed.device)
    _call("dasr_weight_expand_s2_bf16", _p(w_packed), _pa(out), Cin, Cout)
    return out


def weight_collapse_s2(dw_expanded, Cin, Cout):
    out = torch.empty((3, 3, Cin, Cout), dtype=torch.float32, device=dw_expanded.device)
    _call("dasr_weight_collapse_s2", _p(dw_expanded), _p(out), Cin, Cout)
    return out


def weight_expand_t2(w_packed, bias):
    """fp32 packed kernel (2,3,3,Cin,Cout) of a stride-2 ConvTranspose2d (+ bias) -> bf16 packed kernel (2,3,3,Cin,4Cout) and
    bias [4Cout] of the stride-1 convolution whose PixelShuffle(2) is the transposed convolution."""
    _, KH, KW, Cin, Cout = w_packed.shape
    assert (KH, KW) == (3, 3)
    out = torch.empty((2, 3, 3, Cin, 4 * Cout), dtype=BF16, device=w_packed.device)
    b4 = torch.empty((4 * Cout,), dtype=torch.float32, device=w_packed.device) if bias is not None else None
    _call("dasr_weight_expand_t2_bf16", _p(w_packed), _p(bias, True), _pa(out), _p(b4, True), Cin, Cout)
    return out, b4


def weight_collapse_t2(dw_expanded, db_expanded, Cin, Cout):
    dw = torch.empty((3, 3, Cin, Cout), dtype=torch.float32, device=dw_expanded.device)
    db = torch.empty((Cout,), dtype=torch.float32, device=dw_expanded.device) if db_expanded is not None else None
    _call("dasr_weight_collapse_t2", _p(dw_expanded), _p(db_expanded, True), _p(dw), _p(db, True), Cin, Cout)
    return dw, db


def zeros(shape, like):
    return torch.zeros(shape, dtype=torch.float32, device=like.device)


# ---- weights ------------------------------------------------------------------------------
def weight_pack(v, g, transposed=False, out=None, o_off=0, dtype=torch.float32):
    """Pack a PyTorch-layout fp32 kernel (optionally weight-normed) into HWIO, fp32 or bf16.  Returns (w, inv_norm)."""
    if transposed:
        I, O, KH, KW = v.shape
    else:
        O, I, KH, KW = v.shape
    if out is None:
        out = empty((2, KH, KW, I, O), v, dtype)   # [0]: HWIO, [1]: memory holds the per-tap transpose [KH,KW,O,I]
    ldo = out.shape[4]
    inv = empty((I if transposed else O,), v) if g is not None else None
    if out.dtype == BF16:
        _call("dasr_weight_pack_fwd_bf16", _p(v), _p(g, True), _pa(out), _p(inv, True), O, I, KH, KW, int(transposed), ldo,
              int(o_off))
    else:
        _call("dasr_weight_pack_fwd", _p(v), _p(g, True), _p(out), _p(inv, True), O, I, KH, KW, int(transposed), ldo,
              int(o_off))
    return out, inv


class PackJob(ctypes.Structure):
    """include/dasr.h: dasr_pack_job."""
    _fields_ = [("v", ctypes.c_void_p), ("g", ctypes.c_void_p), ("w", ctypes.c_void_p), ("inv_norm", ctypes.c_void_p),
                ("amax", ctypes.c_void_p)] + [(n, ctypes.c_int) for n in
                                              ("O", "I", "KK", "transposed", "ldo", "o_off", "bf16", "plain", "wg_begin", "reserved")]


class SplitJob(ctypes.Structure):
    """include/dasr.h: dasr_split_job."""
    _fields_ = [("w_packed", ctypes.c_void_p), ("wmax", ctypes.c_void_p), ("w_split", ctypes.c_void_p)] + \
               [(n, ctypes.c_int) for n in ("Cin", "Cout", "wg_begin", "reserved")]


class JobTable:
    """A job table of one multi-tensor launch: the ctypes array (host copy) and its image in device memory, kept together
    with the tensors the jobs point at."""

    def __init__(self, jobs, device, keep):
        self.n = len(jobs)
        self.host = (type(jobs[0]) * self.n)(*jobs)
        self.dev = torch.frombuffer(bytearray(bytes(self.host)), dtype=torch.uint8).to(device)
        self.keep = keep


def pack_job(v, g, w, inv, amax, transposed, o_off, plain, wg_begin):
    """One dasr_weight_pack_fwd call as a table entry (``w``: the packed buffer (2, KH, KW, I, ldo), or a flat buffer of
    ldo floats when ``plain``: a bias vector placed at o_off)."""
    if plain:
        O, I, KK, ldo = v.numel(), 1, 1, w.numel()
    else:
        if transposed:
            I, O, KH, KW = v.shape
        else:
            O, I, KH, KW = v.shape
        KK, ldo = KH * KW, w.shape[4]
    bf = w.dtype == BF16
    return PackJob(_p(v), _p(g, True), _lib.ptr(w, False, w.dtype), _p(inv, True), _p(amax, True), O, I, KK, int(transposed), ldo,
                   int(o_off), int(bf), int(plain), int(wg_begin), 0), (I if transposed else O)


def weight_pack_multi(table):
    _call("dasr_weight_pack_multi", ctypes.addressof(table.host), table.dev.data_ptr(), table.n)


def split_job(w, wmax, ws, wg_begin):
    KH, KW, Cin, Cout = _wdims(w)
    assert (KH, KW) == (3, 3) and w.dtype == torch.float32
    return (SplitJob(_p(w), _p(wmax), _lib.ptr(ws, False, torch.float16), Cin, Cout, int(wg_begin), 0),
            int(_lib.get().dasr_conv3x3_split2_weights_slabs(Cin, Cout)))


def conv3x3_split2_weights_multi(table):
    _call("dasr_conv3x3_split2_weights_multi", ctypes.addressof(table.host), table.dev.data_ptr(), table.n)


def weight_pack_bwd(dw, v, g, inv, transposed=False, o_off=0):
    if transposed:
        I, O, KH, KW = v.shape
    else:
        O, I, KH, KW = v.shape
    dv = torch.empty_like(v)
    dg = torch.empty_like(g) if g is not None else None
    assert dw.dim() == 4, "weight gradients are plain HWIO"
    _call("dasr_weight_pack_bwd", _p(dw), _p(v), _p(g, True), _p(inv, True), _p(dv), _p(dg, True), O, I, KH, KW,
          int(transposed), dw.shape[3], int(o_off))
    return dv, dg


# ---- convolution ----------------------------------------------------------------------------
def conv_out_hw(H, W, KH, KW, stride, pad, transposed):
    if transposed:
        return (H - 1) * stride - 2 * pad + KH, (W - 1) * stride - 2 * pad + KW
    return (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1


def pack_hwio(w4):
    """Packed kernel (HWIO + per-tap transpose) from a plain HWIO tensor - for tests and micro-benchmarks; the
    product path gets packed kernels from weight_pack (dasr_weight_pack_fwd)."""
    KH, KW, I, O = w4.shape
    out = torch.empty((2, KH, KW, I, O), dtype=w4.dtype, device=w4.device)      # bf16 in, bf16 packed
    out[0].copy_(w4)
    out[1].view(KH, KW, O, I).copy_(w4.permute(0, 1, 3, 2))
    return out


def _wdims(w):
    if w.dim() != 5 or w.shape[0] != 2:
        raise ValueError("dasr_amd: convolution kernels must be packed (ops.weight_pack / ops.pack_hwio)")
    return tuple(w.shape[1:])


def conv2d_fwd(x, w, bias=None, residual=None, stride=1, pad=1, transposed=False, act=ACT_NONE, ps_r=1, out_dtype=None,
               amax=None):
    """``out_dtype=torch.bfloat16`` with an fp32 ``x`` is the mask layer of the bf16 path (fp32 depth map in, bf16
    activation out); a bf16 ``x`` selects the bf16 path by itself (the 9x9 output conv then returns fp32).
    ``amax`` (fp32 path): a zeroed device float that receives max |y| and is attached to y (set_amax)."""
    B, H, W, Cin = x.shape
    KH, KW, wi, Cout = _wdims(w)
    assert wi == Cin, (w.shape, x.shape)
    Ho, Wo = conv_out_hw(H, W, KH, KW, stride, pad, transposed)
    bf = x.dtype == BF16 or out_dtype == BF16
    ydt = torch.float32
    if bf and not (KH == 9 and Cout <= 3):
        ydt = BF16
    if ps_r > 1:
        y = empty((B, Ho * ps_r, Wo * ps_r, Cout // (ps_r * ps_r)), x, ydt)
    else:
        y = empty((B, Ho, Wo, Cout), x, ydt)
    if bf:
        _call("dasr_conv2d_fwd_bf16", _pa(x), _pa(w), _p(bias, True), _pa(residual, True), _pa(y), B, H, W, Cin, Ho, Wo,
              Cout, KH, KW, stride, pad, int(transposed), act, ps_r)
    else:
        _call("dasr_conv2d_fwd", _p(x), _p(w), _p(bias, True), _p(residual, True), _p(y), _p(amax, True), B, H, W, Cin, Ho, Wo,
              Cout, KH, KW, stride, pad, int(transposed), act, ps_r)
        set_amax(y, amax)
    return y


def conv2d_epilogue_bwd(dy, y, Ho, Wo, Cout, act, ps_r, amax=None):
    B = y.shape[0]
    dconv = empty((B, Ho, Wo, Cout), y, y.dtype)
    if _is_bf(dy, y):
        _call("dasr_conv2d_epilogue_bwd_bf16", _pa(dy), _pa(y), _pa(dconv), B, Ho, Wo, Cout, act, ps_r)
    else:
        _call("dasr_conv2d_epilogue_bwd", _p(dy), _p(y), _p(dconv), _p(amax, True), B, Ho, Wo, Cout, act, ps_r)
        set_amax(dconv, amax)
    return dconv


def conv2d_dgrad(dconv, w, x_shape, stride=1, pad=1, transposed=False, out=None, out_dtype=torch.float32):
    """``out_dtype`` = dtype of the conv's input (bf16 input -> bf16 gradient; the 9x9 output conv takes an fp32 dconv)."""
    B, H, W, Cin = x_shape
    KH, KW, _, Cout = _wdims(w)
    _, Ho, Wo, _ = dconv.shape
    acc = out is not None
    if out is None:
        out = torch.empty(x_shape, dtype=out_dtype, device=dconv.device)
    drop_amax(out)
    if out.dtype == BF16:
        _call("dasr_conv2d_dgrad_bf16", _pa(dconv), _pa(w), _pa(out), int(acc), B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride,
              pad, int(transposed))
    else:
        _call("dasr_conv2d_dgrad", _p(dconv), _p(w), _p(out), int(acc), B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad,
              int(transposed))
    return out


def conv2d_dgrad_act_supported(x_shape, w, dconv_shape, stride, pad, transposed, ps_r):
    B, H, W, Cin = x_shape
    KH, KW, _, Cout = _wdims(w)
    _, Ho, Wo, _ = dconv_shape
    return bool(_lib.get().dasr_conv2d_dgrad_act_supported(B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad,
                                                          int(transposed), ps_r))


def conv2d_dgrad_act(dconv, w, x_act, act, ps_r, stride=1, pad=1, transposed=False):
    """Gradient w.r.t. the PRODUCER conv's raw output: dgrad * act'(x_act), stored un-shuffled."""
    B, H, W, Cin = x_act.shape
    KH, KW, _, Cout = _wdims(w)
    _, Ho, Wo, _ = dconv.shape
    out = torch.empty((B, H // ps_r, W // ps_r, Cin * ps_r * ps_r), dtype=torch.float32, device=dconv.device)
    _call("dasr_conv2d_dgrad_act", _p(dconv), _p(w), _p(x_act), _p(out), B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad,
          int(transposed), act, ps_r)
    return out


def conv2d_wgrad(x, dconv, w_shape, stride=1, pad=1, transposed=False, want_bias=True):
    B, H, W, Cin = x.shape
    KH, KW, _, Cout = w_shape
    _, Ho, Wo, _ = dconv.shape
    lib = _lib.get()
    bf = x.dtype == BF16
    wsfn = lib.dasr_conv2d_wgrad_workspace_bf16 if bf else lib.dasr_conv2d_wgrad_workspace
    nbytes = wsfn(B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, int(transposed))
    ws = torch.empty((max(1, (nbytes + 3) // 4),), dtype=torch.float32, device=x.device)
    dw = empty(w_shape, x)
    db = empty((Cout,), x) if want_bias else None
    if bf:
        _call("dasr_conv2d_wgrad_bf16", _pa(x), _pa(dconv), _p(dw), _p(db, True), _p(ws), nbytes, B, H, W, Cin, Ho, Wo,
              Cout, KH, KW, stride, pad, int(transposed))
    else:
        _call("dasr_conv2d_wgrad", _p(x), _p(dconv), _p(dw), _p(db, True), _p(ws), nbytes, B, H, W, Cin, Ho, Wo, Cout, KH,
              KW, stride, pad, int(transposed))
    return dw, db


def conv2d_wgrad_act(x, dy, y, w_shape, act, stride=1, pad=1, transposed=False, want_bias=True):
    """Weight/bias gradient of a conv whose input needs no gradient, activation backward fused."""
    B, H, W, Cin = x.shape
    KH, KW, _, Cout = w_shape
    _, Ho, Wo, _ = y.shape
    lib = _lib.get()
    if _is_bf(dy, y):                      # the mask layer of the bf16 path: fused kernel only
        dw = empty(w_shape, x)
        db = empty((Cout,), x) if want_bias else None
        _call("dasr_conv2d_wgrad_act_bf16", _p(x), _pa(dy), _pa(y), _p(dw), _p(db, True), B, H, W, Cin, Ho, Wo, Cout, KH, KW,
              stride, pad, int(transposed), act)
        return dw, db
    nbytes = lib.dasr_conv2d_wgrad_workspace(B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, int(transposed))
    ws = torch.empty((max(1, (nbytes + 3) // 4),), dtype=torch.float32, device=x.device)
    fused = bool(lib.dasr_conv2d_wgrad_act_fused(B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, int(transposed)))
    scratch = None if fused else torch.empty_like(y)
    dw = empty(w_shape, x)
    db = empty((Cout,), x) if want_bias else None
    _call("dasr_conv2d_wgrad_act", _p(x), _p(dy), _p(y), _p(dw), _p(db, True), _p(scratch, True), _p(ws), nbytes, B, H,
          W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, int(transposed), act)
    return dw, db


# ---- fp32 convolutions on the bf16 matrix cores (split bf16 x 3: three pieces per operand, six products) -------
def conv3x3_split_supported(H, W, Cin, Cout):
    return bool(_lib.get().dasr_conv3x3_split_supported(H, W, Cin, Cout))


def conv3x3_split_weights(w):
    """fp32 packed kernel (2,3,3,Cin,Cout) -> the bf16 three-piece image both split kernels read."""
    KH, KW, Cin, Cout = _wdims(w)
    assert (KH, KW) == (3, 3) and w.dtype == torch.float32
    n = int(_lib.get().dasr_conv3x3_split_weights_bytes(Cin, Cout)) // 2
    ws = torch.empty((n,), dtype=BF16, device=w.device)
    _call("dasr_conv3x3_split_weights", _p(w), _pa(ws), Cin, Cout)
    return ws


def conv3x3_fwd_split(x, ws, bias, Cout, residual=None, act=ACT_NONE, ps_r=1):
    B, H, W, Cin = x.shape
    if ps_r > 1:
        y = empty((B, H * ps_r, W * ps_r, Cout // (ps_r * ps_r)), x)
    else:
        y = empty((B, H, W, Cout), x)
    _call("dasr_conv3x3_fwd_split", _p(x), _pa(ws), _p(bias, True), _p(residual, True), _p(y), B, H, W, Cin, Cout, act, ps_r)
    return y


def conv3x3_wgrad_split(x, dconv, want_bias=True):
    """(dw [3,3,Cin,Cout], dbias | None) of a plain 3x3 convolution, fp32, on the split-bf16 kernel."""
    B, H, W, Cin = x.shape
    Cout = dconv.shape[3]
    nbytes = int(_lib.get().dasr_conv3x3_wgrad_split_workspace(B, H, W, Cin, Cout))
    ws = torch.empty((max(1, (nbytes + 3) // 4),), dtype=torch.float32, device=x.device)
    dw = empty((3, 3, Cin, Cout), x)
    db = empty((Cout,), x) if want_bias else None
    _call("dasr_conv3x3_wgrad_split", _p(x), _p(dconv), _p(dw), _p(db, True), _p(ws), nbytes, B, H, W, Cin, Cout)
    return dw, db


def conv3x3_dgrad_split(dconv, ws, x_shape, out=None):
    B, H, W, Cin = x_shape
    Cout = dconv.shape[3]
    acc = out is not None
    if out is None:
        out = empty(tuple(x_shape), dconv)
    drop_amax(out)
    _call("dasr_conv3x3_dgrad_split", _p(dconv), _pa(ws), _p(out), 1 if acc else 0, B, H, W, Cin, Cout)
    return out


# ---- the same with two fp16 pieces / three products; every tensor operand comes with its max |.| in device memory ----
def absmax(x):
    """max |x| as an amax buffer on the device (the fp16 scheme's kernels derive their power-of-two scale from it)."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = amax_buffer(x)
    _call("dasr_absmax", _p(x), x.numel(), _p(out))
    return out


def conv3x3_split2_weights(w):
    """fp32 packed kernel (2,3,3,Cin,Cout) -> (fp16 two-piece image, max |w| device scalar)."""
    KH, KW, Cin, Cout = _wdims(w)
    assert (KH, KW) == (3, 3) and w.dtype == torch.float32
    wmax = absmax(w[0])
    n = int(_lib.get().dasr_conv3x3_split2_weights_bytes(Cin, Cout)) // 2
    ws = torch.empty((n,), dtype=torch.float16, device=w.device)
    _call("dasr_conv3x3_split2_weights", _p(w), _p(wmax), _lib.ptr(ws, False, torch.float16), Cin, Cout)
    return ws, wmax


def conv3x3_fwd_split2(x, xmax, ws, bias, Cout, residual=None, act=ACT_NONE, ps_r=1, amax=None):
    B, H, W, Cin = x.shape
    if ps_r > 1:
        y = empty((B, H * ps_r, W * ps_r, Cout // (ps_r * ps_r)), x)
    else:
        y = empty((B, H, W, Cout), x)
    _call("dasr_conv3x3_fwd_split2", _p(x), _p(xmax), _lib.ptr(ws[0], False, torch.float16), _p(ws[1]), _p(bias, True),
          _p(residual, True), _p(y), _p(amax, True), B, H, W, Cin, Cout, act, ps_r)
    return set_amax(y, amax)


def conv3x3_wgrad_split2(x, xmax, dconv, dmax, want_bias=True):
    B, H, W, Cin = x.shape
    Cout = dconv.shape[3]
    nbytes = int(_lib.get().dasr_conv3x3_wgrad_split_workspace(B, H, W, Cin, Cout))
    ws = torch.empty((max(1, (nbytes + 3) // 4),), dtype=torch.float32, device=x.device)
    dw = empty((3, 3, Cin, Cout), x)
    db = empty((Cout,), x) if want_bias else None
    _call("dasr_conv3x3_wgrad_split2", _p(x), _p(xmax), _p(dconv), _p(dmax), _p(dw), _p(db, True), _p(ws), nbytes, B, H, W, Cin,
          Cout)
    return dw, db


def conv3x3_dgrad_split2(dconv, dmax, ws, x_shape, out=None):
    B, H, W, Cin = x_shape
    Cout = dconv.shape[3]
    acc = out is not None
    if out is None:
        out = empty(tuple(x_shape), dconv)
    drop_amax(out)
    _call("dasr_conv3x3_dgrad_split2", _p(dconv), _p(dmax), _lib.ptr(ws[0], False, torch.float16), _p(ws[1]), _p(out), 1 if acc else 0, B, H, W, Cin, Cout)
    return out


# ---- the 9x9 output convolution in the same scheme (csrc/conv9_split.hip); w = the packed kernel (2,9,9,Cin,Cout) ------
def conv9_split_supported(H, W, Cin, Cout):
    return bool(_lib.get().dasr_conv9_split_supported(H, W, Cin, Cout))


def conv9_fwd_split2(x, xmax, w, wmax, bias):
    B, H, W, Cin = x.shape
    Cout = w.shape[4]
    y = empty((B, H, W, Cout), x)
    _call("dasr_conv9_fwd_split2", _p(x), _p(xmax), _p(w), _p(wmax), _p(bias, True), _p(y), B, H, W, Cin, Cout)
    return y


def conv9_dgrad_split2(dconv, dmax, w, wmax, x_shape, out=None):
    B, H, W, Cin = x_shape
    Cout = dconv.shape[3]
    acc = out is not None
    if out is None:
        out = empty(tuple(x_shape), dconv)
    drop_amax(out)
    _call("dasr_conv9_dgrad_split2", _p(dconv), _p(dmax), _p(w), _p(wmax), _p(out), 1 if acc else 0, B, H, W, Cin, Cout)
    return out


def conv9_wgrad_split2(x, xmax, dconv, dmax, want_bias=True):
    B, H, W, Cin = x.shape
    Cout = dconv.shape[3]
    nbytes = int(_lib.get().dasr_conv9_wgrad_split2_workspace(B, H, W, Cin, Cout))
    ws = torch.empty((max(1, (nbytes + 3) // 4),), dtype=torch.float32, device=x.device)
    dw = empty((9, 9, Cin, Cout), x)
    db = empty((Cout,), x) if want_bias else None
    _call("dasr_conv9_wgrad_split2", _p(x), _p(xmax), _p(dconv), _p(dmax), _p(dw), _p(db, True), _p(ws), nbytes, B, H, W, Cin, Cout)
    return dw, db


def conv2d_fwd_stats(x, w, bias):
    """3x3 / stride 1 / pad 1 conv (+bias) and the InstanceNorm statistics of its output: (y, mean[B,C], var[B,C])."""
    B, H, W, Cin = x.shape
    KH, KW, Cin2, Cout = w.shape[1:]
    assert (KH, KW) == (3, 3) and Cin2 == Cin
    y = empty((B, H, W, Cout), x)
    mean = empty((B, Cout), x)
    var = empty((B, Cout), x)
    nbytes = int(_lib.get().dasr_conv2d_fwd_stats_workspace(B, H, W, Cin, Cout))
    ws = torch.empty((max(1, (nbytes + 3) // 4),), dtype=torch.float32, device=x.device)
    _call("dasr_conv2d_fwd_stats", _p(x), _p(w), _p(bias, True), _p(y), _p(mean), _p(var), _p(ws), nbytes, B, H, W, Cin,
          Cout)
    return y, mean, var


# ---- instance norm / SEAN -------------------------------------------------------------------
def instnorm_stats(x):
    B, H, W, C = x.shape
    mean = empty((B, C), x)
    var = empty((B, C), x)
    nbytes = _lib.get().dasr_instnorm_stats_workspace(B, H * W, C)
    ws = torch.empty((max(1, nbytes // 4),), dtype=torch.float32, device=x.device)
    if x.dtype == BF16:
        _call("dasr_instnorm_stats_bf16", _pa(x), _p(mean), _p(var), _p(ws), nbytes, B, H * W, C)
    else:
        _call("dasr_instnorm_stats", _p(x), _p(mean), _p(var), _p(ws), nbytes, B, H * W, C)
    return mean, var


def dynk_fwd(st, A_w, A_b, Wg, Wb):
    B, K, L = st.shape
    C = Wg.shape[0]
    stp = torch.empty_like(st)
    D = empty((B, 2, 9, K, C), st)
    _call("dasr_dynk_fwd", _p(st), _p(A_w), _p(A_b), _p(Wg), _p(Wb), _p(stp), _p(D), B, K, L, C)
    return stp, D


def dynk_bwd(dD, st, stp, A_w, Wg, Wb, dst_accum):
    """Returns (dWg, dWb, dA_w, dA_b); adds the depth-matrix gradient into ``dst_accum`` in place."""
    B, K, L = st.shape
    C = Wg.shape[0]
    dWg, dWb = torch.empty_like(Wg), torch.empty_like(Wb)
    dA_w = torch.empty_like(A_w)
    dA_b = empty((K,), st)
    scratch = torch.empty_like(st)
    _call("dasr_dynk_bwd", _p(dD), _p(st), _p(stp), _p(A_w), _p(Wg), _p(Wb), _p(dWg), _p(dWb), _p(dA_w), _p(dA_b),
          _p(dst_accum), _p(scratch), B, K, L, C)
    return dWg, dWb, dA_w, dA_b


def soft_mask_max_regions():
    return int(_lib.get().dasr_sean_soft_mask_max_regions())


def mask_compress(mask):
    """Region index per pixel (uint8 [B,H,W]) + device flag (int32 [1], non-zero = masks are not one-hot)."""
    B, K, H, W = mask.shape
    region = torch.empty((B, H, W), dtype=torch.uint8, device=mask.device)
    flag = torch.empty((1,), dtype=torch.int32, device=mask.device)
    _call("dasr_mask_compress", _p(mask), _lib.ptr(region, dtype=torch.uint8), _lib.ptr(flag, dtype=torch.int32), B, K,
          H, W)
    return region, flag


def depth_to_masks(depth, num_masks=10, fixed_edges=None, want_planes=True):
    """Device getDepthMask (dasr_depth_to_masks): depth [B,1,h,w] or [B,h,w] -> (planes [B,K,h,w] float or None,
    region bytes [B,h,w]).  fixed_edges: None (per-sample min/max range) or K+1 float32 device values."""
    h, w = depth.shape[-2:]
    B = depth.numel() // (h * w)
    d = depth.contiguous()
    planes = torch.empty((B, num_masks, h, w), dtype=torch.float32, device=d.device) if want_planes else None
    region = torch.empty((B, h, w), dtype=torch.uint8, device=d.device)
    nbytes = int(_lib.get().dasr_depth_to_masks_workspace(B, h * w))
    ws = torch.empty((max(nbytes, 4) // 4,), dtype=torch.float32, device=d.device)
    _call("dasr_depth_to_masks", _p(d), _p(fixed_edges, True) if fixed_edges is not None else None,
          _p(planes) if planes is not None else None, _lib.ptr(region, dtype=torch.uint8), _p(ws), nbytes, B, h * w,
          num_masks)
    return planes, region


def _rf(region, flag):
    if region is None:
        return None, None
    return _lib.ptr(region, dtype=torch.uint8), (_lib.ptr(flag, dtype=torch.int32) if flag is not None else None)


def sean_fwd(t, mean, var, gb2, mask, region, flag, D, bias_g, bias_b, alpha_g, alpha_b, residual, relu, amax=None):
    B, H, W, C = t.shape
    K = mask.shape[1]
    assert mask.shape == (B, K, H, W) and gb2.shape == (B, H, W, 2 * C) and D.shape == (B, 2, 9, K, C)
    out = torch.empty_like(t)
    rp, fp = _rf(region, flag)
    if _is_bf(t, gb2, residual):
        _call("dasr_sean_fwd_bf16", _pa(t), _p(mean), _p(var), _pa(gb2), _p(mask), rp, fp, _p(D), _p(bias_g), _p(bias_b),
              _p(alpha_g), _p(alpha_b), _pa(residual, True), _pa(out), int(relu), B, H, W, C, K, IN_EPS)
    else:
        _call("dasr_sean_fwd", _p(t), _p(mean), _p(var), _p(gb2), _p(mask), rp, fp, _p(D), _p(bias_g), _p(bias_b),
              _p(alpha_g), _p(alpha_b), _p(residual, True), _p(out), _p(amax, True), int(relu), B, H, W, C, K, IN_EPS)
        set_amax(out, amax)
    return out


def sean_bwd(dout, out, t, mean, var, gb2, mask, region, flag, D, bias_g, bias_b, alpha_g, alpha_b, relu, want_dres,
             dt_amax=None, dgb2_amax=None):
    B, H, W, C = t.shape
    K = mask.shape[1]
    lib = _lib.get()
    nbytes = lib.dasr_sean_bwd_workspace(B, H, W, C, K)
    ws = torch.empty((max(1, (nbytes + 3) // 4),), dtype=torch.float32, device=t.device)
    dt = torch.empty_like(t)
    dgb2 = torch.empty_like(gb2)
    dD = torch.empty_like(D)
    dbg, dbb = empty((C,), t), empty((C,), t)        # fp32 whatever the activation dtype
    dag, dab = empty((1,), t), empty((1,), t)
    dres = torch.empty_like(t) if want_dres else None
    rp, fp = _rf(region, flag)
    if _is_bf(dout, out, t, gb2):
        _call("dasr_sean_bwd_bf16", _pa(dout), _pa(out), _pa(t), _p(mean), _p(var), _pa(gb2), _p(mask), rp, fp, _p(D), _p(bias_g),
              _p(bias_b), _p(alpha_g), _p(alpha_b), _pa(dt), _pa(dgb2), _p(dD), _p(dbg), _p(dbb), _p(dag), _p(dab),
              _pa(dres, True), _p(ws), nbytes, int(relu), B, H, W, C, K, IN_EPS)
    else:
        _call("dasr_sean_bwd", _p(dout), _p(out), _p(t), _p(mean), _p(var), _p(gb2), _p(mask), rp, fp, _p(D), _p(bias_g),
              _p(bias_b), _p(alpha_g), _p(alpha_b), _p(dt), _p(dgb2), _p(dD), _p(dbg), _p(dbb), _p(dag), _p(dab),
              _p(dres, True), _p(dt_amax, True), _p(dgb2_amax, True), _p(ws), nbytes, int(relu), B, H, W, C, K, IN_EPS)
        set_amax(dt, dt_amax)
        set_amax(dgb2, dgb2_amax)
    return dt, dgb2, dD, dbg, dbb, dag, dab, dres


# ---- region pooling -------------------------------------------------------------------------
def region_pool_fwd(feat, mask):
    B, h, w, L = feat.shape
    _, K, H, W = mask.shape
    maskr = empty((B, K, h, w), feat)
    area = empty((B, K), feat)
    out = empty((B, K, L), feat)
    _call("dasr_region_pool_fwd", _p(feat), _p(mask), _p(maskr), _p(area), _p(out), B, K, L, h, w, H, W)
    return out, maskr, area


def region_pool_bwd(dout, maskr, area, feat_shape):
    B, h, w, L = feat_shape
    K = maskr.shape[1]
    dfeat = torch.empty(feat_shape, dtype=torch.float32, device=dout.device)
    _call("dasr_region_pool_bwd", _p(dout), _p(maskr), _p(area), _p(dfeat), B, K, L, h, w)
    return dfeat


# ---- harness losses (one-hot masks) ------------------------------------------------------------
def loss_sums(sr, hr, region, K):
    B, C, H, W = sr.shape
    _, h, w = region.shape
    scale = H // h
    assert H == h * scale and W == w * scale
    sums = empty((2 * K + 1,), sr)
    _call("dasr_loss_sums", _p(sr), _p(hr), _lib.ptr(region, dtype=torch.uint8), _p(sums), B, C, h, w, scale, K)
    return sums


def loss_bwd(sr, hr, region, dsums, K):
    B, C, H, W = sr.shape
    _, h, w = region.shape
    dsr = torch.empty_like(sr)
    _call("dasr_loss_bwd", _p(sr), _p(hr), _lib.ptr(region, dtype=torch.uint8), _p(dsums), _p(dsr), B, C, h, w, H // h, K)
    return dsr
