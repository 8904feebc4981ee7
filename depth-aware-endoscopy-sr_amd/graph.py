"""Forward plan of DepthNet as a sequence of HIP kernel calls recorded on a Tape.

Mirrors DepthNet.forward (reference codes/models/modules/sftmd_arch.py:912-950) and the blocks
it calls: Encoder.forward (:771-783), Depth_Residual_Block_Mask.forward (:826-834),
Classic_Residual_Block.forward (:147-151), SEAN.forward (normalization.py:52-92).  What differs
from the reference is HOW, not WHAT: NHWC activations, the 256-channel style map collapsed into
per-sample dynamic 3x3 kernels over the K-channel mask, the two instance norms collapsed into one
scale, gamma_o/beta_o run as one 2C->2C convolution, activation / PixelShuffle / residual fused
into convolution epilogues.
"""
import math

import torch

from . import ops
from .tape import Tape, Var, accum


# dasr_conv2d_dgrad_act (the producer's activation / PixelShuffle backward applied in the consumer's dgrad epilogue)
# is OFF by default.  Measured on MI355X at the x8 bench shapes it removes the epilogue-backward passes (26 -> 6 ms
# per 4 steps) but the masked epilogue's extra 4-byte loads and the scattered un-shuffle stores are not hidden
# under matrix work: the 9x9 dgrad went 3.6 -> 6.2 ms and the 128->128 dgrad 0.82 -> 1.14 ms, a net loss of 4-10 %
# of the step.  Kept as a tested entry point for shapes where the producer tensor is small.
FUSE_DGRAD_ACT = False

# Run the depth-map branch of the SEANs on a side HIP stream (forward and backward).
SIDE_STREAM = True
# Weight / bias gradients are off the critical path of the backward (nothing but the optimiser waits for them): they
# run on a third stream and fill whatever the data-gradient chain leaves idle.
import os as _os
import threading as _threading
TAIL_WGRAD_SIDE = False  # weight gradients of the HR tail on the (then idle) depth-branch stream; see conv(side_wgrad=)
FUSE_INSTNORM_STATS = False  # measured: 115.5 -> 111.5 frames/s when on (two more barriers + reductions in the 64->64 conv epilogue cost more than the statistics pass they replace); the entry point stays, tested
ENCODER_S2D = True     # bf16 path: encoder layers 2-5 on the bf16 stride-1 kernels (space-to-depth form); False: fp32 gather kernels
# fp32 path: every 3x3 / stride 1 trunk convolution with channel counts that are multiples of 32 (gamma_o|beta_o 128 -> 128, the
# DGB 64 -> 64 ones, the HR tail) - forward, dgrad and weight gradient - at fp32 accuracy on the 16-bit matrix cores
# (csrc/conv_split_bf16.hip); False: the exact-fp32 MFMA kernels everywhere.
SPLIT_BF16 = True
SPLIT_WGRAD = True              # ... and their weight gradients (K = pixels: operands split at read time)
# 3: three bf16 pieces, six products.  2: two fp16 pieces, three products, every tensor operand scaled by a power of two taken
# from its max |.| (ops.absmax, device side): half the matrix work and measurably closer to float64 (csrc/conv_split_bf16.hip).
SPLIT_PIECES = 2
# ... and the kernels that PRODUCE those tensors (SEAN forward / backward, the mask layer, the split kernels' own epilogue, the
# activation / PixelShuffle backward) leave max |.| behind themselves; False: one ops.absmax pass per operand (A/B, tests)
FUSE_AMAX = True
SPLIT_CONV9 = True              # ... and the 9x9 output convolution (csrc/conv9_split.hip)
SPLIT_MIN_PIXELS = 1 << 14      # below this the launch is latency-bound either way
WGRAD_STREAM = False   # measured: 167.7 -> 182.3 ms/step when on (contention between co-running MFMA kernels)
_SIDE = {}
_SIDE_LOCK = _threading.Lock()


def _side_stream(device, which="branch"):
    key = (str(device), which)
    with _SIDE_LOCK:                      # nn.DataParallel calls forward from one thread per replica
        if key not in _SIDE:
            # high priority: the depth branch is the longer of the two chains in both directions (measured +0.8 %)
            _SIDE[key] = torch.cuda.Stream(device=device, priority=-1)
        return _SIDE[key]


# ---------------------------------------------------------------------------------------------
# recorded primitives
# ---------------------------------------------------------------------------------------------
def _folded(tape, key, tensors, make):
    """Inference-time weight folding: with the tape off and a cache attached
    (DepthNet keeps one per module), a packed kernel - weight_norm g*v/||v|| folded in, HWIO + per-tap transpose - is
    built once and reused until one of its source parameters changes (torch's version counters)."""
    cache = tape.fold_cache
    if tape.enabled or cache is None or key is None:
        return make()
    ver = tuple((t.data_ptr(), ops.tensor_version(t)) for t in tensors)
    key = key + (str(tensors[0].device),)      # DataParallel replicas share the cache dict, one entry per device
    hit = cache.get(key)
    if hit is not None and hit[0] == ver:
        return hit[1]
    val = make()
    cache[key] = (ver, val)
    return val


# Training: the ~120 packed kernels of a step (weight_norm + HWIO pack), the max |w| and fp16 x 2 image of the ~55 split
# convolutions and the 26 bias pairs in TWO launches at the start of the forward (dasr_weight_pack_multi,
# dasr_conv3x3_split2_weights_multi) instead of ~300 small ones spread over it; False: one launch per tensor (A/B, tests).
PREPACK = True


class Prepack:
    """The packed kernels of one network on one device.  The first training forward runs the per-tensor path and is recorded
    (which parameters are packed how, which packed kernels get a split image); its output buffers are kept and become the
    targets of the job tables.  Every later forward with the same signature (flags, input shape, parameter storage) refills
    them in two launches before anything else is issued and hands them out by key."""

    def __init__(self):
        self.sig = None
        self.ready = False
        self.entries = {}
        self.tables = None
        self.misses = 0             # tables built and then not used by the next forward: parameters that move every step
        self.off = False            # (nn.DataParallel replicas are fresh copies) or a new input shape every time make the
                                    # recording pointless - three in a row turn it off
        self.lock = _threading.Lock()   # one forward at a time (replicas on one device run in threads: the others go per tensor)

    def __deepcopy__(self, memo):       # a copied / pickled module starts with an empty recording (device buffers and the
        return Prepack()                # lock do not travel)

    def __reduce__(self):
        return (Prepack, ())

    @staticmethod
    def signature(inp, act_dtype):
        return (tuple(inp.shape), str(inp.device), str(act_dtype), SPLIT_BF16, SPLIT_WGRAD, SPLIT_PIECES, SPLIT_CONV9,
                SPLIT_MIN_PIXELS, ENCODER_S2D)

    def begin(self, P, sig):
        """Start of a training forward: launch the tables if they fit this step, else start recording afresh."""
        if self.ready and sig == self.sig and all(P[n].data.data_ptr() == ptr and P[n].data.shape == shp
                                                  for n, ptr, shp in self.sources):
            ops.weight_pack_multi(self.tables[0])
            if self.tables[1] is not None:
                ops.conv3x3_split2_weights_multi(self.tables[1])
            self.misses = 0
            return
        if self.ready:
            self.misses += 1
            self.off = self.off or self.misses >= 3
        self.sig, self.ready, self.entries, self.tables = sig, False, {}, None

    def get(self, key):
        return self.entries.get(key) if self.ready else None

    def note(self, key, jobs, w, inv=None):
        """Recording: ``jobs`` = [(v Var, g Var or None, transposed, o_off, plain)] that fill the buffer ``w``."""
        if not self.ready and not self.off and key is not None:
            self.entries[key] = {"jobs": jobs, "w": w, "inv": inv, "split": None}

    def note_split(self, key, split):
        if not self.ready and key in self.entries:
            self.entries[key]["split"] = split

    def finish(self):
        """End of the recorded forward: build the two job tables over the buffers that forward produced."""
        if self.ready or not self.entries:
            return
        pj, sj, keep, sources = [], [], [], []
        npk = nsl = 0
        for e in self.entries.values():
            w, am = e["w"], None
            if e["split"] is not None:                 # (ws fp16 image, wmax) or ("c9", wmax): wmax now comes from the pack jobs
                am = ops.amax_buffer(w)
                if isinstance(e["split"][0], str):
                    e["split"] = (e["split"][0], am)
                else:
                    ws = e["split"][0]
                    job, n = ops.split_job(w, am, ws, nsl)
                    sj.append(job)
                    nsl += n
                    e["split"] = (ws, am)
            for v, g, transposed, o_off, plain in e["jobs"]:
                job, n = ops.pack_job(v.data, g.data if g is not None else None, w, e["inv"], am, transposed, o_off, plain, npk)
                pj.append(job)
                npk += n
                for t in (v, g):
                    if t is not None:
                        sources.append((t.name, t.data.data_ptr(), t.data.shape))
            keep.append((w, e["inv"], am, e["split"]))
            e["jobs"] = None
        dev = keep[0][0].device
        self.tables = (ops.JobTable(pj, dev, keep), ops.JobTable(sj, dev, keep) if sj else None)
        self.sources = sources
        self.ready = True


def pack(tape, v, g=None, transposed=False, dtype=torch.float32):
    """weight_norm (if g) + OIHW -> HWIO; ``dtype`` bf16 for the trunk kernels of the mixed-precision path (fp32 master
    weights: the packed copy is rounded, the gradient comes back in fp32)."""
    srcs = [v.data] + ([g.data] if g is not None else [])
    key = ("pack", v.name, str(dtype)) if v.name else None
    pp = tape.prepack if (key is not None and (g is None or g.name)) else None
    hit = pp.get(key) if pp is not None else None
    if hit is not None:
        w, inv = hit["w"], hit["inv"]
    else:
        w, inv = _folded(tape, key, srcs,
                         lambda: ops.weight_pack(v.data, g.data if g is not None else None, transposed, dtype=dtype))
        if pp is not None:
            pp.note(key, [(v, g, transposed, 0, False)], w, inv)
    out = Var(w, v.requires_grad or (g is not None and g.requires_grad))
    if pp is not None:
        out.pack = key
        out.split = hit["split"] if hit is not None else None

    def bwd():
        if out.grad is None:
            return
        if out.grad_event is not None:
            torch.cuda.current_stream().wait_event(out.grad_event)
            out.grad.record_stream(torch.cuda.current_stream())
        dv, dg = ops.weight_pack_bwd(out.grad, v.data, g.data if g is not None else None, inv, transposed)
        out.grad = None
        accum(v, dv)
        if g is not None:
            accum(g, dg)

    tape.record(bwd)
    return out


def pack_pair(tape, va, vb, dtype=torch.float32):
    """Two plain OIHW kernels with equal Cin side by side along Cout (mlp_gamma_o | mlp_beta_o)."""
    Oa, I, KH, KW = va.data.shape
    Ob = vb.data.shape[0]

    def make():
        w = ops.empty((2, KH, KW, I, Oa + Ob), va.data, dtype)
        ops.weight_pack(va.data, None, False, out=w, o_off=0)
        ops.weight_pack(vb.data, None, False, out=w, o_off=Oa)
        return w

    key = ("pair", va.name, str(dtype)) if va.name else None
    pp = tape.prepack if (key is not None and vb.name) else None
    hit = pp.get(key) if pp is not None else None
    if hit is not None:
        w = hit["w"]
    else:
        w = _folded(tape, key, [va.data, vb.data], make)
        if pp is not None:
            pp.note(key, [(va, None, False, 0, False), (vb, None, False, Oa, False)], w)
    out = Var(w, va.requires_grad or vb.requires_grad)
    if pp is not None:
        out.pack = key
        out.split = hit["split"] if hit is not None else None

    def bwd():
        if out.grad is None:
            return
        if out.grad_event is not None:
            torch.cuda.current_stream().wait_event(out.grad_event)
            out.grad.record_stream(torch.cuda.current_stream())
        da, _ = ops.weight_pack_bwd(out.grad, va.data, None, None, False, o_off=0)
        db, _ = ops.weight_pack_bwd(out.grad, vb.data, None, None, False, o_off=Oa)
        out.grad = None
        accum(va, da)
        accum(vb, db)

    tape.record(bwd)
    return out


def bias_pair(tape, ba, bb):
    na, nb = ba.data.numel(), bb.data.numel()

    def make():
        buf = ops.empty((na + nb,), ba.data)
        ops.copy_(buf[:na], ba.data)
        ops.copy_(buf[na:], bb.data)
        return buf

    key = ("bias", ba.name) if ba.name else None
    pp = tape.prepack if (key is not None and bb.name and ba.data.dtype == torch.float32) else None
    hit = pp.get(key) if pp is not None else None
    if hit is not None:
        buf = hit["w"]
    else:
        buf = _folded(tape, key, [ba.data, bb.data], make)
        if pp is not None:
            pp.note(key, [(ba, None, False, 0, True), (bb, None, False, na, True)], buf)
    out = Var(buf, ba.requires_grad or bb.requires_grad)

    def bwd():
        if out.grad is None:
            return
        if out.grad_event is not None:
            torch.cuda.current_stream().wait_event(out.grad_event)
            out.grad.record_stream(torch.cuda.current_stream())
        g = out.grad
        out.grad = None
        accum(ba, g[:na])
        accum(bb, g[na:])

    tape.record(bwd)
    return out


def _is_c9(w):
    return isinstance(w.split, tuple) and isinstance(w.split[0], str)


def _amax(var):
    """max |var.data| on the device for the fp16 x 2 split kernels: what the producing kernel left with the tensor
    (ops.set_amax: valid wherever the tensor is), else one ops.absmax pass, kept per tensor and stream (the forward
    convolution and its weight gradient share it)."""
    a = ops.get_amax(var.data)
    if a is not None:
        return a
    key = torch.cuda.current_stream().cuda_stream if var.data.is_cuda else 0
    if var.amax is None or var.amax[0] != key:
        var.amax = (key, ops.absmax(var.data))
    return var.amax[1]


def _amax_t(t):
    """The same for a bare tensor (a gradient on its way through the tape)."""
    a = ops.get_amax(t)
    return a if a is not None else ops.absmax(t)


def want_amax(tape, like, shape=None):
    """An amax buffer for a producer kernel to fill (ops.amax_buffer) when the tensor it is about to write - shaped ``shape``
    (default: like ``like``), on ``like``'s device - can be the operand of an fp16 x 2 split convolution, else None: NHWC
    fp32 on the fp32 path, at least SPLIT_MIN_PIXELS pixels."""
    if not (SPLIT_BF16 and SPLIT_PIECES == 2 and FUSE_AMAX) or like.dtype != torch.float32 or tape.act_dtype != torch.float32:
        return None
    shape = tuple(like.shape) if shape is None else shape
    if len(shape) != 4 or shape[0] * shape[1] * shape[2] < SPLIT_MIN_PIXELS:
        return None
    return ops.amax_buffer(like)


def conv(tape, x, w, bias=None, *, stride=1, pad=1, transposed=False, act=ops.ACT_NONE, ps_r=1, residual=None,
         side_wgrad=False, want_stats=False, out_dtype=None):
    stats = None
    f32 = x.data.dtype == torch.float32 and out_dtype in (None, torch.float32)
    if (want_stats and FUSE_INSTNORM_STATS and f32 and act == ops.ACT_NONE and ps_r == 1 and residual is None and
            stride == 1 and pad == 1 and not transposed and tuple(w.data.shape[1:3]) == (3, 3)):
        # the InstanceNorm statistics of the output come out of the convolution's epilogue (no second pass over y)
        y, mean, var = ops.conv2d_fwd_stats(x.data, w.data, bias.data if bias is not None else None)
        stats = (mean, var)
    elif (SPLIT_BF16 and f32 and w.data.dtype == torch.float32 and stride == 1 and pad == 1 and not transposed
          and tuple(w.data.shape[1:3]) == (3, 3)
          and (ps_r == 1 or (ps_r == 2 and residual is None and w.data.shape[4] % 128 == 0))
          and x.data.shape[0] * x.data.shape[1] * x.data.shape[2] >= SPLIT_MIN_PIXELS
          and ops.conv3x3_split_supported(x.data.shape[1], x.data.shape[2], w.data.shape[3], w.data.shape[4])):
        if w.split is None:           # (eval + no_grad: cached with the folded kernel it is derived from)
            w.split = _folded(tape, ("split%d" % SPLIT_PIECES, "@%x" % w.data.data_ptr()), [w.data],
                              (lambda: ops.conv3x3_split2_weights(w.data)) if SPLIT_PIECES == 2 else
                              (lambda: ops.conv3x3_split_weights(w.data)))
            if w.pack is not None and isinstance(w.split, tuple):
                tape.prepack.note_split(w.pack, w.split)
        if isinstance(w.split, tuple):
            y = ops.conv3x3_fwd_split2(x.data, _amax(x), w.split, bias.data if bias is not None else None, w.data.shape[4],
                                       residual.data if residual is not None else None, act, ps_r,
                                       amax=want_amax(tape, x.data))
        else:
            y = ops.conv3x3_fwd_split(x.data, w.split, bias.data if bias is not None else None, w.data.shape[4],
                                      residual.data if residual is not None else None, act, ps_r)
    elif (SPLIT_BF16 and SPLIT_PIECES == 2 and SPLIT_CONV9 and f32 and w.data.dtype == torch.float32 and stride == 1 and pad == 4
          and not transposed and tuple(w.data.shape[1:3]) == (9, 9) and act == ops.ACT_NONE and ps_r == 1 and residual is None
          and x.data.shape[0] * x.data.shape[1] * x.data.shape[2] >= SPLIT_MIN_PIXELS
          and ops.conv9_split_supported(x.data.shape[1], x.data.shape[2], w.data.shape[3], w.data.shape[4])):
        # the 9x9 output convolution, same scheme (csrc/conv9_split.hip): the kernel is split when it is staged, only its max |.| is needed
        if w.split is None:
            w.split = ("c9", _folded(tape, ("c9max", "@%x" % w.data.data_ptr()), [w.data], lambda: ops.absmax(w.data[0])))
            if w.pack is not None:
                tape.prepack.note_split(w.pack, w.split)
        y = ops.conv9_fwd_split2(x.data, _amax(x), w.data, w.split[1], bias.data if bias is not None else None)
    else:
        # (the mask layer 1 -> 2C feeds the gamma_o|beta_o convolution, the encoder's first layer the head: their kernel
        # leaves max |y| behind)
        am = want_amax(tape, x.data) if (f32 and x.data.shape[3] in (1, 3) and ps_r == 1) else None
        y = ops.conv2d_fwd(x.data, w.data, bias.data if bias is not None else None,
                           residual.data if residual is not None else None, stride, pad, transposed, act, ps_r, out_dtype,
                           amax=am)
    needs = x.requires_grad or w.requires_grad or (bias is not None and bias.requires_grad) or \
        (residual is not None and residual.requires_grad)
    out = Var(y, needs)
    out.stats = stats
    x.uses += 1
    if residual is not None:
        residual.uses += 1
    if (act != ops.ACT_NONE or ps_r > 1) and residual is None:
        out.epilogue = (act, ps_r)
    KH, KW, Cin, Cout = w.data.shape[1:]
    wshape = (KH, KW, Cin, Cout)
    B, H, W_, _ = x.data.shape
    Ho, Wo = ops.conv_out_hw(H, W_, KH, KW, stride, pad, transposed)

    def bwd():
        if out.grad is None:
            return
        dy = out.grad
        out.grad = None
        if out.grad_event is not None:       # the gradient was produced on another stream
            torch.cuda.current_stream().wait_event(out.grad_event)
            dy.record_stream(torch.cuda.current_stream())
        if not x.requires_grad and residual is None and ps_r == 1 and act != ops.ACT_NONE:
            # leaf layer (the depth-map branch): activation backward fused into the weight gradient
            dw, db = ops.conv2d_wgrad_act(x.data, dy, y, wshape, act, stride, pad, transposed,
                                          want_bias=bias is not None)
            accum(w, dw)
            if bias is not None:
                accum(bias, db)
            return
        if out.grad_is_preact:
            dconv = dy                       # the (single) consumer's dgrad already applied this epilogue's backward
        elif act != ops.ACT_NONE or ps_r > 1:
            dconv = ops.conv2d_epilogue_bwd(dy, y, Ho, Wo, Cout, act, ps_r,
                                            amax=want_amax(tape, dy, (B, Ho, Wo, Cout)) if isinstance(w.split, tuple) else None)
        else:
            dconv = dy
        dmax = None                          # (fp16 x 2 scheme) max |dconv|, shared by the weight and the data gradient
        if w.requires_grad or (bias is not None and bias.requires_grad):
            ws = _side_stream(dconv.device, "wgrad") if (WGRAD_STREAM and dconv.is_cuda) else None
            if ws is None and side_wgrad and TAIL_WGRAD_SIDE and SIDE_STREAM and dconv.is_cuda:
                ws = _side_stream(dconv.device)       # the depth-branch stream: idle while the HR tail runs backward
            if ws is None and _is_c9(w):
                dmax = _amax_t(dconv)
                dw, db = ops.conv9_wgrad_split2(x.data, _amax(x), dconv, dmax, want_bias=bias is not None)
            elif ws is None and w.split is not None and SPLIT_WGRAD:
                if isinstance(w.split, tuple):
                    dmax = _amax_t(dconv)
                    dw, db = ops.conv3x3_wgrad_split2(x.data, _amax(x), dconv, dmax, want_bias=bias is not None)
                else:
                    dw, db = ops.conv3x3_wgrad_split(x.data, dconv, want_bias=bias is not None)
            elif ws is None:
                dw, db = ops.conv2d_wgrad(x.data, dconv, wshape, stride, pad, transposed, want_bias=bias is not None)
            else:
                cur = torch.cuda.current_stream()
                ready = cur.record_event()
                if ws not in tape.side_streams:
                    tape.side_streams.append(ws)
                with torch.cuda.stream(ws):
                    ws.wait_event(ready)
                    dconv.record_stream(ws)
                    dw, db = ops.conv2d_wgrad(x.data, dconv, wshape, stride, pad, transposed,
                                              want_bias=bias is not None)
                    done = ws.record_event()
                w.grad_event = done
                if bias is not None:
                    bias.grad_event = done
            accum(w, dw)
            if bias is not None:
                accum(bias, db)
        if x.requires_grad:
            if (FUSE_DGRAD_ACT and f32 and x.grad is None and x.uses == 1 and x.epilogue is not None and
                    ops.conv2d_dgrad_act_supported(x.data.shape, w.data, dconv.shape, stride, pad, transposed,
                                                   x.epilogue[1])):
                # x = PixelShuffle(act(prev conv)) and this conv is its only consumer: write d(prev conv output)
                x.grad = ops.conv2d_dgrad_act(dconv, w.data, x.data, x.epilogue[0], x.epilogue[1], stride, pad,
                                              transposed)
                x.grad_is_preact = True
            elif _is_c9(w):
                if dmax is None:
                    dmax = _amax_t(dconv)
                x.grad = ops.conv9_dgrad_split2(dconv, dmax, w.data, w.split[1], x.data.shape, out=x.grad)
            elif isinstance(w.split, tuple):
                if dmax is None:
                    dmax = _amax_t(dconv)
                x.grad = ops.conv3x3_dgrad_split2(dconv, dmax, w.split, x.data.shape, out=x.grad)
            elif w.split is not None:
                if x.grad is None:
                    x.grad = ops.conv3x3_dgrad_split(dconv, w.split, x.data.shape)
                else:
                    ops.conv3x3_dgrad_split(dconv, w.split, x.data.shape, out=x.grad)
            elif x.grad is None:
                x.grad = ops.conv2d_dgrad(dconv, w.data, x.data.shape, stride, pad, transposed, out_dtype=x.data.dtype)
            else:
                ops.conv2d_dgrad(dconv, w.data, x.data.shape, stride, pad, transposed, out=x.grad)
        if residual is not None:
            accum(residual, dconv)

    tape.record(bwd)
    return out


def to_bf16(tape, x):
    """fp32 encoder -> bf16 trunk boundary (the encoder's first feature map feeds the head): one rounding forward, the
    bf16 gradient is added into the fp32 gradient of the source on the way back."""
    x.uses += 1
    out = Var(ops.cast_to_bf16(x.data), x.requires_grad)

    def bwd():
        if out.grad is None:
            return
        g = out.grad
        out.grad = None
        if x.requires_grad:
            if x.grad is None:
                x.grad = ops.cast_to_f32(g)
            else:
                ops.accumulate_(x.grad, g)

    tape.record(bwd)
    return out


def to_f32(tape, x):
    """bf16 -> fp32 (the encoder's last feature map feeds the fp32 region pooling); the gradient comes back rounded."""
    x.uses += 1
    out = Var(ops.cast_to_f32(x.data), x.requires_grad)

    def bwd():
        if out.grad is None:
            return
        g = out.grad
        out.grad = None
        if x.requires_grad:
            accum(x, ops.cast_to_bf16(g))

    tape.record(bwd)
    return out


def space_to_depth2(tape, x, valid_hw=None):
    """x -> bf16 space-to-depth image (ops.space_to_depth2); x fp32 or bf16.  ``valid_hw``: extents of x that exist in the
    reference (rows / columns beyond them read as zeros and get a zero gradient)."""
    x.uses += 1
    out = Var(ops.space_to_depth2(x.data, valid_hw), x.requires_grad)

    def bwd():
        if out.grad is None:
            return
        g = out.grad
        out.grad = None
        if x.requires_grad:
            if x.grad is None:
                x.grad = ops.depth_to_space2_bwd(g, x.data.shape, x.data.dtype, valid_hw=valid_hw)
            else:
                ops.depth_to_space2_bwd(g, x.data.shape, x.data.dtype, out=x.grad, valid_hw=valid_hw)

    tape.record(bwd)
    return out


def expand_s2(tape, wp):
    """fp32 packed kernel of a 3x3 stride-2 conv -> bf16 packed kernel of its stride-1 form on the space-to-depth image."""
    _, _, _, Cin, Cout = wp.data.shape
    # (eval + no_grad: wp.data is itself the cached folded kernel, so its identity / version key this second cache level)
    out = Var(_folded(tape, ("s2", "@%x" % wp.data.data_ptr()), [wp.data], lambda: ops.weight_expand_s2(wp.data)), wp.requires_grad)

    def bwd():
        if out.grad is None:
            return
        if out.grad_event is not None:
            torch.cuda.current_stream().wait_event(out.grad_event)
            out.grad.record_stream(torch.cuda.current_stream())
        g = out.grad
        out.grad = None
        accum(wp, ops.weight_collapse_s2(g, Cin, Cout))

    tape.record(bwd)
    return out


def expand_t2(tape, wp, bias):
    """fp32 packed kernel + bias of a 3x3 stride-2 ConvTranspose2d -> (bf16 packed kernel, bias) of the stride-1 convolution
    Cin -> 4 Cout whose PixelShuffle(2) it is."""
    _, _, _, Cin, Cout = wp.data.shape
    w4, b4 = _folded(tape, ("t2", "@%x" % wp.data.data_ptr()), [wp.data] + ([bias.data] if bias is not None else []),
                     lambda: ops.weight_expand_t2(wp.data, bias.data if bias is not None else None))
    wout = Var(w4, wp.requires_grad)
    bout = Var(b4, bias.requires_grad) if bias is not None else None

    def bwd():
        if wout.grad is None:
            return
        for v in (wout, bout):
            if v is not None and v.grad_event is not None:
                torch.cuda.current_stream().wait_event(v.grad_event)
                if v.grad is not None:
                    v.grad.record_stream(torch.cuda.current_stream())
        gw, gb = wout.grad, (bout.grad if bout is not None else None)
        wout.grad = None
        if bout is not None:
            bout.grad = None
        dw, db = ops.weight_collapse_t2(gw, gb, Cin, Cout)
        accum(wp, dw)
        if bias is not None and db is not None:
            accum(bias, db)

    tape.record(bwd)
    return wout, bout


def add(tape, a, b):
    a.uses += 1
    b.uses += 1
    out = Var(ops.add(a.data, b.data), a.requires_grad or b.requires_grad)

    def bwd():
        if out.grad is None:
            return
        g = out.grad
        out.grad = None
        accum(a, g)
        accum(b, g, owned=False)

    tape.record(bwd)
    return out


def region_pool(tape, feat, mask):
    feat.uses += 1
    st, maskr, area = ops.region_pool_fwd(feat.data, mask)
    out = Var(st, feat.requires_grad)

    def bwd():
        if out.grad is None:
            return
        accum(feat, ops.region_pool_bwd(out.grad, maskr, area, feat.data.shape))
        out.grad = None

    tape.record(bwd)
    return out


def dynk(tape, st, A_w, A_b, Wg, Wb):
    stp, D = ops.dynk_fwd(st.data, A_w.data, A_b.data, Wg.data, Wb.data)
    out = Var(D, True)

    def bwd():
        if out.grad is None:
            return
        if st.grad is None:
            st.grad = ops.zeros(st.data.shape, st.data)
        dWg, dWb, dAw, dAb = ops.dynk_bwd(out.grad, st.data, stp, A_w.data, Wg.data, Wb.data, st.grad)
        out.grad = None
        accum(Wg, dWg)
        accum(Wb, dWb)
        accum(A_w, dAw)
        accum(A_b, dAb)

    tape.record(bwd)
    return out


def sean_mod(tape, t, gb2, mask, D, bias_g, bias_b, alpha_g, alpha_b, residual, relu):
    """``mask`` is a MaskPack (float planes + region index + one-hot flag)."""
    t.uses += 1
    gb2.uses += 1
    if residual is not None:
        residual.uses += 1
    if gb2.event is not None:                # gamma2/beta2 were computed on the side stream
        torch.cuda.current_stream().wait_event(gb2.event)
        gb2.data.record_stream(torch.cuda.current_stream())
    mean, var = t.stats if t.stats is not None else ops.instnorm_stats(t.data)
    y = ops.sean_fwd(t.data, mean, var, gb2.data, mask.planes, mask.region, mask.flag, D.data, bias_g.data, bias_b.data,
                     alpha_g.data, alpha_b.data, residual.data if residual is not None else None, relu,
                     amax=want_amax(tape, t.data))
    out = Var(y, True)

    def bwd():
        if out.grad is None:
            return
        want_dres = residual is not None and residual.requires_grad
        dt, dgb2, dD, dbg, dbb, dag, dab, dres = ops.sean_bwd(
            out.grad, y, t.data, mean, var, gb2.data, mask.planes, mask.region, mask.flag, D.data, bias_g.data,
            bias_b.data, alpha_g.data, alpha_b.data, relu, want_dres,
            dt_amax=want_amax(tape, t.data), dgb2_amax=want_amax(tape, gb2.data))
        out.grad = None
        accum(t, dt)
        accum(gb2, dgb2)
        if gb2.event is not None:
            gb2.grad_event = torch.cuda.current_stream().record_event()
        accum(D, dD)
        accum(bias_g, dbg)
        accum(bias_b, dbb)
        accum(alpha_g, dag)
        accum(alpha_b, dab)
        if want_dres:
            accum(residual, dres)

    tape.record(bwd)
    return out


FORCE_GENERAL_SEAN = False


def attached_region(mask):
    """Region bytes that dasr_amd.prep.depth_to_masks attached to a mask tensor, or None.  They are trusted only while
    the tensor is unmodified: an in-place edit (a flip, a crop written back) bumps torch's version counter."""
    region = getattr(mask, "_dasr_region", None)
    if region is None:
        return None
    if tuple(region.shape) != (mask.shape[0],) + tuple(mask.shape[2:]) or region.device != mask.device:
        return None
    if getattr(mask, "_dasr_version", None) != ops.tensor_version(mask):
        return None
    return region


class MaskPack:
    """The depth masks as the reference delivers them ([B,K,H,W] float planes) plus their compressed form
    (one region byte per pixel and a device-side "not one-hot" flag, dasr_mask_compress).

    The host never reads the flag: with (region, flag) both set every SEAN call launches the gather kernel and the
    general kernel and the flag decides ON THE DEVICE which of them works (the other returns at once); masks prepared
    by prep.depth_to_masks are one-hot by construction (flag None: gather kernel only)."""
    __slots__ = ("planes", "region", "flag", "_resized")

    def __init__(self, planes, region=None, flag=None):
        self.planes = planes
        self._resized = {}
        if FORCE_GENERAL_SEAN:               # tests: run the soft-mask kernels on one-hot masks (independent code path)
            self.region, self.flag = None, None
            return
        if region is not None:               # prepared on the device (one-hot by construction) or resized from a pack
            self.region, self.flag = region, flag
            return
        self.region, self.flag = ops.mask_compress(planes)
        if planes.shape[1] > ops.soft_mask_max_regions():
            # more regions than the soft-mask kernels hold in LDS (K = 15, 16): the device cannot fall back to them, so
            # the flag has to be known here - the one place the host still reads it
            if int(self.flag.item()) != 0:
                raise NotImplementedError("dasr_amd: soft (non one-hot) masks are supported for up to %d regions, got %d"
                                          % (ops.soft_mask_max_regions(), planes.shape[1]))
            self.flag = None

    def resized(self, H, W):
        """F.interpolate(mask, mode='nearest') at a block's feature size (normalization.py:59), once per size per
        forward.  The region bytes are resized themselves (no second compression pass): nearest resize commutes with
        the one-hot encoding, and a pack that is not one-hot keeps its flag."""
        if tuple(self.planes.shape[2:]) == (H, W):
            return self
        hit = self._resized.get((H, W))
        if hit is None:
            planes = ops.resize_nearest_nchw(self.planes, H, W)
            if self.region is None:
                hit = MaskPack(planes)
            else:
                hit = MaskPack(planes, ops.resize_nearest_u8(self.region, H, W), self.flag)
            self._resized[(H, W)] = hit
        return hit

    @property
    def shape(self):
        return self.planes.shape


# ---------------------------------------------------------------------------------------------
# blocks
# ---------------------------------------------------------------------------------------------
def _wn(tape, P, prefix, transposed=False, dtype=torch.float32):
    return pack(tape, P[prefix + ".weight_v"], P[prefix + ".weight_g"], transposed, dtype)


def block_plan(cfg):
    """Channel plan of DepthNet.__init__ (sftmd_arch.py:879-889): [(module_name, kind, channels)]."""
    nb, scale = cfg["nb"], cfg["scale"]
    num_last_block = 1 if scale == 3 else int(math.log(scale, 2))
    plan = []
    for i in range(nb):
        ch = 32 if i > nb - num_last_block else cfg["nf"]
        kind = "depth" if i in cfg["which_ResBlk_depth"] else "classic"
        plan.append((("depth-residual%d" if kind == "depth" else "classic-residual%d") % (i + 1), kind, ch))
    return plan


def sean_depth_branch(tape, P, pre, depth_map):
    """gamma2 | beta2 of one SEAN (normalization.py:61,73-74): a function of the depth map and parameters only, so
    the forward plan may compute it ahead of the trunk, on another stream."""
    dt = tape.act_dtype
    w_m = pack(tape, P[pre + ".mlp_mask.0.weight"])                # the depth map stays fp32: fp32 kernel, dt output
    actv = conv(tape, depth_map, w_m, P[pre + ".mlp_mask.0.bias"], act=ops.ACT_RELU, out_dtype=dt)
    w_gb = pack_pair(tape, P[pre + ".mlp_gamma_o.weight"], P[pre + ".mlp_beta_o.weight"], dt)
    b_gb = bias_pair(tape, P[pre + ".mlp_gamma_o.bias"], P[pre + ".mlp_beta_o.bias"])
    return conv(tape, actv, w_gb, b_gb)


def sean(tape, P, pre, t, depth_map, mask, st, residual, relu, consts, gb2=None):
    """SEAN.forward (normalization.py:52-92) + the caller's ReLU / residual."""
    B, H, W, C = t.data.shape
    assert st.data.shape[1] == mask.shape[1], "depth matrix regions != mask channels"   # normalization.py:54
    assert st.data.shape[2] == P[pre + ".mlp_gamma_s.weight"].data.shape[1], "len_latent != depth matrix width"
    if P[pre + ".A_i_j.weight"].data.shape[0] != st.data.shape[1]:
        # the reference fails here inside nn.Conv2d (A_i_j expects depthRangeNum channels): same error class
        raise RuntimeError("SEAN.A_i_j expects %d depth regions, got %d mask channels"
                           % (P[pre + ".A_i_j.weight"].data.shape[0], st.data.shape[1]))
    if gb2 is None:
        gb2 = sean_depth_branch(tape, P, pre, depth_map)
    # gamma1 / beta1: per-sample dynamic kernels over the K-channel mask
    D = dynk(tape, st, P[pre + ".A_i_j.weight"], P[pre + ".A_i_j.bias"], P[pre + ".mlp_gamma_s.weight"],
             P[pre + ".mlp_beta_s.weight"])
    a_g = P[pre + ".alpha_gamma"] if (pre + ".alpha_gamma") in P else consts["alpha_gamma"]
    a_b = P[pre + ".alpha_beta"] if (pre + ".alpha_beta") in P else consts["alpha_beta"]
    return sean_mod(tape, t, gb2, mask, D, P[pre + ".mlp_gamma_s.bias"], P[pre + ".mlp_beta_s.bias"], a_g, a_b,
                    residual, relu)


def block_depth_inputs(x_hw, depth_map, mask):
    """Depth map / masks at the block's feature size: F.interpolate(..., mode='nearest'), normalization.py:58-59."""
    B = depth_map.data.shape[0]
    H, W = x_hw
    if tuple(depth_map.data.shape[1:3]) != (H, W):
        d = ops.resize_nearest_nchw(depth_map.data.view(B, 1, *depth_map.data.shape[1:3]), H, W)
        depth_map = Var(d.view(B, H, W, 1))
        mask = mask.resized(H, W)
    return depth_map, mask


def depth_block(tape, P, name, x, depth_map, mask, st, consts, gb2_pair=None):
    """Depth_Residual_Block_Mask.forward (sftmd_arch.py:826-834)."""
    B, H, W, C = x.data.shape
    depth_map, mask = block_depth_inputs((H, W), depth_map, mask)
    g1, g2 = gb2_pair if gb2_pair is not None else (None, None)
    dt = tape.act_dtype
    t1 = conv(tape, x, pack(tape, P[name + ".conv1.0.weight"], dtype=dt), P[name + ".conv1.0.bias"], want_stats=True)
    a = sean(tape, P, name + ".norm1", t1, depth_map, mask, st, None, True, consts, g1)
    t2 = conv(tape, a, pack(tape, P[name + ".conv2.0.weight"], dtype=dt), P[name + ".conv2.0.bias"], want_stats=True)
    return sean(tape, P, name + ".norm2", t2, depth_map, mask, st, x, True, consts, g2)


def classic_block(tape, P, name, x):
    """Classic_Residual_Block.forward (sftmd_arch.py:147-151)."""
    dt = tape.act_dtype
    h = conv(tape, x, _wn(tape, P, name + ".block.0", dtype=dt), P[name + ".block.0.bias"], act=ops.ACT_RELU,
             side_wgrad=True)
    return conv(tape, h, _wn(tape, P, name + ".block.2", dtype=dt), P[name + ".block.2.bias"], act=ops.ACT_RELU,
                residual=x, side_wgrad=True)


def upscale(tape, P, name, x, r, second):
    """upscale1/2/3 (sftmd_arch.py:891-908): conv -> PixelShuffle(r) -> LeakyReLU [-> conv -> LeakyReLU]."""
    dt = tape.act_dtype
    x = conv(tape, x, _wn(tape, P, name + ".0", dtype=dt), P[name + ".0.bias"], act=ops.ACT_LRELU, ps_r=r,
             side_wgrad=True)
    if second:
        x = conv(tape, x, _wn(tape, P, name + ".3", dtype=dt), P[name + ".3.bias"], act=ops.ACT_LRELU, side_wgrad=True)
    return x


def encoder_forward(tape, P, cfg, x0, depth_mask):
    """Encoder.forward (sftmd_arch.py:771-783) + RegionWiseAvgPooling: (e1, e5, st); e5 / st are None for a net without
    depth blocks (the reference's ``isBaseline`` short-circuit, :774-775)."""
    L = ops.ACT_LRELU
    e1 = conv(tape, x0, _wn(tape, P, "encoder.layer1"), P["encoder.layer1.bias"], act=L)
    e5 = st = None
    # (the bf16 kernels want channel counts in multiples of 32: true for the reference's 32/64/128/L = 256 or 32 plan)
    s2d_ok = all(P["encoder.layer%d.weight_v" % i].data.shape[0] % 32 == 0 and
                 P["encoder.layer%d.weight_v" % i].data.shape[1] % 32 == 0 for i in (2, 3, 4, 5))
    if len(cfg["which_ResBlk_depth"]) > 0 and tape.act_dtype == torch.bfloat16 and ENCODER_S2D and s2d_ok:
        # mixed precision: the stride-2 layers as stride-1 bf16-MFMA convolutions of space-to-depth images / with a
        # PixelShuffle(2) epilogue (csrc/s2d.hip); fp32 master weights, weight norm and its gradient as everywhere
        def s2(name, x, act, valid_hw=None):
            return conv(tape, space_to_depth2(tape, x, valid_hw), expand_s2(tape, _wn(tape, P, name)), P[name + ".bias"],
                        act=act)
        e2 = s2("encoder.layer2", e1, L)
        e3 = s2("encoder.layer3", e2, L)
        w4, b4 = expand_t2(tape, _wn(tape, P, "encoder.layer4", True), P["encoder.layer4.bias"])
        e4 = conv(tape, e3, w4, b4, act=L, ps_r=2)
        # sftmd_arch.py:748: ConvTranspose2d(3, stride=2, padding=1) WITHOUT output_padding -> (2H3-1) x (2W3-1); the
        # PixelShuffle image e4 is 2H3 x 2W3, its last row / column do not exist in the reference: layer5 must see its
        # zero padding there (and send no gradient back)
        H3, W3 = e3.data.shape[1], e3.data.shape[2]
        e5 = to_f32(tape, s2("encoder.layer5", e4, ops.ACT_NONE, (2 * H3 - 1, 2 * W3 - 1)))
        st = region_pool(tape, e5, depth_mask)
    elif len(cfg["which_ResBlk_depth"]) > 0:
        e2 = conv(tape, e1, _wn(tape, P, "encoder.layer2"), P["encoder.layer2.bias"], stride=2, act=L)
        e3 = conv(tape, e2, _wn(tape, P, "encoder.layer3"), P["encoder.layer3.bias"], stride=2, act=L)
        e4 = conv(tape, e3, _wn(tape, P, "encoder.layer4", True), P["encoder.layer4.bias"], stride=2,
                  transposed=True, act=L)
        e5 = conv(tape, e4, _wn(tape, P, "encoder.layer5"), P["encoder.layer5.bias"], stride=2)
        st = region_pool(tape, e5, depth_mask)
    return e1, e5, st


def depthnet_forward(tape, P, cfg, consts, inp, depth_map, depth_mask, region=None):
    """DepthNet.forward (sftmd_arch.py:912-950). ``inp`` [B,3,H,W], ``depth_map`` [B,1,h,w],
    ``depth_mask`` [B,K,h,w] are the caller's NCHW tensors; returns (out NCHW tensor, out Var).
    ``region``: the masks' region bytes [B,h,w] when they were prepared on the device (prep.depth_to_masks)."""
    pp = tape.prepack
    if pp is not None and not (PREPACK and tape.enabled and pp.lock.acquire(blocking=False)):
        pp = tape.prepack = None
    try:
        if pp is not None:
            pp.begin(P, Prepack.signature(inp, tape.act_dtype))
        out = _depthnet_forward(tape, P, cfg, consts, inp, depth_map, depth_mask, region)
        if pp is not None:
            pp.finish()
        return out
    finally:
        if pp is not None:
            pp.lock.release()


def _depthnet_forward(tape, P, cfg, consts, inp, depth_map, depth_mask, region):
    plan = block_plan(cfg)
    nb, scale = cfg["nb"], cfg["scale"]
    B = inp.shape[0]
    x0 = Var(ops.nchw_to_nhwc(inp))
    dm = Var(depth_map.reshape(B, depth_map.shape[2], depth_map.shape[3], 1))   # [B,1,h,w] == [B,h,w,1]
    L = ops.ACT_LRELU
    e1, _e5, st = encoder_forward(tape, P, cfg, x0, depth_mask)
    mask_pack = MaskPack(depth_mask, region) if st is not None else None
    # head (:920).  Mixed precision: the encoder (0.8 % of the FLOPs, feeds the fp32 depth matrix) stays fp32; its first
    # feature map is rounded to bf16 here and everything up to conv_output's fp32 result runs on bf16 activations
    adt = tape.act_dtype
    e1h = to_bf16(tape, e1) if adt == torch.bfloat16 else e1
    h1 = conv(tape, e1h, _wn(tape, P, "head.0", dtype=adt), P["head.0.bias"], act=L)
    fea_bef = conv(tape, h1, _wn(tape, P, "head.2", dtype=adt), P["head.2.bias"], act=L)

    # The depth-map branch of every SEAN (mlp_mask -> ReLU -> gamma_o|beta_o, 55 % of the forward FLOPs) depends on
    # the depth map and parameters only: it is issued on a side HIP stream ahead of the trunk, and its backward runs
    # there too, so its large convolutions fill the tails and small-kernel gaps of the trunk.
    executed = list(range(nb - 3)) + [nb - 2, nb - 1]
    branch = {}
    side = _side_stream(inp.device) if (SIDE_STREAM and inp.is_cuda and mask_pack is not None) else None
    if side is not None:
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        feat_hw = {}
        hw = (x0.data.shape[1], x0.data.shape[2])
        for i in executed:                 # feature size seen by block i (upscale1 before nb-2, upscale2 before nb-1)
            if i == nb - 2 and scale == 8:
                hw = (hw[0] * 2, hw[1] * 2)
            if i == nb - 1 and scale >= 4:
                hw = (hw[0] * 2, hw[1] * 2)
            feat_hw[i] = hw
        with tape.on_stream(side):
            for i in executed:
                name, kind, _ = plan[i]
                if kind != "depth":
                    continue
                d_i, _m = block_depth_inputs(feat_hw[i], dm, mask_pack)
                pair = []
                for norm in (".norm1", ".norm2"):
                    gb2 = sean_depth_branch(tape, P, name + norm, d_i)
                    gb2.event = side.record_event()
                    pair.append(gb2)
                branch[i] = tuple(pair)

    def run_block(i, x):
        name, kind, _ = plan[i]
        if kind == "depth":
            return depth_block(tape, P, name, x, dm, mask_pack, st, consts, branch.get(i))
        return classic_block(tape, P, name, x)

    # tape.mark(): gradient-bucket boundaries of the data-parallel harness (tape order = reverse backward order): the HR
    # tail, the later and the earlier half of the LR blocks, then everything recorded before the trunk (encoder, head and
    # the depth branch, whose backward is issued last)
    fea = fea_bef
    tape.mark()
    for i in range(nb - 3):                       # :923 — block index nb-3 is constructed but never called
        fea = run_block(i, fea)
        if i == (nb - 3) // 2 - 1:
            tape.mark()
    fea = add(tape, fea, fea_bef)                 # :931
    tape.mark()
    if scale == 8:
        fea = upscale(tape, P, "upscale1", fea, 2, True)
    fea = run_block(nb - 2, fea)
    if scale >= 4:
        fea = upscale(tape, P, "upscale2", fea, 2, True)
    fea = run_block(nb - 1, fea)
    fea = upscale(tape, P, "upscale3", fea, 3 if scale == 3 else 2, False)
    y = conv(tape, fea, pack(tape, P["conv_output.weight"]), P["conv_output.bias"], pad=4, side_wgrad=True)     # :948
    lo, hi = cfg["out_min"], cfg["out_max"]
    out = Var(ops.clamp_to_nchw(y.data, lo, hi), True)                                          # :950

    def bwd():
        if out.grad is None:
            return
        accum(y, ops.clamp_to_nchw_bwd(out.grad, y.data, lo, hi))
        out.grad = None

    tape.record(bwd)
    return out
