"""A minimal reverse-mode tape over HIP kernel calls.

The generator's forward is a fixed sequence of kernel launches; its backward is the same
sequence reversed, each step calling the matching ``*_bwd`` / dgrad / wgrad entry points.  The
tape records one closure per forward step; ``backward()`` pops and runs them.  Gradients that
fan in (residual connections, the depth matrix feeding every SEAN) accumulate in place with
``dasr_accumulate`` — there is no tracing compiler and no PyTorch autograd inside the net.
"""
from . import ops


class Var:
    """A device tensor plus its gradient slot."""
    __slots__ = ("data", "grad", "requires_grad", "name", "uses", "epilogue", "grad_is_preact", "event",
                 "grad_event", "stats", "split", "amax", "pack")

    def __init__(self, data, requires_grad=False, name=None):
        self.data = data
        self.grad = None
        self.requires_grad = requires_grad
        self.name = name
        self.uses = 0                # differentiable consumers recorded during the forward
        self.epilogue = None         # (act, ps_r) when produced by a conv with a fused activation / PixelShuffle
        self.grad_is_preact = False  # the consumer already applied the epilogue backward (dasr_conv2d_dgrad_act)
        self.event = None            # HIP event after the producing kernel when it ran on another stream
        self.grad_event = None       # HIP event after the kernel that produced .grad on another stream
        self.stats = None            # (mean, var) per (b, c) when the producing conv computed them in its epilogue
        self.split = None            # packed fp32 kernels: their bf16 three-piece / fp16 two-piece image (graph.SPLIT_BF16), built at first use
        self.amax = None             # (stream, max |data| device scalar) for the fp16 x 2 split scheme (graph._amax)
        self.pack = None             # packed kernels of a training step: their key in the step's graph.Prepack

    @property
    def shape(self):
        return self.data.shape


class Tape:
    """Backward closures in forward order.  A closure recorded inside ``on_stream(s)`` runs its backward on ``s``
    too (the depth-map branch of every SEAN lives on a side stream, see graph.depthnet_forward)."""

    def __init__(self, enabled=True, act_dtype=None):
        import torch
        self.enabled = enabled
        self.act_dtype = act_dtype if act_dtype is not None else torch.float32   # storage type of the trunk's activations
        self.nodes = []
        self._stream = None
        self.side_streams = []
        self.fold_cache = None      # inference only: {param name: (versions, packed tensor)}, see graph._folded
        self.prepack = None         # training: the network's graph.Prepack (every packed kernel of the step in two launches)
        self.marks = []             # node counts at which the forward plan called mark(): gradient-bucket boundaries
        self.on_mark = None         # called during backward() each time the tape has unwound below a mark

    def record(self, fn):
        if self.enabled:
            self.nodes.append((fn, self._stream))

    def mark(self):
        """Bucket boundary for data-parallel training: every parameter whose packing node was recorded AFTER this point
        has its gradient complete once backward() has unwound to here (harness.Trainer starts that bucket's all-reduce
        from ``on_mark`` while the rest of the backward is still being issued)."""
        if self.enabled:
            self.marks.append(len(self.nodes))

    def on_stream(self, stream):
        tape = self

        class _Ctx:
            def __enter__(self_inner):
                import torch
                self_inner.prev = tape._stream
                tape._stream = stream
                if stream not in tape.side_streams:
                    tape.side_streams.append(stream)
                self_inner.guard = torch.cuda.stream(stream)
                self_inner.guard.__enter__()

            def __exit__(self_inner, *exc):
                self_inner.guard.__exit__(*exc)
                tape._stream = self_inner.prev

        return _Ctx()

    def backward(self):
        import torch
        while self.nodes:
            fn, stream = self.nodes.pop()
            if stream is None:
                fn()
            else:
                with torch.cuda.stream(stream):
                    fn()
            while self.marks and len(self.nodes) <= self.marks[-1]:
                self.marks.pop()
                if self.on_mark is not None:
                    self.on_mark()
        for s in self.side_streams:          # parameter gradients produced on side streams are complete
            torch.cuda.current_stream().wait_stream(s)


def accum(var, g, owned=True):
    """Add gradient ``g`` into ``var``.  ``owned=False`` means ``g`` is shared with another
    consumer, so the first assignment must take a private copy before anyone adds into it."""
    if var is None or not var.requires_grad or g is None:
        return
    if var.grad is None:
        if owned:
            var.grad = g
        else:
            var.grad = ops.copy_(ops.empty(g.shape, g, g.dtype), g)
    else:
        ops.accumulate_(var.grad, g)
