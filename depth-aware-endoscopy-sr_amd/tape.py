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
    __slots__ = ("data", "grad", "requires_grad", "name", "uses", "epilogue", "grad_is_preact")

    def __init__(self, data, requires_grad=False, name=None):
        self.data = data
        self.grad = None
        self.requires_grad = requires_grad
        self.name = name
        self.uses = 0                # differentiable consumers recorded during the forward
        self.epilogue = None         # (act, ps_r) when produced by a conv with a fused activation / PixelShuffle
        self.grad_is_preact = False  # the consumer already applied the epilogue backward (dasr_conv2d_dgrad_act)

    @property
    def shape(self):
        return self.data.shape


class Tape:
    def __init__(self, enabled=True):
        self.enabled = enabled
        self.nodes = []

    def record(self, fn):
        if self.enabled:
            self.nodes.append(fn)

    def backward(self):
        while self.nodes:
            self.nodes.pop()()


def accum(var, g, owned=True):
    """Add gradient ``g`` into ``var``.  ``owned=False`` means ``g`` is shared with another
    consumer, so the first assignment must take a private copy before anyone adds into it."""
    if var is None or not var.requires_grad or g is None:
        return
    if var.grad is None:
        if owned:
            var.grad = g
        else:
            var.grad = ops.copy_(ops.empty(g.shape, g), g)
    else:
        ops.accumulate_(var.grad, g)
