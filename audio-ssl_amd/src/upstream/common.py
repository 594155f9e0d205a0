"""Pieces shared by the DeLoRes experts: the Barlow `Projection` head and the fused-step plumbing."""
import torch
import torch.nn as nn

from src import _native as N
from src import engine as E
from src.flat import FlatGroup


class Projection(nn.Module):
    """3-layer MLP projector + affine-free BN + cross-correlation loss
    (`src/upstream/delores_s/upstream_expert.py:11-46`; lambda coerced with float()).
    The layers are parameter containers; forward/backward run in `engine.barlow_forward_backward`."""

    def __init__(self, in_dim, lambd=5e-5, scale_loss=1 / 32):
        super().__init__()
        sizes = [in_dim, 2048, 2048, 2048]
        layers = []
        for i in range(len(sizes) - 2):
            layers += [nn.Linear(sizes[i], sizes[i + 1], bias=False), nn.BatchNorm1d(sizes[i + 1]), nn.ReLU(inplace=True)]
        layers.append(nn.Linear(sizes[-2], sizes[-1], bias=False))
        self.projector = nn.Sequential(*layers)
        self.lambd = float(lambd)
        self.scale_loss = float(eval(scale_loss)) if isinstance(scale_loss, str) else float(scale_loss)
        self.bn = nn.BatchNorm1d(sizes[-1], affine=False)

    def param_dict(self):
        P = {n: p.data for n, p in self.named_parameters()}
        P.update({n: b for n, b in self.named_buffers()})
        return P

    def forward(self, y1, y2):
        return _ProjectionFn.apply(self, y1, y2, *[p for p in self.parameters()])


class _ProjectionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, y1, y2, *params):
        dt = N.F32 if y1.dtype == torch.float32 else N.BF16
        loss = torch.zeros(1, dtype=torch.float32, device=y1.device)
        G = {n: torch.zeros_like(p, dtype=torch.float32) for n, p in mod.named_parameters()}
        need1, need2 = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        B = y1.shape[0]
        dY = E.barlow_forward_backward(mod.param_dict(), G, torch.cat([y1, y2]).contiguous(), dt, mod.lambd,
                                       mod.scale_loss, loss, need_dy1=need1 or need2, need_dy2=need2,
                                       update_running=mod.training, all_reduce=getattr(mod, "all_reduce", None),
                                       global_batch=getattr(mod, "global_batch", None))
        dy1 = dY[:B] if (dY is not None and need1) else None
        dy2 = dY[B:] if (dY is not None and need2) else None
        ctx.saved = (dy1, dy2, [G[n] for n, _ in mod.named_parameters()])
        ctx.in_dtype = y1.dtype
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        dy1, dy2, grads = ctx.saved
        return (None, None if dy1 is None else (dy1 * g).to(ctx.in_dtype), None if dy2 is None else (dy2 * g).to(ctx.in_dtype)) + \
            tuple(gr * g for gr in grads)


def strip(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def module_buffers(mod, prefix=""):
    return {prefix + n: b for n, b in mod.named_buffers()}


def make_group(module, prefix=""):
    named = [(prefix + n, p) for n, p in module.named_parameters()]
    return FlatGroup(named)


class FusedStepFn(torch.autograd.Function):
    """One autograd node for the whole training step.  Its forward runs the fused HIP forward AND backward
    (gradients land in the expert's flat gradient buffer); its backward only publishes them through p.grad."""

    @staticmethod
    def forward(ctx, expert, need_grad, img_1, img_2, *params):
        ctx.expert = expert
        ctx.need_grad = need_grad
        return expert.fused_loss(img_1, img_2, need_grad)

    @staticmethod
    def backward(ctx, g):
        ex = ctx.expert
        if not ctx.need_grad:
            raise RuntimeError("training_step ran under no_grad; nothing to back-propagate")
        ex.publish_grads(g)
        return (None, None, None, None) + (None,) * len(ex.flat.params)


class FusedExpertMixin:
    """Flat storage + optimiser + gradient publication shared by the experts."""
    flat = None
    hip_optimizer = None
    precision = None

    def trainable_named(self):
        return [(n, p) for n, p in self.named_parameters() if p.requires_grad]

    def ensure_flat(self):
        named = self.trainable_named()
        dev = named[0][1].device
        ok = self.flat is not None and self.flat.data.device == dev and len(self.flat.params) == len(named)
        if ok:
            p0, pl = named[0][1], named[-1][1]
            ok = p0.data_ptr() == self.flat.data.data_ptr() and \
                pl.data_ptr() == self.flat.data.data_ptr() + 4 * self.flat.offsets[-1]
        if not ok:
            for _, p in named:
                p.grad = None
            self.flat = FlatGroup(named)
            self.on_reflatten()
            if self.hip_optimizer is not None:
                self.hip_optimizer.flat_groups = [self.flat]
        return self.flat

    def on_reflatten(self):
        pass

    def publish_grads(self, g):
        if self.hip_optimizer is None:
            self.flat.grad.mul_(g)            # generic optimisers read p.grad: apply the upstream gradient
        else:
            self.hip_optimizer.grad_scale_tensor = g
        self.flat.attach_grads()

    def configure_optimizers(self):
        from src.optim import HipSGD
        self.ensure_flat()
        self.hip_optimizer = HipSGD([self.flat], [p for _, p in self.trainable_named()], self.hparams.learning_rate,
                                    momentum=self.hparams.momentum, weight_decay=self.hparams.weight_decay)
        return self.hip_optimizer

    # ---- data-parallel gradient reduction (RCCL over xGMI) --------------------------------------------------------
    # The flat gradient buffer is reduced in two large segments, each launched (async, on RCCL's own stream) the moment
    # its producers have been enqueued: the loss heads' segment (83 % of the bytes for delores_m) right after the
    # heads' backward, i.e. BEFORE the encoder backward runs, so most of the traffic hides under compute; the encoder
    # segment after the encoder backward.  `all_reduce_grads()` joins them; 1/world is folded into the SGD launch.
    _pending = None

    def head_offset(self):
        """First flat-buffer element that belongs to a loss head (the encoder parameters come first)."""
        for n, o in zip(self.flat.names, self.flat.offsets):
            if not n.startswith(("encoder.", "encoder_q.")):
                return o
        return self.flat.numel

    def reduce_begin(self, which):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        ho = self.head_offset()
        lo, hi = (ho, self.flat.numel) if which == "heads" else (0, ho)
        if hi > lo:
            if self._pending is None:
                self._pending = []
            self._pending.append((which, dist.all_reduce(self.flat.grad[lo:hi], async_op=True)))

    def all_reduce_grads(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        done = {w for w, _ in (self._pending or [])}
        for which in ("heads", "enc"):
            if which not in done:
                self.reduce_begin(which)
        for _, work in self._pending:
            work.wait()                       # the current stream waits for RCCL's stream
        self._pending = None
        if self.hip_optimizer is not None:
            self.hip_optimizer.grad_scale = 1.0 / dist.get_world_size()
        else:
            self.flat.grad.div_(dist.get_world_size())
