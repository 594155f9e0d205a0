"""Pieces shared by the DeLoRes experts: the Barlow `Projection` head and the fused-step plumbing."""
import os

import torch
import torch.nn as nn

from src import _native as N
from src import engine as E
from src.flat import FlatGroup, cached_param_dict
from src.utils import concat_all_gather



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
        return cached_param_dict(self)

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


class EagerPhases:
    """Phase runner of the eagerly issued step: a phase is just a call, buffers are fresh allocations."""

    def phase(self, name, fn):
        return fn()

    def static(self, name, t):
        return t

    def alloc(self, name, make):
        return make()


EAGER = EagerPhases()


class GraphPhases:
    """Phase runner of the data-parallel graph step.  The step is cut at its collectives into phases (query encoder,
    key encoder, loss heads, encoder backward); each phase is captured into its own hipGraph the first time it runs and
    replayed afterwards on whatever stream is current, so the key encoder still overlaps the query encoder.  The RCCL
    calls between the phases stay ordinary eager calls.  Tensors handed from a phase to a later one are outputs of the
    capture (fixed addresses); tensors produced by a collective are copied into fixed buffers with `static`."""

    def __init__(self):
        self.graphs, self.results, self.buffers = {}, {}, {}
        self.broken = None                 # set to the exception text if a capture failed: phases then run eagerly

    def phase(self, name, fn):
        if self.broken is not None:
            return fn()
        g = self.graphs.get(name)
        if g is None:
            g = torch.cuda.CUDAGraph()
            try:
                # thread-local capture mode: RCCL's watchdog thread polls its events while we capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"), E.capture_scope():
                    self.results[name] = fn()
            except RuntimeError as e:          # nothing of the phase has executed yet: issue it eagerly from now on
                import warnings
                self.broken = f"{name}: {e}"
                warnings.warn(f"hipGraph capture of phase '{name}' failed ({e}); this rank issues the step eagerly")
                return fn()
            self.graphs[name] = g
        g.replay()
        return self.results[name]

    def static(self, name, t):
        buf = self.buffers.get(name)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            if name in self.buffers:
                raise RuntimeError(f"graph step: shape of '{name}' changed; build a new graphed step")
            buf = self.buffers[name] = torch.empty_like(t)
        buf.copy_(t)
        return buf

    def alloc(self, name, make):
        if name not in self.buffers:
            self.buffers[name] = make()
        return self.buffers[name]


class GraphedStep:
    """The whole training step of one rank - zero the flat gradient, fused forward + backward, optimiser - captured once
    into a hipGraph and replayed with a single launch per step.

    Why: the step is ~270 kernels of 5-200 us each; issued one by one the launch side costs as much wall time as the GPU
    needs to run them.  What makes the capture valid: every per-step scalar that changes (dropout
    counter, MoCo queue pointer, BatchNorm batch counters) lives in device memory, shapes are fixed, and the side
    streams (key encoder, loss heads, weight gradients) fork from and rejoin the capturing stream.

    The first `eager_steps` calls run the ordinary eager path (they size the allocator pools, initialise the momentum
    buffer and set the kernels' LDS attributes); the next call captures, every call from then on copies the two views
    into the graph's input buffers and replays.  A change of batch shape or of the optimiser's hyper-parameters triggers
    a new capture.  Data-parallel ranks (world size > 1) cannot put RCCL calls inside one graph: there the step is cut
    at its collectives and every collective-free phase is its own graph (`GraphPhases`).
    """

    def __init__(self, expert, optimizer, eager_steps=2, world=1, phases=False):
        self.expert, self.opt, self.eager_steps, self.world = expert, optimizer, eager_steps, world
        # experts whose step has several independent branches replay one graph per phase even on a single rank: ROCm's
        # graph executor runs the branches of ONE graph back to back, separately launched graphs on separate streams overlap
        self.use_phases = phases or world > 1 or getattr(expert, "prefer_phases", False)
        self.calls = 0
        self.graph = None
        self.phases = None                 # GraphPhases of the data-parallel variant
        import inspect
        self._takes_optimizer = "optimizer" in inspect.signature(expert.fused_loss).parameters
        self.key = None
        self.replays = 0
        # The SGD pass also writes the bf16 weight copies and clears the gradients (nobody reads the gradients between the
        # optimiser pass and the next forward): the cast sweep over the 38 M parameters (228 MB, ~65 us) and the gradient clear
        # leave the next step.  Round 1 measured this as a small loss (2.86 vs 2.81 ms: the sweeps ran beside the key
        # encoder's convolutions); with the serial kernels shorter it is a gain (2.41 vs 2.43-2.48 ms).  AUDIOSSL_FUSED_REFRESH=0
        # switches it off.
        if hasattr(optimizer, "fused_refresh") and os.environ.get("AUDIOSSL_FUSED_REFRESH", "1") == "1":
            optimizer.fused_refresh = True

    def _hyper(self):
        g = self.opt.param_groups[0]
        return tuple((k, g[k]) for k in sorted(g) if k != "params" and isinstance(g[k], (int, float)))

    def _eager(self, img_1, img_2, runner=None):
        kw = {} if runner is None else {"runner": runner}
        if self._takes_optimizer:
            kw["optimizer"] = self.opt                 # lets the expert step finished parameter segments early
        self.opt.grad_scale_tensor = None
        loss = self.expert.fused_loss(img_1, img_2, True, **kw)
        self.expert.all_reduce_grads()             # no-op on one rank
        # No host drain here.  Round 1 needed one to hide a wrong update in the two-rank graph step; its cause was the
        # hipMemsetAsync nodes inside the phase graphs (csrc/common.h ASSL_ZERO), which no longer exist.
        self.opt.grad_scale_tensor = None
        self.opt.step()
        return loss

    @torch.no_grad()
    def __call__(self, img_1, img_2):
        self.calls += 1
        if self.calls <= self.eager_steps:
            return self._eager(img_1, img_2)
        key = (tuple(img_1.shape), tuple(img_2.shape), img_1.dtype, img_2.dtype, self._hyper(), self.expert.training,
               self.expert.flat.data.data_ptr(), getattr(self.expert, "graph_key", tuple)())
        if self.use_phases:
            # data parallel: one graph per collective-free phase, RCCL calls in between
            if self.phases is None or key != self.key:
                self.phases, self.key = GraphPhases(), key

            a, b = self.phases.static("img_1", img_1), self.phases.static("img_2", img_2)
            self.replays += 1
            return self._eager(a, b, runner=self.phases)
        if self.graph is None or key != self.key:
            self.in_1, self.in_2 = torch.empty_like(img_1), torch.empty_like(img_2)
            torch.cuda.synchronize()
            # AUDIOSSL_GRAPH_COPIES instantiations of the same step, replayed in turn (same buffers, same stream: the
            # device-side order is unchanged): launching an executable graph again while its previous launch is still
            # running makes the runtime wait on the host before it enqueues anything
            n = max(1, int(os.environ.get("AUDIOSSL_GRAPH_COPIES", "1")))
            self.graphs, self.losses = [], []
            for _ in range(n):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g), E.capture_scope():
                    self.losses.append(self._eager(self.in_1, self.in_2))
                self.graphs.append(g)
            self.graph, self.loss = self.graphs[0], self.losses[0]
            self.key = key
        E.fast_copy(self.in_1, img_1)
        E.fast_copy(self.in_2, img_2)
        i = self.replays % len(self.graphs)
        self.graphs[i].replay()
        self.replays += 1
        return self.losses[i]


def _world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class MocoQueueMixin:
    """MoCo-v2 machinery shared by the experts with a query/key encoder pair and a negatives queue (delores_m, slicer):
    EMA of the key encoder as ONE launch over the flat buffers, enqueue with the write pointer on the device, and the
    data-parallel batch shuffle / unshuffle (`src/upstream/delores_m/upstream_expert.py:147-219` of the reference; the
    removed `trainer.use_ddp` switch is replaced by the state of the process group)."""
    flat_k = None

    def on_reflatten(self):
        self.flat_k = FlatGroup([(n, p) for n, p in self.encoder_k.named_parameters()])

    @torch.no_grad()
    def _momentum_update_key_encoder(self, shadow_dtype=None):
        """EMA of the key encoder (`delores_m/upstream_expert.py:93-99`); with shadow_dtype = bf16 the same pass leaves the bf16
        copy of the updated weights (the cast launch of `flat_k.refresh_shadow` is then skipped)."""
        self.ensure_flat()
        sh = self.flat_k.fresh_shadow_buffer(shadow_dtype) if shadow_dtype is not None else None
        N.call("ema_update", self.flat_k.data, self.flat.data, self.flat_k.numel, float(self.hparams.encoder_momentum), sh)

    def queue_shadow(self, dtype):
        """bf16 copy of the negatives' queue, kept current by `enqueue` (which writes the new keys into both): one full cast
        when it is first needed or after the queue tensor was replaced / overwritten by torch (load_state_dict, .to())."""
        if dtype == N.F32:
            return self.queue
        tag = (self.queue.data_ptr(), self.queue._version)
        if getattr(self, "_qshadow", None) is None or self._qshadow_tag != tag:
            self._qshadow = E.cast(dtype, self.queue)
            self._qshadow_tag = tag
        return self._qshadow

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys32, shadow):
        if _world() > 1:
            keys32 = concat_all_gather(keys32)
        batch_size = keys32.shape[0]
        K = self.hparams.num_negatives
        assert K % batch_size == 0  # for simplicity
        # the write position lives in the `queue_ptr` buffer and is advanced on the device (no host mirror: the step
        # can be captured in a hipGraph and replayed)
        N.call("enqueue", self.precision, keys32, batch_size, keys32.shape[1], K, 0, self.queue_ptr, self.queue, shadow)

    @torch.no_grad()
    def _shuffle_begin(self, x):
        """Start the key-batch all-gather (13 MB per rank at B=512) on RCCL's stream; it overlaps the query encoder."""
        import torch.distributed as dist
        out = torch.empty((_world() * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        return out, dist.all_gather_into_tensor(out, x.contiguous(), async_op=True)

    @torch.no_grad()
    def _shuffle_end(self, pending, batch_size_this):
        import torch.distributed as dist
        x_gather, work = pending
        work.wait()
        batch_size_all = x_gather.shape[0]
        num_gpus = batch_size_all // batch_size_this
        idx_shuffle = torch.randperm(batch_size_all, device=x_gather.device)
        dist.broadcast(idx_shuffle, src=0)
        idx_unshuffle = torch.argsort(idx_shuffle)
        idx_this = idx_shuffle.view(num_gpus, -1)[dist.get_rank()]
        return x_gather[idx_this], idx_unshuffle

    @torch.no_grad()
    def _batch_shuffle_ddp(self, x):
        import torch.distributed as dist
        batch_size_this = x.shape[0]
        x_gather = concat_all_gather(x)
        batch_size_all = x_gather.shape[0]
        num_gpus = batch_size_all // batch_size_this
        idx_shuffle = torch.randperm(batch_size_all, device=x.device)
        dist.broadcast(idx_shuffle, src=0)
        idx_unshuffle = torch.argsort(idx_shuffle)
        idx_this = idx_shuffle.view(num_gpus, -1)[dist.get_rank()]
        return x_gather[idx_this], idx_unshuffle

    @torch.no_grad()
    def _batch_unshuffle_ddp(self, x, idx_unshuffle):
        import torch.distributed as dist
        batch_size_this = x.shape[0]
        x_gather = concat_all_gather(x)
        num_gpus = x_gather.shape[0] // batch_size_this
        idx_this = idx_unshuffle.view(num_gpus, -1)[dist.get_rank()]
        return x_gather[idx_this]


class FusedExpertMixin:
    """Flat storage + optimiser + gradient publication shared by the experts."""
    flat = None
    hip_optimizer = None
    precision = None

    def trainable_named(self):
        return [(n, p) for n, p in self.named_parameters() if p.requires_grad]

    def ensure_flat(self):
        if self.flat is not None and self.flat.intact():       # per-step fast path: no walk over the module tree
            return self.flat
        named = self.trainable_named()
        for _, p in named:
            p.grad = None
        self.flat = FlatGroup(named)
        self.on_reflatten()
        if self.hip_optimizer is not None:
            self.hip_optimizer.flat_groups = [self.flat]
        return self.flat

    def on_reflatten(self):
        pass

    def graphed_step(self, optimizer=None, eager_steps=2, phases=False):
        """-> callable(img_1, img_2) -> loss that runs zero_grad + fused forward/backward [+ gradient all-reduce] +
        optimiser from captured hipGraphs: ONE graph on a single rank, one graph per collective-free phase with the RCCL
        calls in between on data-parallel ranks.  The returned loss tensor is overwritten by the next call."""
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if world > 1 and not self.graph_phases_supported():
            raise RuntimeError("this configuration has a collective inside a phase; use training_step + all_reduce_grads")
        opt = optimizer or self.hip_optimizer or self.configure_optimizers()
        self.ensure_flat()
        return GraphedStep(self, opt, eager_steps, world, phases)

    def graph_phases_supported(self):
        from src import engine as E
        return E.SYNC_BN is None                # SyncBatchNorm puts all-reduces inside the phases: eager steps

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
