"""Fused SGD-with-momentum over flat parameter groups (`torch.optim.SGD` semantics of
`src/upstream/delores_s/upstream_expert.py:236-243`: wd added to the gradient, first step buf = g, no dampening)."""
import torch

from src import _native as N


class FlatState:
    """state_dict / load_state_dict of the flat optimisers.  Their state is not torch's per-parameter dict but one buffer per
    flat group (momentum, or Adam's two moments + step count), so the schema is this package's own and says so
    (`schema: audiossl-flat-v1`): a checkpoint written by Lightning / torch.optim (`{state, param_groups}`) is recognised and
    declined - load_state_dict returns False and the optimiser starts with fresh state - instead of dying on a missing key."""
    SCHEMA = "audiossl-flat-v1"
    state_buffers = ("momentum",)

    def state_dict(self):
        sd = {"schema": self.SCHEMA, "kind": type(self).__name__, "steps": int(getattr(self, "steps", 0)),
              "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}
        for name in self.state_buffers:
            sd[name] = [None if getattr(fg, name, None) is None else getattr(fg, name).clone() for fg in self.flat_groups]
        sc = getattr(self, "step_count", None)
        if sc is not None:
            sd["step_count"] = int(sc.item())
        return sd

    def load_state_dict(self, sd):
        if not isinstance(sd, dict) or sd.get("schema") != self.SCHEMA or sd.get("kind") != type(self).__name__ or \
                any(len(sd.get(n, ())) != len(self.flat_groups) for n in self.state_buffers):
            import warnings
            warnings.warn(f"{type(self).__name__}: optimiser state of another schema "
                          f"({sorted(sd) if isinstance(sd, dict) else type(sd).__name__}); starting with fresh optimiser state")
            return False
        self.steps = int(sd["steps"])
        for name in self.state_buffers:
            for fg, m in zip(self.flat_groups, sd[name]):
                if m is not None and m.numel() != fg.numel:
                    raise ValueError(f"{name}: {m.numel()} elements in the checkpoint, {fg.numel} in the model")
                setattr(fg, name, None if m is None else m.to(fg.data.device).clone())
        for g, saved in zip(self.param_groups, sd.get("param_groups", ())):
            g.update({k: v for k, v in saved.items() if k in g and k != "params"})
        if "step_count" in sd and self.flat_groups:
            self.step_count = torch.full((1,), int(sd["step_count"]), dtype=torch.int64, device=self.flat_groups[0].data.device)
        return True


class HipSGD(FlatState, torch.optim.Optimizer):
    def __init__(self, flat_groups, params, lr, momentum=0.9, weight_decay=0.0, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.flat_groups = list(flat_groups)
        self.grad_scale = grad_scale
        self.grad_scale_tensor = None      # optional device scalar multiplied in (loss.backward(g))
        self.steps = 0
        self._early = {}                   # id(flat group) -> first element already stepped by step_tail() this step
        self._zero_tables = {}
        # fused_refresh (set by GraphedStep): the SGD pass also writes the bf16 shadow of the updated weights and clears the
        # gradient, and tells the flat group so - its next refresh_shadow() / zero_grad() are then no-ops.  Off by default:
        # with it, gradients read back as zero after step().
        self.fused_refresh = False

    def _tail(self, fg, lo, hi):
        fused = self.fused_refresh and fg._shadow is not None
        return (fg._shadow[lo:hi] if fused else None), int(fused)

    @torch.no_grad()
    def step(self, closure=None):
        g0 = self.param_groups[0]
        for fg in self.flat_groups:
            first = fg.momentum is None
            if first:
                fg.momentum = torch.empty_like(fg.data)
            lo, hi = 0, fg.numel
            done = self._early.pop(id(fg), None)
            if done is not None:                      # [done, numel) was already stepped by step_tail() during this step
                hi = done
            shadow, zero = self._tail(fg, lo, hi)
            if hi > lo:
                N.call("sgd_momentum", fg.data[lo:hi], fg.grad[lo:hi], fg.momentum[lo:hi], hi - lo, float(g0["lr"]),
                       float(g0["momentum"]), float(g0["weight_decay"]), int(first), float(self.grad_scale), self.grad_scale_tensor,
                       shadow, zero)
            if zero and (done is None or fg._early_fused):
                fg.mark_fresh(partial=fg._early_fused == "partial")   # the whole buffer has been stepped with the fused tail
            fg._early_fused = False
        self.steps += 1

    @torch.no_grad()
    def step_tail(self, fg, start, stored=(), stepped=False):
        """Step the slice [start, numel) of one flat group now, on the current stream; the next step() covers the rest.
        Used by the fused experts to update the loss-head parameters (83 % of delores_m's buffer) as soon as their
        gradients are final, underneath the encoder backward, instead of in the serial tail of the step.
        stored: names of parameters in the slice whose gradient is STORED by its only writer every step (the heads' weight
        gradients): the fused tail then clears only the other tensors of the slice (one small launch over a segment table)
        instead of writing zeros over 126 MB that the next step overwrites anyway; the flat group is told (`partial`)."""
        if fg.momentum is None or start >= fg.numel or start % 64:
            return False
        g0 = self.param_groups[0]
        shadow, zero = self._tail(fg, start, fg.numel)
        segs = self._zero_table(fg, start, stored) if (zero and stored) else None
        if stepped:
            # stepped: the `stored` tensors were already updated by their weight-gradient GEMMs (fused_wgrad) - the SGD pass covers the
            # other tensors of the slice only, and clears their gradients in the same launch
            if segs is None:
                raise RuntimeError("step_tail(stepped=True) needs the fused tail and the list of stepped tensors")
            N.call("sgd_momentum_segments", fg.data, fg.grad, fg.momentum, segs[0], segs[1], segs[2], float(g0["lr"]),
                   float(g0["momentum"]), float(g0["weight_decay"]), 0, float(self.grad_scale), self.grad_scale_tensor, fg._shadow, 1)
            fg._early_fused = "partial"
            return True
        N.call("sgd_momentum", fg.data[start:], fg.grad[start:], fg.momentum[start:], fg.numel - start, float(g0["lr"]),
               float(g0["momentum"]), float(g0["weight_decay"]), 0, float(self.grad_scale), self.grad_scale_tensor, shadow,
               0 if segs is not None else zero)
        if segs is not None:
            N.call("zero_segments", fg.grad, segs[0], segs[1], segs[2])
        fg._early_fused = ("partial" if segs is not None else True) if zero else False
        return True

    def fused_wgrad(self, fg, prefixes):
        """Context for weight-gradient GEMMs that apply this optimiser's update themselves (engine.gemm_multi_sgd): per prefix a
        dict name -> momentum view, plus the hyper-parameters of the launch.  None while the momentum buffer does not exist (first
        step) or the fused tail is off."""
        if fg.momentum is None or not self.fused_refresh or fg._shadow is None:
            return None
        key = ("mom", fg.momentum.data_ptr(), tuple(prefixes))
        views = fg._views.get(key)
        if views is None:
            views = fg._views[key] = [{n[len(pre):]: fg.momentum[o:o + p.numel()].view(p.shape)
                                       for n, p, o in zip(fg.names, fg.params, fg.offsets) if n.startswith(pre)} for pre in prefixes]
        g0 = self.param_groups[0]
        return {"momentum": views, "hyper": (float(g0["lr"]), float(g0["momentum"]), float(g0["weight_decay"]), float(self.grad_scale),
                                             self.grad_scale_tensor)}

    def _zero_table(self, fg, start, stored):
        """Device table of the (offset, length) runs of [start, numel) that are NOT gradients of `stored` parameters; None when
        there is nothing to skip."""
        key = (id(fg), start, tuple(stored))
        hit = self._zero_tables.get(key)
        if hit is None:
            skip = set(stored)
            runs = []
            for n, p, o in zip(fg.names, fg.params, fg.offsets):
                if o + p.numel() <= start or n in skip:
                    continue
                lo = max(o, start)
                if runs and runs[-1][0] + runs[-1][1] == lo:
                    runs[-1][1] += o + p.numel() - lo
                else:
                    runs.append([lo, o + p.numel() - lo])
            covered = sum(r[1] for r in runs)
            if not runs or covered == fg.numel - start:
                hit = (None,)
            else:
                table = torch.tensor([v for r in runs for v in r], dtype=torch.int64, device=fg.data.device)
                hit = ((table, len(runs), max(r[1] for r in runs)),)
            self._zero_tables[key] = hit
        return hit[0]

    def mark_early(self, fg, start):
        """Host-side bookkeeping for step_tail (kept separate: a replayed graph re-issues the launch, not this)."""
        self._early[id(fg)] = start

    def zero_grad(self, set_to_none=True):
        for fg in self.flat_groups:
            for p in fg.params:
                p.grad = None


class HipLARS(FlatState, torch.optim.Optimizer):
    """LARS of `extras/delores-s/multi_proc.py:4-43` on flat parameter groups: two launches per group (per-tensor
    norms, then the fused trust-ratio / momentum / update).  `lr_weights` applies to tensors with ndim > 1, `lr_biases`
    to 1-D tensors (the reference's two param groups, `adjust_learning_rate` :45-57)."""

    def __init__(self, flat_groups, params, lr_weights, lr_biases=None, weight_decay=0.0, momentum=0.9, eta=0.001,
                 weight_decay_filter=False, lars_adaptation_filter=False):
        super().__init__(params, dict(lr=lr_weights, weight_decay=weight_decay, momentum=momentum, eta=eta))
        self.flat_groups = list(flat_groups)
        self.lr_weights, self.lr_biases = lr_weights, lr_weights if lr_biases is None else lr_biases
        self.wd_filter, self.adapt_filter = weight_decay_filter, lars_adaptation_filter
        self.grad_scale = 1.0
        self._tables = {}

    def _table(self, fg):
        import numpy as np
        key = id(fg)
        if key not in self._tables:
            seg = np.zeros(len(fg.params), dtype=[("off", "<i8"), ("n", "<i8"), ("flags", "<i4"), ("pad", "<i4")])
            for i, (p, o) in enumerate(zip(fg.params, fg.offsets)):
                is1d = p.ndim == 1
                seg[i] = (o, p.numel(), (1 if (not self.wd_filter or not is1d) else 0) |
                          (2 if (not self.adapt_filter or not is1d) else 0), 0)
            dev = fg.data.device
            self._tables[key] = (torch.from_numpy(seg.view(np.uint8)).to(dev), [p.ndim == 1 for p in fg.params],
                                 torch.empty(2 * len(fg.params), dtype=torch.float64, device=dev))
        return self._tables[key]

    def set_lr(self, lr_weights, lr_biases):
        self.lr_weights, self.lr_biases = lr_weights, lr_biases

    @torch.no_grad()
    def step(self, closure=None):
        g0 = self.param_groups[0]
        for fg in self.flat_groups:
            if fg.momentum is None:
                fg.momentum = torch.zeros_like(fg.data)
            seg, is1d, norms = self._table(fg)
            lr = torch.tensor([self.lr_biases if b else self.lr_weights for b in is1d], dtype=torch.float32, device=fg.data.device)
            N.call("lars_step", fg.data, fg.grad, fg.momentum, seg, len(is1d), lr, float(g0["weight_decay"]),
                   float(g0["momentum"]), float(g0["eta"]), float(self.grad_scale), norms)

    def zero_grad(self, set_to_none=True):
        for fg in self.flat_groups:
            for p in fg.params:
                p.grad = None


class HipLARC(FlatState, torch.optim.Optimizer):
    """apex `LARC(torch.optim.SGD(params, lr, momentum=0.9, weight_decay=wd), trust_coefficient=0.001, clip=False)` -
    the optimiser of `extras/decar-v2/main.py:92-97, 111` - on flat parameter groups: two launches per group (per-tensor
    norms, fused adaptive-rate / decay / momentum / update).  `skip` = indices of tensors without a gradient this step (the
    reference sets `p.grad = None` for the prototypes while they are frozen: neither LARC nor SGD touches them)."""

    supports_skip = True           # step(skip=...) exists (torch wraps `step`, so its signature cannot be inspected through the instance)

    def __init__(self, flat_groups, params, lr, momentum=0.9, weight_decay=0.0, trust_coefficient=0.001, eps=1e-8, clip=False):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, trust_coefficient=trust_coefficient,
                                      eps=eps, clip=clip))
        self.flat_groups = list(flat_groups)
        self.grad_scale = 1.0
        self._tables = {}
        self.steps = 0

    def _table(self, fg, skip):
        import numpy as np
        key = (id(fg), tuple(sorted(skip)))
        if key not in self._tables:
            seg = np.zeros(len(fg.params), dtype=[("off", "<i8"), ("n", "<i8"), ("flags", "<i4"), ("pad", "<i4")])
            for i, (p, o) in enumerate(zip(fg.params, fg.offsets)):
                seg[i] = (o, p.numel(), 4 if i in skip else 0, 0)
            dev = fg.data.device
            self._tables[key] = (torch.from_numpy(seg.view(np.uint8)).to(dev),
                                 torch.empty(2 * len(fg.params), dtype=torch.float64, device=dev))
        return self._tables[key]

    @torch.no_grad()
    def step(self, closure=None, skip=()):
        g0 = self.param_groups[0]
        for fg in self.flat_groups:
            if fg.momentum is None:
                fg.momentum = torch.zeros_like(fg.data)
            seg, norms = self._table(fg, set(skip))
            N.call("larc_step", fg.data, fg.grad, fg.momentum, seg, len(fg.params), float(g0["lr"]), float(g0["weight_decay"]),
                   float(g0["momentum"]), float(g0["trust_coefficient"]), float(g0["eps"]), int(bool(g0["clip"])),
                   float(self.grad_scale), norms)
        self.steps += 1

    def zero_grad(self, set_to_none=True):
        for fg in self.flat_groups:
            for p in fg.params:
                p.grad = None


# ---- learning-rate schedules of the extras trainers ------------------------------------------------------------------------
def lars_adjust_learning_rate(optimizer, step, epochs, steps_per_epoch, batch_size):
    """`adjust_learning_rate` of `extras/delores-s/multi_proc.py:45-57` (the reference forgets to import math): linear warm-up
    over 10 epochs to batch_size / 256, cosine to 0.1 % of it; weights get 0.2 x, biases 0.0048 x.  -> (lr_weights, lr_biases),
    also applied to a HipLARS optimiser."""
    import math
    max_steps = epochs * steps_per_epoch
    warmup_steps = 10 * steps_per_epoch
    base_lr = batch_size / 256
    if step < warmup_steps:
        lr = base_lr * step / warmup_steps
    else:
        s, m = step - warmup_steps, max_steps - warmup_steps
        q = 0.5 * (1 + math.cos(math.pi * s / m))
        lr = base_lr * q + base_lr * 0.001 * (1 - q)
    if optimizer is not None:
        if hasattr(optimizer, "set_lr"):
            optimizer.set_lr(lr * 0.2, lr * 0.0048)
        else:
            optimizer.param_groups[0]["lr"] = lr * 0.2
            if len(optimizer.param_groups) > 1:
                optimizer.param_groups[1]["lr"] = lr * 0.0048
    return lr * 0.2, lr * 0.0048


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """`extras/decar-v2/multi_proc.py:61-72`: per-iteration values, linear warm-up then half a cosine."""
    import numpy as np
    warmup_iters = warmup_epochs * niter_per_ep
    warmup = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 else np.array([])
    iters = np.arange(epochs * niter_per_ep - warmup_iters)
    schedule = np.concatenate((warmup, final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * iters / len(iters)))))
    assert len(schedule) == epochs * niter_per_ep
    return schedule


def dcv2_lr_schedule(base_lr, final_lr, epochs, niter_per_ep, warmup_epochs=10):
    """The schedule `extras/decar-v2/main.py:118-122` builds: linear 0 -> base_lr over 10 epochs, cosine to final_lr."""
    import numpy as np
    warm = np.linspace(0, base_lr, niter_per_ep * warmup_epochs)
    t = np.arange(niter_per_ep * (epochs - warmup_epochs))
    cos = final_lr + 0.5 * (base_lr - final_lr) * (1 + np.cos(np.pi * t / (niter_per_ep * (epochs - warmup_epochs))))
    return np.concatenate((warm, cos))


class HipAdamW(FlatState, torch.optim.Optimizer):
    """torch.optim.AdamW (`extras/mast_new/mast/moco_model.py:373-379`) as one launch per flat parameter group.  The step
    count lives in device memory and is advanced by a device op, so the optimiser step can sit inside a captured hipGraph."""
    state_buffers = ("exp_avg", "exp_avg_sq")

    def __init__(self, flat_groups, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.flat_groups = list(flat_groups)
        self.grad_scale = grad_scale
        self.grad_scale_tensor = None
        self.step_count = None
        self.steps = 0

    @torch.no_grad()
    def step(self, closure=None):
        g0 = self.param_groups[0]
        for fg in self.flat_groups:
            if getattr(fg, "exp_avg", None) is None:
                fg.exp_avg = torch.zeros_like(fg.data)
                fg.exp_avg_sq = torch.zeros_like(fg.data)
            if self.step_count is None or self.step_count.device != fg.data.device:
                self.step_count = torch.zeros(1, dtype=torch.int64, device=fg.data.device)
        self.step_count.add_(1)
        for fg in self.flat_groups:
            N.call("adamw", fg.data, fg.grad, fg.exp_avg, fg.exp_avg_sq, fg.numel, float(g0["lr"]), float(g0["betas"][0]),
                   float(g0["betas"][1]), float(g0["eps"]), float(g0["weight_decay"]), float(self.grad_scale), self.step_count)
        self.steps += 1

    def zero_grad(self, set_to_none=True):
        for fg in self.flat_groups:
            for p in fg.params:
                p.grad = None
