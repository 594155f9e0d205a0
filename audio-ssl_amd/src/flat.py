"""Flat parameter storage: every trainable tensor of a group is a view into ONE fp32 buffer.

MI355X-first layout choice: the optimiser is a single elementwise launch over the flat buffer, the MoCo EMA of
the key encoder is one launch, and data-parallel gradient reduction is ONE large RCCL all-reduce per group
(xGMI rings are per-link bound: few large collectives beat many small ones).  state_dict keys / shapes are
unchanged - `load_state_dict` copies into the views.
"""
import torch


def _align(n, a=64):
    return (n + a - 1) // a * a


def cached_param_dict(mod):
    """name -> fp32 tensor for every parameter and buffer of `mod`.  The step asks for this several times per
    iteration; the dictionary is rebuilt only when the storage moved (`.to()`, re-flattening)."""
    p0 = next(mod.parameters())
    c = mod.__dict__.get("_pd_cache")
    if c is None or c[0] != p0.data_ptr():
        P = {n: p.data for n, p in mod.named_parameters()}
        P.update({n: b for n, b in mod.named_buffers()})
        c = mod.__dict__["_pd_cache"] = (p0.data_ptr(), P)
    return c[1]


class FlatGroup:
    def __init__(self, named_params, device=None):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        device = device or self.params[0].device
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += _align(p.numel())           # 256-byte aligned slices keep every view 16-byte aligned
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.momentum = None
        self._shadow = None
        self._fresh_shadow = False             # the optimiser's fused tail (HipSGD.fused_refresh) left the bf16 shadow current
        self._fresh_grad = False               # ... and the gradient zero; each flag buys ONE skipped sweep
        self._early_fused = False
        self._views = {}                       # cached view dictionaries (the step asks for the same ones every time)
        for p, o in zip(self.params, self.offsets):
            v = self.data[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v

    def intact(self):
        """True while the parameters still alias this buffer (a `.to()` / re-assignment of p.data breaks it)."""
        p0, pl = self.params[0], self.params[-1]
        base = self.data.data_ptr()
        return p0.data_ptr() == base and pl.data_ptr() == base + 4 * self.offsets[-1] and p0.device == self.data.device

    def refresh_shadow(self, dtype):
        """One cast launch over the whole buffer: bf16 copies of every weight for the MFMA GEMMs of this step."""
        from src import _native as N
        if dtype == N.F32:
            self._shadow = None
            return
        if self._shadow is None or self._shadow.device != self.data.device:
            self._shadow = torch.empty(self.numel, dtype=torch.bfloat16, device=self.data.device)
            self._fresh_shadow = False
        if self._fresh_shadow:                 # written by the SGD pass of the previous step
            self._fresh_shadow = False
            return
        N.call("cast", dtype, self.data, self._shadow, self.numel)

    def fresh_shadow_buffer(self, dtype):
        """The bf16 shadow buffer for a kernel that is about to fill it itself (EMA pass); None on the fp32 path.  The next
        refresh_shadow() is then a no-op."""
        from src import _native as N
        if dtype == N.F32:
            return None
        if self._shadow is None or self._shadow.device != self.data.device:
            self._shadow = torch.empty(self.numel, dtype=torch.bfloat16, device=self.data.device)
        self._fresh_shadow = True
        return self._shadow

    def shadow_dict(self, prefix=""):
        """name (prefix stripped) -> weight in the activation dtype (the fp32 parameter itself on the fp32 path)."""
        src = self.data if self._shadow is None else self._shadow
        key = ("w", prefix, src.data_ptr())
        out = self._views.get(key)
        if out is None:
            out = {}
            for n, p, o in zip(self.names, self.params, self.offsets):
                if n.startswith(prefix):
                    out[n[len(prefix):]] = src[o:o + p.numel()].view(p.shape)
            self._views[key] = out
        return out

    def grad_view(self, i):
        p, o = self.params[i], self.offsets[i]
        return self.grad[o:o + p.numel()].view(p.shape)

    def grad_views(self):
        key = ("g", self.grad.data_ptr())
        out = self._views.get(key)
        if out is None:
            out = self._views[key] = [self.grad_view(i) for i in range(len(self.params))]
        return out

    def grad_dict(self, prefix=""):
        """name (with `prefix` stripped) -> fp32 view into the flat gradient."""
        key = ("gd", prefix, self.grad.data_ptr())
        out = self._views.get(key)
        if out is None:
            gv = self.grad_views()
            out = self._views[key] = {n[len(prefix):]: gv[i] for i, n in enumerate(self.names) if n.startswith(prefix)}
        return out

    def param_dict(self, prefix=""):
        return {n[len(prefix):]: p.data for n, p in zip(self.names, self.params) if n.startswith(prefix)}

    def zero_grad(self, stores_ok=False):
        """stores_ok: the caller's large gradients are STORED by their single writer this step (not accumulated into), so a
        gradient the previous step's SGD pass cleared only partially (`mark_fresh(partial=True)`: everything but those tensors)
        is good enough; any other caller gets a full clear."""
        fresh, self._fresh_grad = self._fresh_grad, False
        if fresh is True or (fresh == "partial" and stores_ok):   # cleared by the SGD pass of the previous step
            return
        self.grad.zero_()

    def mark_fresh(self, partial=False):
        self._fresh_shadow = True
        self._fresh_grad = "partial" if partial else True

    def attach_grads(self, scale=None):
        """Expose the flat gradient through p.grad (alias when p.grad is None, else accumulate)."""
        if scale is not None:
            self.grad.mul_(scale)
        for p, g in zip(self.params, self.grad_views()):
            if not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = g
            elif p.grad is not g and p.grad.data_ptr() != g.data_ptr():
                p.grad.add_(g)

    def to(self, device):
        if self.data.device == torch.device(device):
            return self
        new = self.data.to(device)
        self.data = new
        self.grad = torch.zeros_like(new)
        self.momentum = None
        self._shadow = None
        self._views = {}
        for p, o in zip(self.params, self.offsets):
            p.data = new[o:o + p.numel()].view(p.shape)
            p.grad = None
        return self
