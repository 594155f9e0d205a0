"""Flat parameter storage: every trainable tensor of a group is a view into ONE fp32 buffer.

MI355X-first layout choice: the optimiser is a single elementwise launch over the flat buffer, the MoCo EMA of
the key encoder is one launch, and data-parallel gradient reduction is ONE large RCCL all-reduce per group
(xGMI rings are per-link bound: few large collectives beat many small ones).  state_dict keys / shapes are
unchanged - `load_state_dict` copies into the views.
"""
import torch


def _align(n, a=64):
    return (n + a - 1) // a * a


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
        for p, o in zip(self.params, self.offsets):
            v = self.data[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v

    def refresh_shadow(self, dtype):
        """One cast launch over the whole buffer: bf16 copies of every weight for the MFMA GEMMs of this step."""
        from src import _native as N
        if dtype == N.F32:
            self._shadow = None
            return
        if self._shadow is None or self._shadow.device != self.data.device:
            self._shadow = torch.empty(self.numel, dtype=torch.bfloat16, device=self.data.device)
        N.call("cast", dtype, self.data, self._shadow, self.numel)

    def shadow_dict(self, prefix=""):
        """name (prefix stripped) -> weight in the activation dtype (the fp32 parameter itself on the fp32 path)."""
        src = self.data if self._shadow is None else self._shadow
        out = {}
        for n, p, o in zip(self.names, self.params, self.offsets):
            if n.startswith(prefix):
                out[n[len(prefix):]] = src[o:o + p.numel()].view(p.shape)
        return out

    def grad_view(self, i):
        p, o = self.params[i], self.offsets[i]
        return self.grad[o:o + p.numel()].view(p.shape)

    def grad_dict(self, prefix=""):
        """name (with `prefix` stripped) -> fp32 view into the flat gradient."""
        out = {}
        for i, n in enumerate(self.names):
            if n.startswith(prefix):
                out[n[len(prefix):]] = self.grad_view(i)
        return out

    def param_dict(self, prefix=""):
        return {n[len(prefix):]: p.data for n, p in zip(self.names, self.params) if n.startswith(prefix)}

    def zero_grad(self):
        self.grad.zero_()

    def attach_grads(self, scale=None):
        """Expose the flat gradient through p.grad (alias when p.grad is None, else accumulate)."""
        if scale is not None:
            self.grad.mul_(scale)
        for i, p in enumerate(self.params):
            if not p.requires_grad:
                continue
            g = self.grad_view(i)
            if p.grad is None:
                p.grad = g
            elif p.grad.data_ptr() != g.data_ptr():
                p.grad.add_(g)

    def to(self, device):
        if self.data.device == torch.device(device):
            return self
        new = self.data.to(device)
        self.data = new
        self.grad = torch.zeros_like(new)
        self.momentum = None
        for p, o in zip(self.params, self.offsets):
            p.data = new[o:o + p.numel()].view(p.shape)
            p.grad = None
        return self
