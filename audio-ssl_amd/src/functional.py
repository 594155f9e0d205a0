"""torch.autograd bridges for the small module-level ops (compat path; the training hot path is the fused step of
the experts, which calls `src.engine` directly)."""
import torch

from src import _native as N
from src import engine as E


def _dtype_of(t):
    return N.F32 if t.dtype == torch.float32 else N.BF16


class MaxMeanFn(torch.autograd.Function):
    """x [N,T,d] (post-ReLU) -> max_T + mean_T  (`upstream_encoder.py:26-28`)."""

    @staticmethod
    def forward(ctx, h):
        h = h.contiguous()
        dt = _dtype_of(h)
        y, arg = E.maxmean_forward(dt, h)
        ctx.save_for_backward(h, arg)
        return y

    @staticmethod
    def backward(ctx, gy):
        h, arg = ctx.saved_tensors
        dt = _dtype_of(h)
        Nimg, T, d = h.shape
        gh = torch.empty_like(h)
        # gate-free variant: the ReLU gate belongs to the encoder's own backward, so route through an all-ones gate
        ones = torch.ones_like(h)
        N.call("maxmean_bwd", dt, N.F32, gy.float().contiguous(), arg, ones, gh.view(Nimg * T, d), Nimg, T, d)
        return gh


class LinearFn(torch.autograd.Function):
    """y = x W^T + b on the MFMA GEMM; x in the activation dtype, W / b fp32 parameters; y fp32."""

    @staticmethod
    def forward(ctx, x, W, b):
        x = x.contiguous()
        dt = _dtype_of(x)
        M, K = x.shape
        Wc = E.cast(dt, W.data)
        y = E.linear_fwd(dt, x, Wc, M, W.shape[0], K, bias=None if b is None else b.data, out_f32=1)
        ctx.save_for_backward(x, Wc)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, Wc = ctx.saved_tensors
        dt = _dtype_of(x)
        M, K = x.shape
        Nout = Wc.shape[0]
        g = gy.to(x.dtype).contiguous()
        Np = (Nout + 63) // 64 * 64                 # the kernels vectorise by 8 (GEMM) / 64 (column sums): pad the class axis
        if Np != Nout:
            g = torch.nn.functional.pad(g, (0, Np - Nout))
            Wc = torch.nn.functional.pad(Wc, (0, 0, 0, Np - Nout))
        dW = torch.zeros(Np, K, dtype=torch.float32, device=x.device)
        E.linear_bwd_w(dt, g, x, dW, M, Np, K)
        db = None
        if ctx.has_bias:
            db = torch.zeros(Np, dtype=torch.float32, device=x.device)
            E.colsum_add(dt, g, M, Np, db)
            db = db[:Nout]
        dx = E.linear_bwd_x(dt, g, Wc, M, Np, K)
        return dx, dW[:Nout], db
