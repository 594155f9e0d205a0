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
    """y = x W^T + b [ReLU] on the MFMA GEMM; x in the activation dtype, W / b fp32 parameters; y fp32."""

    @staticmethod
    def forward(ctx, x, W, b, relu):
        x = x.contiguous()
        dt = _dtype_of(x)
        M, K = x.shape
        Wc = E.cast(dt, W.data)
        y = E.linear_fwd(dt, x, Wc, M, W.shape[0], K, bias=None if b is None else b.data, relu=int(relu), out_f32=1)
        ctx.save_for_backward(x, Wc, y if relu else None)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, Wc, y = ctx.saved_tensors
        dt = _dtype_of(x)
        M, K = x.shape
        Nout = Wc.shape[0]
        if y is not None:                           # ReLU gate on the stored output
            gated = torch.empty_like(y)
            N.call("relu_bwd", N.F32, gy.float().contiguous(), y, gated, y.numel())
            gy = gated
        g32 = gy.float().contiguous()
        g = g32.to(x.dtype)
        Np = (Nout + 63) // 64 * 64                 # the kernels vectorise by 8 (GEMM) / 64 (column sums): pad the class axis
        if Np != Nout:
            g = torch.nn.functional.pad(g, (0, Np - Nout))
            g32 = torch.nn.functional.pad(g32, (0, Np - Nout))
            Wc = torch.nn.functional.pad(Wc, (0, 0, 0, Np - Nout))
        dW = torch.zeros(Np, K, dtype=torch.float32, device=x.device)
        E.linear_bwd_w(dt, g, x, dW, M, Np, K)
        db = None
        if ctx.has_bias:
            db = torch.zeros(Np, dtype=torch.float32, device=x.device)
            E.colsum_add(N.F32, g32, M, Np, db)       # bias gradient from the unrounded fp32 gradient
            db = db[:Nout]
        dx = E.linear_bwd_x(dt, g, Wc, M, Np, K)
        return dx, dW[:Nout], db, None


class SoftmaxRowsFn(torch.autograd.Function):
    """nn.Softmax(dim=1) on fp32 [M, C] (`src/upstream/slicer/upstream_encoder.py:19`)."""

    @staticmethod
    def forward(ctx, x):
        x = x.float().contiguous()
        y = torch.empty_like(x)
        N.call("softmax_rows_fwd", x, y, x.shape[0], x.shape[1])
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, = ctx.saved_tensors
        gx = torch.empty_like(y)
        N.call("softmax_rows_bwd", y, gy.float().contiguous(), gx, y.shape[0], y.shape[1])
        return gx


class MocoCEFn(torch.autograd.Function):
    """InfoNCE of MoCo against the queue: normalise q and k, logits [B, 1+K] / T, CE(label 0)
    (`src/upstream/slicer/upstream_expert.py:196-218, 229-230`).  Forward also computes dq (one fused launch sequence);
    returns (loss, normalised keys fp32 for the enqueue)."""

    @staticmethod
    def forward(ctx, q, k, queue, temperature, dtype):
        loss = torch.zeros(1, dtype=torch.float32, device=q.device)
        shadow = E.cast(dtype, queue) if dtype != N.F32 else queue
        dq, kn32 = E.moco_forward_backward(dtype, q.float().contiguous(), k.float().contiguous(), queue, shadow,
                                           float(temperature), loss, backward=True)
        ctx.save_for_backward(dq)
        ctx.mark_non_differentiable(kn32)
        return loss[0].clone(), kn32

    @staticmethod
    def backward(ctx, g, _):
        dq, = ctx.saved_tensors
        return dq.float() * g, None, None, None, None


class BatchNorm1dFn(torch.autograd.Function):
    """nn.BatchNorm1d (+ optional ReLU) on a fp32 pre-activation [M, C]: batch statistics in fp64 sums (`colstats` +
    `bn_finalize`, which also updates the running buffers), normalise + ReLU into the activation dtype (`colbn_fwd`);
    backward is the two-pass `colbn_bwd`.  Eval mode folds the running statistics into an affine map."""

    @staticmethod
    def forward(ctx, a, weight, bias, running_mean, running_var, dtype, relu, training):
        a = a.float().contiguous()
        M, C = a.shape
        if training:
            st = E._bn_train(N.F32, a, M, C, weight.data, bias.data, running_mean, running_var, True)
        else:
            st = E._bn_eval(a, C, weight.data, bias.data, running_mean, running_var) + (None, None)
        h = torch.empty(M, C, dtype=N.torch_dtype(dtype), device=a.device)
        N.call("colbn_fwd", dtype, N.F32, a, st[0], st[1], int(relu), h, 1, M, C)
        ctx.save_for_backward(a, *[t for t in st if t is not None])
        ctx.meta = (dtype, int(relu), training)
        return h

    @staticmethod
    def backward(ctx, dh):
        dtype, relu, training = ctx.meta
        if not training:
            raise RuntimeError("BatchNorm1dFn: backward through eval-mode statistics is not part of the HIP path")
        a, scale, shift, mean, rstd = ctx.saved_tensors
        M, C = a.shape
        tmp = torch.empty(2 * C, dtype=torch.float64, device=a.device)
        da = torch.empty(M, C, dtype=N.torch_dtype(dtype), device=a.device)
        dgamma = torch.zeros(C, dtype=torch.float32, device=a.device)
        dbeta = torch.zeros(C, dtype=torch.float32, device=a.device)
        g = dh.float().contiguous()
        if E.SYNC_BN is None:
            N.call("colbn_bwd", dtype, N.F32, N.F32, a, g, scale, shift, mean, rstd, relu, 1, M, C, tmp, da, dgamma, dbeta)
        else:                                    # SyncBatchNorm: the two backward means are over the global batch
            N.call("colbn_bwd_stats", dtype, N.F32, N.F32, a, g, scale, shift, mean, rstd, relu, 1, M, C, tmp)
            tg = E.SYNC_BN.all_reduce(tmp.clone())
            N.call("colbn_bwd_apply", dtype, N.F32, N.F32, a, g, scale, shift, mean, rstd, relu, 1, M, C, tg, float(M * E.SYNC_BN.world), da)
            N.call("add_d2f", tmp[:C], dbeta, C)
            N.call("add_d2f", tmp[C:], dgamma, C)
        return da.float(), dgamma, dbeta, None, None, None, None, None
