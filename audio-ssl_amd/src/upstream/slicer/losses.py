"""SLICER's contrastive heads on MI355X: NT-Xent `InstanceLoss` and `ClusterLoss`
(`extras/slicer/contrastive_loss.py:6-92`; the `src/upstream/slicer` plugin of the reference imports a `ClusterLoss`
that does not exist in `src.utils`, SURVEY 2.4 - the maths is taken from `extras/`).

Same constructors and `forward(z_i, z_j)` as the reference.  sim = z z^T / tau runs on the MFMA GEMM, the masked
log-sum-exp and its gradient in `ntxent_fwd/bwd`; dz = (dsim + dsim^T) z / tau is two more GEMMs.  With a process
group, the negatives of all ranks are gathered (all_gather_into_tensor) and the backward reduces dz (the
differentiable-gather pattern of `extras/mast_new/mast/utils.py:220-246`)."""
import torch
import torch.nn as nn

from src import _native as N
from src import engine as E


def _world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class _NTXentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z_i, z_j, temperature, gather):
        import torch.distributed as dist
        B_loc, D = z_i.shape
        world = _world() if gather else 1
        dt = N.F32 if z_i.dtype == torch.float32 else N.BF16
        zi, zj = z_i.contiguous(), z_j.contiguous()
        if world > 1:
            gi = torch.empty(world * B_loc, D, dtype=zi.dtype, device=zi.device)
            gj = torch.empty_like(gi)
            dist.all_gather_into_tensor(gi, zi)
            dist.all_gather_into_tensor(gj, zj)
            zi, zj = gi, gj
        B = zi.shape[0]
        Nn = 2 * B
        z = torch.cat([zi, zj]).contiguous()
        sim = torch.empty(Nn, Nn, dtype=torch.float32, device=z.device)
        E.gemm(dt, 0, 0, Nn, Nn, D, z, D, z, D, sim, Nn, alpha=1.0 / temperature, out_f32=1)
        lse = torch.empty(Nn, dtype=torch.float32, device=z.device)
        loss = torch.zeros(1, dtype=torch.float32, device=z.device)
        N.call("ntxent_fwd", sim, Nn, B, lse, loss)
        ctx.save_for_backward(z, sim, lse)
        ctx.meta = (dt, B, B_loc, D, temperature, world)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        z, sim, lse = ctx.saved_tensors
        dt, B, B_loc, D, tau, world = ctx.meta
        Nn = 2 * B
        td = N.torch_dtype(dt)
        dsim = torch.empty(Nn, Nn, dtype=td, device=z.device)
        N.call("ntxent_bwd", dt, sim, lse, Nn, B, 1.0 / (Nn * tau), dsim)
        dz = torch.zeros(Nn, D, dtype=torch.float32, device=z.device)
        # dz = dsim z + dsim^T z   (two atomically accumulated GEMMs into the fp32 result)
        E.gemm(dt, 0, 1, Nn, D, Nn, dsim, Nn, z, D, dz, D, out_f32=1, atomic=1, ksplit=E._ksplit(Nn, D, Nn, 256))
        E.gemm(dt, 1, 1, Nn, D, Nn, dsim, Nn, z, D, dz, D, out_f32=1, atomic=1, ksplit=E._ksplit(Nn, D, Nn, 256))
        dz = dz * g
        if world > 1:
            dist.all_reduce(dz)
            r = dist.get_rank()
            dzi, dzj = dz[r * B_loc:(r + 1) * B_loc], dz[B + r * B_loc:B + (r + 1) * B_loc]
        else:
            dzi, dzj = dz[:B], dz[B:]
        return dzi.to(z.dtype), dzj.to(z.dtype), None, None


class InstanceLoss(nn.Module):
    def __init__(self, batch_size, temperature, device=None, gather=False):
        super().__init__()
        self.batch_size, self.temperature, self.device, self.gather = batch_size, temperature, device, gather

    def forward(self, z_i, z_j):
        return _NTXentFn.apply(z_i, z_j, float(self.temperature), self.gather)


class ClusterLoss(nn.Module):
    """NT-Xent between the columns (clusters) of two soft-assignment matrices, cosine similarity.  The entropy term
    of the reference is computed there but not returned (`contrastive_loss.py:92`); it is omitted here."""

    def __init__(self, class_num, temperature, device=None):
        super().__init__()
        self.class_num, self.temperature, self.device = class_num, temperature, device

    def forward(self, c_i, c_j):
        ci = torch.nn.functional.normalize(c_i.t().float(), dim=1)          # tiny [K, B] host-side glue (K = 128)
        cj = torch.nn.functional.normalize(c_j.t().float(), dim=1)
        pad = (-ci.shape[1]) % 8                                            # GEMM K must be a multiple of 8
        if pad:
            ci = torch.nn.functional.pad(ci, (0, pad))
            cj = torch.nn.functional.pad(cj, (0, pad))
        return _NTXentFn.apply(ci, cj, float(self.temperature), False)
