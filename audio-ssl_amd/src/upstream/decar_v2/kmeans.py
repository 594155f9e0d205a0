"""DeepCluster-v2 on MI355X: distributed spherical k-means over the memory bank and the prototype cross-entropy
(`extras/decar-v2/utils.py:276-346` cluster_memory, `extras/decar-v2/main.py:228-233`).

E step = MFMA GEMM (memory x centroids^T) + `row_argmax`; M step = `kmeans_accumulate` (fp32 atomics on the GPU - the
reference does it on the CPU with scipy csr_matrix) + all-reduce of counts / sums over ranks (RCCL) + `kmeans_update`
(mean, L2 normalise, empty clusters keep their centroid).  Seeding and collectives follow the reference: centroids
are `randperm` rows of rank 0's memory, broadcast; assignments / indexes are all-gathered."""
import torch

from src import _native as N
from src import engine as E


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 else None


@torch.no_grad()
def spherical_kmeans(mem, K, n_iters=10, centroids=None, precision=N.F32):
    """mem [N_loc, D] fp32 (device, L2-normalised rows) -> (centroids [K, D] fp32, local assignments int64 [N_loc])."""
    Nloc, D = mem.shape
    dist = _dist()
    if centroids is None:
        centroids = torch.empty(K, D, dtype=torch.float32, device=mem.device)
        if dist is None or dist.get_rank() == 0:
            idx = torch.randperm(Nloc)[:K].to(mem.device)
            assert len(idx) >= K, "please reduce the number of centroids"
            centroids = mem[idx].clone()
        if dist is not None:
            dist.broadcast(centroids, 0)
    centroids = centroids.contiguous().clone()
    dot = torch.empty(Nloc, K, dtype=torch.float32, device=mem.device)
    assign = torch.empty(Nloc, dtype=torch.int64, device=mem.device)
    sums = torch.empty(K, D, dtype=torch.float32, device=mem.device)
    counts = torch.empty(K, dtype=torch.int32, device=mem.device)
    a = mem if precision == N.F32 else E.cast(precision, mem)
    for it in range(n_iters + 1):
        b = centroids if precision == N.F32 else E.cast(precision, centroids)
        E.gemm(precision, 0, 0, Nloc, K, D, a, D, b, D, dot, K, out_f32=1)          # E step
        N.call("row_argmax", dot, Nloc, K, assign)
        if it == n_iters:
            break
        N.call("kmeans_accumulate", mem, assign, Nloc, K, D, sums, counts)         # M step
        if dist is not None:
            dist.all_reduce(counts)
            dist.all_reduce(sums)
        N.call("kmeans_update", sums, counts, K, D, centroids)
    return centroids, assign


@torch.no_grad()
def cluster_memory(mem, index, size_dataset, K, n_iters=10, prototypes_weight=None, precision=N.F32):
    """One head of the reference's cluster_memory: k-means, copy the centroids into the prototype layer, scatter the
    (all-gathered) assignments to dataset order.  Returns int64 [size_dataset], -100 where unseen."""
    centroids, assign = spherical_kmeans(mem, K, n_iters, precision=precision)
    if prototypes_weight is not None:
        prototypes_weight.copy_(centroids)
    dist = _dist()
    if dist is not None:
        w = dist.get_world_size()
        a_all = torch.empty(w * assign.numel(), dtype=assign.dtype, device=assign.device)
        i_all = torch.empty(w * index.numel(), dtype=index.dtype, device=index.device)
        dist.all_gather_into_tensor(a_all, assign)
        dist.all_gather_into_tensor(i_all, index.contiguous())
    else:
        a_all, i_all = assign, index
    out = torch.full((size_dataset,), -100, dtype=torch.int64, device=assign.device)
    out[i_all] = a_all
    return out, centroids


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        B, K = logits.shape
        lg = logits.float().contiguous()
        loss = torch.zeros(1, dtype=torch.float32, device=lg.device)
        cnt = torch.empty(1, dtype=torch.int32, device=lg.device)
        dl = torch.empty(B, K, dtype=torch.float32, device=lg.device)
        N.call("ce_rows", N.F32, lg, target.contiguous(), B, K, ignore_index, cnt, loss, dl)
        ctx.save_for_backward(dl)
        ctx.in_dtype = logits.dtype
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return (dl * g).to(ctx.in_dtype), None, None


def prototype_cross_entropy(scores, targets, ignore_index=-100):
    """nn.CrossEntropyLoss(ignore_index=-100)(scores, targets) of `extras/decar-v2/main.py:205, 228-233`."""
    return _CEFn.apply(scores, targets, ignore_index)
