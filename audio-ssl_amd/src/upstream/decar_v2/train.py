"""DeepCluster-v2 training loop pieces on MI355X - `extras/decar-v2/main.py:198-292` (train) and
`extras/decar-v2/utils.py:244-272` (init_memory) of the reference.

One epoch = distributed spherical k-means over the memory bank (`kmeans.cluster_memory`, which also writes the
centroids into the prototype layer) followed by the supervised-by-assignment steps: embeddings of view 1 feed the
memory bank, prototype scores of view 2 are trained against the clip's cluster with CrossEntropy(ignore_index=-100);
prototype gradients are dropped while `it < freeze_prototypes_niters` (always, with the shipped default 1e10).
Parameters live in one flat buffer (`src.flat.FlatGroup`), so the optimiser step is one launch (`HipSGD` / `HipLARS`;
the reference wraps SGD in apex's LARC, which is not importable here - HipLARS is the in-tree layer-wise alternative).
"""
import torch

from src.flat import FlatGroup
from src.upstream.decar_v2.kmeans import cluster_memory, prototype_cross_entropy


class DeepClusterState:
    """Flat parameter storage + optimiser + memory banks of one rank."""

    def __init__(self, model, optimizer_cls, feat_dim, size_memory, n_crops_for_assign=1, **opt_kwargs):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.model = model
        self.flat = FlatGroup(named)
        self.optimizer = optimizer_cls([self.flat], [p for _, p in named], **opt_kwargs)
        dev = self.flat.data.device
        self.local_memory_index = torch.zeros(size_memory, dtype=torch.long, device=dev)
        self.local_memory_embeddings = torch.zeros(n_crops_for_assign, size_memory, int(feat_dim), device=dev)
        self.iteration = 0


@torch.no_grad()
def init_memory(state, dataloader):
    """Fill the memory bank with the embeddings of view 1 (`utils.py:244-272`); dataloader yields (index, [view1, view2])."""
    start = 0
    for index, inputs in dataloader:
        n = inputs[1].size(0)
        inp = [x.cuda(non_blocking=True) for x in inputs]
        emb = state.model(inp)[0]
        state.local_memory_index[start:start + n] = index.to(emb.device)
        state.local_memory_embeddings[0][start:start + n] = emb.float()
        start += n
    return state.local_memory_index, state.local_memory_embeddings


def cluster_epoch(state, size_dataset, nmb_prototypes=(1024,), n_iters=10, crops_for_assign=(0,)):
    """`cluster_memory` of the reference for every prototype head -> assignments [heads, size_dataset] (int64, -100 unseen)."""
    out, j = [], 0
    for h, K in enumerate(nmb_prototypes):
        w = getattr(state.model.prototypes, f"prototypes{h}").weight.data
        a, _ = cluster_memory(state.local_memory_embeddings[j].contiguous(), state.local_memory_index, size_dataset, K, n_iters,
                              prototypes_weight=w)
        out.append(a)
        j = (j + 1) % len(crops_for_assign)
    return torch.stack(out)


def train_step(state, idx, inputs, assignments, start_idx, nmb_crops=(1,), crops_for_assign=(0,),
               freeze_prototypes_niters=1e10):
    """One iteration of `train` (`main.py:216-247`): forward both views, prototype CE, backward, optimiser step, memory
    bank update.  Returns (loss tensor, new start_idx)."""
    model, flat = state.model, state.flat
    flat.zero_grad()
    for p in flat.params:
        p.grad = None
    flat.attach_grads()                               # p.grad = views of the flat gradient: autograd accumulates in place
    emb, output = model(inputs)
    emb = emb.detach()
    bs = inputs[1].size(0)
    loss = 0
    for h in range(len(output)):
        targets = assignments[h][idx].repeat(sum(nmb_crops)).to(output[h].device)
        loss = loss + prototype_cross_entropy(output[h], targets)           # the reference divides the scores by 1.0
    loss = loss / len(output)
    loss.backward()
    # data parallel (`nn.parallel.DistributedDataParallel`, main.py:84): ONE all-reduce of the flat gradient over RCCL; the
    # 1/world of DDP's averaging is folded into the optimiser launch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat.grad)
        state.optimizer.grad_scale = 1.0 / dist.get_world_size()
    # "cancel some gradients": the reference sets p.grad = None for the prototypes, so neither LARC nor SGD touches them
    skip = [i for i, n in enumerate(flat.names) if "prototypes" in n] if state.iteration < freeze_prototypes_niters else []
    if getattr(state.optimizer, "supports_skip", False):         # HipLARC: the kernel leaves flagged tensors and their momentum alone
        state.optimizer.step(skip=skip)
    else:
        frozen = [(i, flat.params[i].data.clone()) for i in skip]
        state.optimizer.step()
        for i, saved in frozen:
            p, o = flat.params[i], flat.offsets[i]
            p.data.copy_(saved)
            if flat.momentum is not None:
                flat.momentum[o:o + p.numel()].zero_()
    state.local_memory_index[start_idx:start_idx + bs] = idx.to(emb.device)
    for i, crop_idx in enumerate(crops_for_assign):
        state.local_memory_embeddings[i][start_idx:start_idx + bs] = emb[crop_idx * bs:(crop_idx + 1) * bs].float()
    state.iteration += 1
    return loss.detach(), start_idx + bs
