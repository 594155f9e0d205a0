"""CPU, world_size 2 over gloo: the collective logic of the data-parallel path."""
import copy
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import CFG_M, CFG_S


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _gather_fn(rank, world):
    from src.utils import concat_all_gather
    x = torch.arange(6, dtype=torch.float32).view(3, 2) + 100 * rank
    return concat_all_gather(x).numpy()


def test_concat_all_gather_orders_ranks():
    a, b = _run(_gather_fn)
    want = np.concatenate([np.arange(6).reshape(3, 2), np.arange(6).reshape(3, 2) + 100]).astype(np.float32)
    assert np.array_equal(a, want) and np.array_equal(b, want)


def _segmented_reduce_fn(rank, world):
    """The overlapped two-segment gradient all-reduce of FusedExpertMixin (heads first, encoder second)."""
    from src.upstream.common import FusedExpertMixin

    class Tiny(FusedExpertMixin, torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = torch.nn.Linear(8, 4)
            self.p = torch.nn.Linear(4, 4, bias=False)
    t = Tiny()
    t.ensure_flat()
    ho = t.head_offset()
    assert 0 < ho < t.flat.numel and t.flat.names[-1] == "p.weight"
    t.flat.grad.fill_(float(rank + 1))
    t.reduce_begin("heads")                 # async, before the "encoder backward"
    t.flat.grad[:ho].add_(10.0)             # encoder gradients arrive later
    t.all_reduce_grads()                    # launches the encoder segment, joins both
    return t.flat.grad.clone().numpy(), ho


def test_segmented_overlapped_gradient_allreduce():
    (ga, ho), (gb, _) = _run(_segmented_reduce_fn)
    assert np.array_equal(ga, gb)
    assert np.all(ga[ho:] == 1.5) and np.all(ga[:ho] == 11.5)      # summed over 2 ranks, then averaged (no HipSGD attached)


def _async_shuffle_fn(rank, world):
    from src.encoder import AudioNTT2020Task6
    from src.upstream.delores_m.upstream_expert import Upstream_Expert
    ex = Upstream_Expert(copy.deepcopy(CFG_M), base_encoder=AudioNTT2020Task6, num_negatives=64)
    x = torch.arange(4, dtype=torch.float32).view(4, 1) + 10 * rank
    pend = ex._shuffle_begin(x)
    xs, idx_un = ex._shuffle_end(pend, 4)
    return x.numpy(), ex._batch_unshuffle_ddp(xs * 2.0, idx_un).numpy()


def test_async_key_gather_roundtrip():
    for x, back in _run(_async_shuffle_fn):
        assert np.array_equal(back, 2.0 * x)


def _shuffle_fn(rank, world):
    import sys
    from src.encoder import AudioNTT2020Task6
    from src.upstream.delores_m.upstream_expert import Upstream_Expert
    torch.manual_seed(7)
    ex = Upstream_Expert(copy.deepcopy(CFG_M), base_encoder=AudioNTT2020Task6, num_negatives=64)
    x = torch.arange(4, dtype=torch.float32).view(4, 1) + 10 * rank             # local batch of 4 "images"
    xs, idx_un = ex._batch_shuffle_ddp(x)
    back = ex._batch_unshuffle_ddp(xs * 2.0, idx_un)                              # "encoder" = times two
    return x.numpy(), xs.numpy(), back.numpy()


def test_moco_batch_shuffle_roundtrip():
    res = _run(_shuffle_fn)
    allx = np.concatenate([r[0] for r in res])
    shuf = np.concatenate([r[1] for r in res])
    assert sorted(shuf.ravel().tolist()) == sorted(allx.ravel().tolist())        # a permutation of the global batch
    for x, _, back in res:
        assert np.array_equal(back, 2.0 * x)                                      # every rank gets its own keys back


def _barlow_fn(rank, world):
    """Cross-GPU Barlow (extras/delores-s/models_byol.py:108-112): c summed over ranks, gradient flows as identity."""
    from oracle import fill, model as OM
    p = OM.Projection(512, 5e-5)
    fill.fill_state_dict_(p, seed=1)
    p.train()
    B = 8
    y1 = torch.from_numpy(fill.uniform((2 * B, 512), 1, 0, 2))[rank * B:(rank + 1) * B].clone().requires_grad_()
    y2 = torch.from_numpy(fill.uniform((2 * B, 512), 2, 0, 2))[rank * B:(rank + 1) * B].clone().requires_grad_()

    def ar(c):
        c = c.clone()
        dist.all_reduce(c)
        return c
    # divisor must be the GLOBAL batch before the reduce
    z1, z2 = p.projector(y1), p.projector(y2)
    c_local = (p.bn(z1).T @ p.bn(z2)) / (B * world)
    c = c_local + (ar(c_local.detach()) - c_local.detach())
    loss = p.loss_from_c(c)
    loss.backward()
    g = torch.cat([q.grad.flatten() for q in p.parameters()])
    dist.all_reduce(g)                                                           # DDP gradient sum (C1)
    return float(loss), c.detach().numpy(), g.numpy()


def test_cross_gpu_barlow_semantics():
    (la, ca, ga), (lb, cb, gb) = _run(_barlow_fn)
    assert la == lb and np.array_equal(ca, cb) and np.allclose(ga, gb)
    assert ca.shape == (2048, 2048) and abs(np.trace(ca)) <= 2048 + 1e-3


def _flat_allreduce_fn(rank, world):
    from src.flat import FlatGroup
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 4)
    fg = FlatGroup(list(lin.named_parameters()))
    fg.grad.fill_(float(rank + 1))
    dist.all_reduce(fg.grad)
    fg.attach_grads()
    return lin.weight.grad.clone().numpy(), fg.numel


def test_flat_gradient_allreduce_is_one_collective():
    (ga, n), (gb, _) = _run(_flat_allreduce_fn)
    assert np.all(ga == 3.0) and np.all(gb == 3.0) and n % 64 == 0


def _kmeans_fn(rank, world):
    """DeepCluster-v2 M step sharded over ranks == single-process k-means on the concatenated memory (C7-C8)."""
    from oracle import fill, kmeans as OK
    mem = torch.nn.functional.normalize(torch.from_numpy(fill.normalish((256, 32), 5)).abs(), dim=1)
    init = mem[:8].clone()
    dist.broadcast(init, 0)

    def ar(t):
        t = t.clone()
        dist.all_reduce(t)
        return t
    c, a = OK.cluster_memory(mem[rank * 128:(rank + 1) * 128], init, n_iters=4, all_reduce=ar)
    c_full, a_full = OK.cluster_memory(mem, init, n_iters=4)
    return bool(torch.allclose(c, c_full, atol=1e-6)), bool(torch.equal(a, a_full[rank * 128:(rank + 1) * 128]))


def test_distributed_kmeans_equals_single_process():
    assert _run(_kmeans_fn) == [(True, True), (True, True)]
