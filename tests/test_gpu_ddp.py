"""GPU: the data-parallel training step with world_size 2.  Both ranks share the box's single GPU and talk over gloo
(RCCL refuses two ranks on one device); every kernel, stream fork and collective call site of the multi-GPU path runs
exactly as it does under `torch.distributed.run` on 8 GPUs - only the transport differs."""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import CFG_M, CFG_S

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, which, steps, graph, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(0)
    torch.manual_seed(77)                                       # the MoCo batch-shuffle permutation comes from torch's generator
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import fill
        from helpers import closed_queue, views
        from src.encoder import AudioNTT2020Task6
        B, T, K = 16, 96, 256
        if which == "delores_m":
            from src.upstream.delores_m.upstream_expert import Upstream_Expert
            cfg = copy.deepcopy(CFG_M)
            cfg["run"]["precision"] = "bf16"
            m = Upstream_Expert(cfg, base_encoder=AudioNTT2020Task6, num_negatives=K)
        else:
            from src.upstream.delores_s.upstream_expert import Upstream_Expert
            cfg = copy.deepcopy(CFG_S)
            cfg["run"]["precision"] = "bf16"
            m = Upstream_Expert(cfg, base_encoder=AudioNTT2020Task6)
        fill.fill_state_dict_(m, seed=33)                      # identical weights on both ranks
        if which == "delores_m":
            for pq, pk in zip(m.encoder_q.parameters(), m.encoder_k.parameters()):
                pk.data.copy_(pq.data)
            m.queue.copy_(closed_queue(128, K))
        m = m.cuda().train()
        opt = m.configure_optimizers()
        step = m.graphed_step(opt, eager_steps=1) if graph else None
        losses = []
        for s in range(steps):
            a = views(B, T, 9300 + 10 * s + rank).cuda()       # different clips per rank
            b = views(B, T, 9350 + 10 * s + rank).cuda()
            if step is not None:
                loss = step(a, b)
            else:
                opt.zero_grad()
                loss = m.training_step((a, b), s)
                loss.backward()
                m.all_reduce_grads()
                opt.step()
            losses.append(float(loss))
        torch.cuda.synchronize()
        out = {"losses": losses, "w": {n: p.detach().float().cpu().numpy() for n, p in m.named_parameters()},
               "graphs": None if step is None else (sorted(step.phases.graphs), step.phases.broken)}
        if which == "delores_m":
            out["queue"] = m.queue.cpu().numpy()
            out["ptr"] = int(m.queue_ptr[0])
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def _run(which, steps=4, world=2, graph=False):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, which, steps, graph, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


@pytest.mark.parametrize("which", ["delores_m", "delores_s"])
def test_two_rank_step_keeps_replicas_identical(which):
    r0, r1 = _run(which)
    assert all(np.isfinite(r0["losses"])) and all(np.isfinite(r1["losses"]))
    assert r0["losses"] != r1["losses"]                        # local losses differ (different clips) ...
    for n in r0["w"]:                                          # ... the all-reduced update does not
        np.testing.assert_array_equal(r0["w"][n], r1["w"][n], err_msg=n)
    if which == "delores_m":
        assert r0["ptr"] == r1["ptr"] == (4 * 2 * 16) % 256    # the queue advances by the GLOBAL batch
        np.testing.assert_array_equal(r0["queue"], r1["queue"])
        # trainable weights moved, key encoder moved by the EMA only
        assert not np.array_equal(r0["w"]["encoder_q.fc.weight"], r0["w"]["encoder_k.fc.weight"])


@pytest.mark.parametrize("which", ["delores_m", "delores_s"])
def test_two_rank_graph_phases_match_eager(which):
    """Data-parallel graph step (one hipGraph per collective-free phase, collectives in between) against the eagerly
    issued data-parallel step on the same clips: same losses / weights up to fp32 atomic order, replicas identical."""
    from helpers import rel_l2
    e0, _ = _run(which, graph=False)
    g0, g1 = _run(which, graph=True)
    h0, _ = _run(which, graph=True)                           # a second, independent run: the graph step must be repeatable
    np.testing.assert_allclose(h0["losses"], g0["losses"], rtol=2e-3)
    want = ["encoder_bwd", "heads", "key", "moco", "query"] if which == "delores_m" else ["encoder_bwd", "forward"]   # grouped heads: one phase
    assert g0["graphs"] == (want, None) and g1["graphs"] == (want, None)
    for n in g0["w"]:
        np.testing.assert_array_equal(g0["w"][n], g1["w"][n], err_msg=n)
    np.testing.assert_allclose(g0["losses"], e0["losses"], rtol=2e-3)
    for n in g0["w"]:
        assert rel_l2(torch.from_numpy(g0["w"][n]), torch.from_numpy(e0["w"][n])) < 2e-2, n
    if which == "delores_m":
        assert g0["ptr"] == e0["ptr"]
        assert rel_l2(torch.from_numpy(g0["queue"]), torch.from_numpy(e0["queue"])) < 2e-2


def _worker_decar(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import random
        from oracle import fill
        from src.augmentations import AugmentationModule
        from src.dataset import UpstreamFrontEnd
        from src.upstream.decar_v2 import main as DC
        cfg = copy.deepcopy(CFG_S)
        cfg["pretrain"]["input"]["length_wave"] = 0.95
        n_items, K = 128, 16
        np.random.seed(31 + rank)
        random.seed(31 + rank)
        tf = AugmentationModule(cfg, n_items, max_batch=16)
        front = UpstreamFrontEnd(cfg, tf)
        args = DC.default_args(epochs=1, batch_size=32, nmb_prototypes=[K], prototype_sizes=[K], nmb_kmeans_iters=3, d=2048,
                               base_lr=0.3, apply_lr_schedule=True)
        wave = lambda i: torch.from_numpy(fill.uniform((15200,), 700 + i, -0.3, 0.3) +
                                          0.2 * np.sin(2 * np.pi * (150 + 25 * i) * np.arange(15200) / 16000).astype(np.float32))
        state, hist = DC.run(args, n_items, wave, front, max_iters=3, log=lambda *_: None)
        torch.cuda.synchronize()
        ret[rank] = {"losses": [l for _, l in hist], "w": {n: p.detach().float().cpu().numpy() for n, p in state.model.named_parameters()},
                     "lr": state.optimizer.param_groups[0]["lr"], "bank": int((state.local_memory_index >= 0).sum()),
                     "shard": state.local_memory_index.cpu().numpy()}
    finally:
        dist.destroy_process_group()


def test_two_rank_deepcluster_v2_harness_keeps_replicas_identical():
    """BASELINE config 3 plumbing at world size 2 (`extras/decar-v2/main.py:57-292`): contiguous shards, distributed k-means
    (centroids broadcast, counts / sums all-reduced), prototype cross-entropy, ONE flat-gradient all-reduce, LARC, the
    warm-up learning-rate schedule - the two replicas end bit-identical, local losses differ, prototypes stay the centroids."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_decar, args=(2, port, ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert len(r0["losses"]) == 3 and all(np.isfinite(r0["losses"])) and all(np.isfinite(r1["losses"]))
    assert r0["losses"] != r1["losses"]
    for n in r0["w"]:
        np.testing.assert_array_equal(r0["w"][n], r1["w"][n], err_msg=n)
    assert r0["shard"].tolist()[:48] == list(range(48)) and r1["shard"].tolist()[:48] == list(range(64, 112))   # 3 x 16 clips each
    assert 0 < r0["lr"] < 0.3                                       # warm-up value of iteration 2, not base_lr
    np.testing.assert_allclose(np.linalg.norm(r0["w"]["prototypes.prototypes0.weight"], axis=1), 1.0, rtol=1e-4)
