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
        mk = views
        if which.startswith("ssmast"):
            from src.upstream.ssmast.upstream_expert import Upstream_Expert
            B, T, K = 8, 101, 64
            be = {"type": "MAST", "output_dim": 768, "depth": 2, "num_heads": 12, "fstride": 10, "tstride": 10, "return_all_layers": False}
            if which == "ssmast_mvit":                      # the MViTv2 encoder, three blocks with one stage transition
                be.update(model_size="mvit", mvit=dict(depth=3, dim_mul=((1, 2.0),), head_mul=((1, 2.0),), q_strides=((1, 2, 2),)))
            cfg = {"run": {"batch_size": 8, "precision": "bf16"},
                   "pretrain": {"base_encoder": be, "normalization": "mean_var",
                                "input": {"type": "raw_wav", "sampling_rate": 16000, "length_wave": 1.0, "n_mels": 128}}}
            m = Upstream_Expert(cfg, num_negatives=K)
            mk = lambda b, t, salt: torch.cat([views(b, t, salt), views(b, t, salt + 5000)], dim=2).contiguous()
        elif which == "delores_m":
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
        if which == "delores_m" or which.startswith("ssmast"):
            for pq, pk in zip(m.encoder_q.parameters(), m.encoder_k.parameters()):
                pk.data.copy_(pq.data)
            m.queue.copy_(closed_queue(m.queue.shape[0], K))
        m = m.cuda().train()
        opt = m.configure_optimizers()
        step = m.graphed_step(opt, eager_steps=1) if graph else None
        losses = []
        for s in range(steps):
            a = mk(B, T, 9300 + 10 * s + rank).cuda()          # different clips per rank
            b = mk(B, T, 9350 + 10 * s + rank).cuda()
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
        if which == "delores_m" or which.startswith("ssmast"):
            out["queue"] = m.queue.cpu().numpy()
            out["ptr"] = int(m.queue_ptr[0])
        if which.startswith("ssmast"):
            out["shadow_ok"] = bool(torch.equal(m.queue_shadow(1).float(), m.queue.bfloat16().float()))
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
    """BASELINE config 3 plumbing at world size 2 (`extras/decar-v2/main.py:57-292`): DistributedSampler shards, distributed k-means
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
    from torch.utils.data.distributed import DistributedSampler
    for r, got in ((0, r0), (1, r1)):                               # 3 x 16 clips each, DistributedSampler's share at epoch 0
        want = list(DistributedSampler(range(128), 2, r, shuffle=True, seed=0))[:48]
        assert got["shard"].tolist()[:48] == want
    assert 0 < r0["lr"] < 0.3                                       # warm-up value of iteration 2, not base_lr
    np.testing.assert_allclose(np.linalg.norm(r0["w"]["prototypes.prototypes0.weight"], axis=1), 1.0, rtol=1e-4)


def _decar_syncbn_step(world, rank, prec="fp32"):
    """One forward / backward of the DeepCluster-v2 model on this rank's slice of a fixed 16-clip batch -> numpy results."""
    import types
    from oracle import fill
    from helpers import drop_mask, views
    from src import _native as N
    from src import engine as E
    from src.flat import FlatGroup
    from src.upstream.decar_v2.kmeans import prototype_cross_entropy
    from src.upstream.decar_v2.model import AudioNTT2020
    B, T, K = 16, 96, 32
    args = types.SimpleNamespace(nmb_prototypes=[K], prototype_sizes=[K], crops_for_assign=[0])
    m = AudioNTT2020(args, 512, n_mels=64, d=2048, nmb_prototypes=[K])
    fill.fill_state_dict_(m, seed=41)
    m = m.cuda().train()
    m.precision = {"fp32": N.F32, "bf16": N.BF16}[prec]
    named = [(n, p) for n, p in m.named_parameters()]
    flat = FlatGroup(named)
    flat.zero_grad()
    flat.attach_grads()
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    x1, x2 = views(B, T, 9900)[lo:hi].cuda(), views(B, T, 9901)[lo:hi].cuda()
    m.dropout_masks.queue = [drop_mask((B, T // 8, 2048), 9902)[lo:hi], drop_mask((B, T // 8, 2048), 9903)[lo:hi]]
    tgt = torch.from_numpy((fill.uniform01((B,), 9904) * K).astype(np.int64))[lo:hi].cuda()
    E.set_sync_bn(E.SyncBN() if world > 1 else None)
    try:
        emb, scores = m([x1, x2])
        loss = prototype_cross_entropy(scores[0], tgt)
        loss.backward()
        if world > 1:
            dist.all_reduce(flat.grad)
            flat.grad.div_(world)
        torch.cuda.synchronize()
    finally:
        E.set_sync_bn(None)
    return {"emb": emb.detach().float().cpu().numpy(), "loss": float(loss),
            "g": {n: flat.grad_view(i).detach().cpu().numpy().copy() for i, (n, _) in enumerate(named)},
            "rm1": m.features[1].running_mean.cpu().numpy(), "rv3": m.features[9].running_var.cpu().numpy(),
            "rmp": m.projection_head[1].running_mean.cpu().numpy(), "rvp": m.projection_head[1].running_var.cpu().numpy()}


def _worker_syncbn(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _decar_syncbn_step(world, rank)
    finally:
        dist.destroy_process_group()


def test_two_rank_sync_batchnorm_equals_one_rank_on_the_whole_batch():
    """SyncBatchNorm as a statistics exchange (`nn.SyncBatchNorm.convert_sync_batchnorm` + DDP, extras/decar-v2/
    main.py:82-84): two ranks with 8 clips each - stem tap moments, conv sum / sumsq, projection-head column sums all-reduced
    forward, the BatchNorm-backward sums all-reduced backward, gradients averaged - against ONE rank on all 16 clips with plain
    BatchNorm: same embeddings, same loss, same gradients (fp32 path: 2e-3), same running statistics on both ranks."""
    from helpers import rel_l2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_syncbn, args=(2, port, ret), nprocs=2, join=True)
    one = _decar_syncbn_step(1, 0)
    r0, r1 = ret[0], ret[1]
    emb = np.concatenate([r0["emb"], r1["emb"]])
    assert rel_l2(torch.from_numpy(emb), torch.from_numpy(one["emb"])) < 2e-4
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - one["loss"]) <= 2e-4 * abs(one["loss"])
    for n, g in one["g"].items():
        if n in ("features.0.bias", "features.4.bias", "features.8.bias", "projection_head.0.bias"):
            assert float(np.abs(r0["g"][n]).max()) < 1e-6        # a bias in front of a train-mode BatchNorm: zero gradient (rounding residue)
            continue
        np.testing.assert_array_equal(r0["g"][n], r1["g"][n], err_msg=n)
        assert rel_l2(torch.from_numpy(r0["g"][n]), torch.from_numpy(g)) < 2e-3, n
    for k in ("rm1", "rv3", "rmp", "rvp"):
        np.testing.assert_allclose(r0[k], one[k], rtol=2e-4, atol=1e-6, err_msg=k)
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)


def _delores_s_syncbn_step(world, rank, prec):
    """One fused forward / backward of DeLoRes-S on this rank's slice of a fixed 16-clip batch, with the extras trainer's
    semantics at world > 1: SyncBatchNorm everywhere (projector included) and the cross-GPU Barlow correlation."""
    from oracle import fill
    from helpers import drop_mask, views
    from src import engine as E
    from src.encoder import AudioNTT2020Task6
    from src.upstream.delores_s.upstream_expert import Upstream_Expert
    B, T = 16, 96
    cfg = copy.deepcopy(CFG_S)
    cfg["run"]["precision"] = prec
    cfg["run"]["cross_gpu_barlow"] = world > 1
    m = Upstream_Expert(cfg, base_encoder=AudioNTT2020Task6)
    fill.fill_state_dict_(m, seed=35)
    m = m.cuda().train()
    flat = m.ensure_flat()
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    x1, x2 = views(B, T, 9700)[lo:hi].cuda(), views(B, T, 9701)[lo:hi].cuda()
    m.encoder.encoder.dropout_masks.queue = [drop_mask((B, T // 8, 2048), 9702)[lo:hi], drop_mask((B, T // 8, 2048), 9703)[lo:hi]]
    E.set_sync_bn(E.SyncBN() if world > 1 else None)
    try:
        loss = m.fused_loss(x1, x2, True)
        if world > 1:
            m.all_reduce_grads()                                    # joins the step's own all-reduces: the MEAN over ranks, as DistributedDataParallel
            flat.grad.mul_(world)                                   # with the global correlation every rank backpropagates the same loss through its
        torch.cuda.synchronize()                                    # own clips: the SUM over ranks is the whole-batch gradient
    finally:
        E.set_sync_bn(None)
    sd = m.state_dict()
    return {"loss": float(loss), "g": {n: flat.grad_view(i).detach().cpu().numpy().copy() for i, n in enumerate(flat.names)},
            "rm": {k: v.float().cpu().numpy() for k, v in sd.items() if k.endswith(("running_mean", "running_var"))}}


def _worker_syncbn_s(rank, world, port, prec, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _delores_s_syncbn_step(world, rank, prec)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_two_rank_sync_batchnorm_barlow_projector_equals_one_rank(prec):
    """SyncBatchNorm of the Barlow projector (`extras/delores-s/main.py:79` converts EVERY BatchNorm; models_byol.py:108-112 sums
    the correlation over the ranks): two ranks with 8 clips each - global statistics in every BatchNorm forward and backward of
    encoder AND projector, the first projector GEMM on the un-centred hi + lo operand, the D x D correlation all-reduced - against
    ONE rank on all 16 clips with plain BatchNorm: same loss on both ranks, the summed gradients equal the whole-batch gradient,
    the running statistics agree.  bf16: the one-rank run centres the projector input per view, the two-rank run splits it."""
    from helpers import rel_l2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_syncbn_s, args=(2, port, prec, ret), nprocs=2, join=True)
    one = _delores_s_syncbn_step(1, 0, prec)
    r0, r1 = ret[0], ret[1]
    # bf16: two bf16 runs with different arithmetic in the first projector layer and different BatchNorm partial sums; the end of the
    # backward chain (stem weights) differs most
    tol_l, tol_g = (2e-4, 3e-3) if prec == "fp32" else (2e-2, 1e-1)
    assert abs(r0["loss"] - r1["loss"]) <= 2e-6 * abs(r0["loss"])    # the loss of the GLOBAL correlation on both ranks (replica sums: fp32 atomics)
    assert abs(r0["loss"] - one["loss"]) <= tol_l * abs(one["loss"])
    for n, g in one["g"].items():
        np.testing.assert_array_equal(r0["g"][n], r1["g"][n], err_msg=n)
        if float(np.abs(g).max()) < 1e-6 and float(np.abs(r0["g"][n]).max()) < 1e-5:
            continue                                                # a bias in front of a train-mode BatchNorm: zero gradient
        assert rel_l2(torch.from_numpy(r0["g"][n]), torch.from_numpy(g)) < tol_g, n
    for k in one["rm"]:
        np.testing.assert_allclose(r0["rm"][k], one["rm"][k], rtol=2e-3 if prec == "fp32" else 2e-2, atol=1e-5 if prec == "fp32" else 2e-3,
                                   err_msg=k)
        np.testing.assert_array_equal(r0["rm"][k], r1["rm"][k], err_msg=k)


@pytest.mark.parametrize("which", ["ssmast", "ssmast_mvit"])
def test_two_rank_ssmast_eager_and_graph_phases(which):
    """BASELINE config 4 is an 8-GPU configuration: the SS-MAST step (`extras/mast_new/mast/moco_model.py:253-340`: both directions,
    batch shuffle / unshuffle around the key encoder, key gather before each enqueue, one all-reduce of the flat gradient, AdamW)
    on two ranks sharing the GPU over gloo - with the AST-base blocks and with the MViTv2 pooling-attention blocks.  Replicas stay
    bit-identical, the queue advances by twice the GLOBAL batch per step and its bf16 shadow follows it; the graph-phase step
    (one hipGraph per collective-free phase) reproduces the eagerly issued one."""
    from helpers import rel_l2
    steps = 3
    e0, e1 = _run(which, steps=steps, graph=False)
    assert all(np.isfinite(e0["losses"])) and e0["losses"] != e1["losses"]
    for n in e0["w"]:
        np.testing.assert_array_equal(e0["w"][n], e1["w"][n], err_msg=n)
    assert e0["ptr"] == e1["ptr"] == (steps * 2 * 2 * 8) % 64
    np.testing.assert_array_equal(e0["queue"], e1["queue"])
    assert e0["shadow_ok"] and e1["shadow_ok"]
    assert not np.array_equal(e0["w"]["encoder_q.fc.weight"], e0["w"]["encoder_k.fc.weight"])
    g0, g1 = _run(which, steps=steps, graph=True)
    names, broken = g0["graphs"]
    assert broken is None and g1["graphs"][1] is None
    assert set(names) == {"prep", "query0", "key0", "moco0", "enqueue0", "query1", "key1", "moco1", "enqueue1", "backward"}
    for n in g0["w"]:
        np.testing.assert_array_equal(g0["w"][n], g1["w"][n], err_msg=n)
    np.testing.assert_allclose(g0["losses"], e0["losses"], rtol=2e-3)
    assert g0["ptr"] == e0["ptr"] and g0["shadow_ok"]
    for n in g0["w"]:
        assert rel_l2(torch.from_numpy(g0["w"][n]), torch.from_numpy(e0["w"][n])) < 2e-2, n


def _downstream_steps(world, rank):
    """Two optimisation steps of the frozen-encoder linear probe on this rank's slice of fixed 16-clip batches (train() mode:
    BatchNorm on batch statistics), after `train_downstream.sync_replicas` -> final weights, losses, a running statistic."""
    import importlib.util
    from oracle import fill
    from helpers import drop_mask, views
    from src import _native as N
    from src import engine as E
    from src.downstream import DownstreamEncoder
    from src.encoder import AudioNTT2020Task6
    from src.utils import freeze_encoder
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_downstream_hip", os.path.join(root, "audio-ssl_amd", "train_downstream.py"))
    td = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(td)
    dcfg = {"downstream": {"finetune_layer": -1, "base_encoder": {"return_all_layers": False, "output_dim": 2048}, "input": {"n_mels": 64}}}
    B, T, C = 16, 96, 7
    torch.manual_seed(1000 + 17 * rank)                         # every rank draws ANOTHER random head, as separate processes do
    model = DownstreamEncoder(dcfg, None, AudioNTT2020Task6, C)
    sd = model.state_dict()
    enc = {k: v for k, v in sd.items() if k.startswith("encoder.")}
    fill.fill_state_dict_(model.encoder, seed=23)               # the (pre-trained) encoder is the same file on every rank
    model = model.cuda()
    model.encoder.precision = N.F32
    freeze_encoder(model)
    head_before = model.final.weight.detach().cpu().clone()
    td.sync_replicas(model, world)
    try:
        tr = td.ProbeTrainer(model, lr=1e-2)
        model.train()
        lo, hi = rank * (B // world), (rank + 1) * (B // world)
        losses = []
        for s in range(2):
            x = views(B, T, 9850 + s)[lo:hi]
            y = torch.from_numpy((fill.uniform01((B,), 9860 + s) * C).astype(np.int64))[lo:hi]
            model.encoder.dropout_masks.queue = [drop_mask((B, T // 8, 2048), 9870 + s)[lo:hi]]
            losses.append(float(tr.step(x.cuda(), y.cuda())))
        torch.cuda.synchronize()
    finally:
        E.set_sync_bn(None)
    return {"head0": head_before.numpy(), "w": model.final.weight.detach().cpu().numpy(), "b": model.final.bias.detach().cpu().numpy(),
            "losses": losses, "rm": model.encoder.features_1[1].running_mean.cpu().numpy()}


def _worker_downstream(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _downstream_steps(world, rank)
    finally:
        dist.destroy_process_group()


def test_two_rank_downstream_probe_is_one_model():
    """`train_downstream.py:80-85` of the reference wraps the model in SyncBatchNorm + DistributedDataParallel: rank 0's parameters
    everywhere, BatchNorm on the global batch.  Here: two ranks that drew DIFFERENT random heads end up, after `sync_replicas`
    and two all-reduced Adam steps on 8 clips each, bit-identical to each other and equal to ONE rank training rank 0's head on
    all 16 clips (fp32 path: 2e-4 on the losses, 2e-3 on the head)."""
    from helpers import rel_l2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_downstream, args=(2, port, ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert not np.array_equal(r0["head0"], r1["head0"])             # they really started apart
    np.testing.assert_array_equal(r0["w"], r1["w"])
    np.testing.assert_array_equal(r0["b"], r1["b"])
    np.testing.assert_array_equal(r0["rm"], r1["rm"])
    one = _downstream_steps(1, 0)                                   # same seed as rank 0: the head rank 0 broadcast
    np.testing.assert_array_equal(one["head0"], r0["head0"])
    for s in range(2):
        assert abs(0.5 * (r0["losses"][s] + r1["losses"][s]) - one["losses"][s]) <= 2e-4 * abs(one["losses"][s])
    assert rel_l2(torch.from_numpy(r0["w"]), torch.from_numpy(one["w"])) < 2e-3
    np.testing.assert_allclose(r0["rm"], one["rm"], rtol=2e-4, atol=1e-6)
