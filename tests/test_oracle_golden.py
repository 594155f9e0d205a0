"""CPU: pin the oracle (our restatement) against fixtures produced by the reference itself."""
import random

import numpy as np
import pytest
import torch

from oracle import augment as OA
from oracle import fill
from oracle import frontend as FE
from oracle import model as OM
from helpers import closed_queue, drop_mask, grad_digest, views


# ---------------------------------------------------------------- G1 window
def test_extract_window_matches_reference(golden):
    rows = golden("window")["rows"]
    for seed, n, first, nz0, last, after in rows:
        random.seed(int(seed))
        wav = torch.arange(int(n), dtype=torch.float32)
        out = FE.extract_window(wav, data_size=1.0)
        assert len(out) == 16000
        assert float(out[0]) == first and float(out[-1]) == last
        assert random.random() == after          # same number of draws consumed
        random.seed(int(seed))
        start, left = FE.window_start(int(n), 16000)
        if n > 16000:
            assert start == int(first)
        else:
            assert start == 0 and left == (16000 - int(n)) // 2


# -------------------------------------------------------------- G2 RunningNorm
def test_running_norm_recurrence(golden):
    g = golden("runnorm")
    rn = OA.RunningNorm(epoch_samples=2 * 3)
    k = 0
    for c in range(64):
        x = torch.from_numpy(fill.normalish((1, 64, 101), 500 + c) * (2.0 + 0.1 * c) - 8.0 + 0.05 * c)
        y = rn(x)
        assert abs(float(rn.mean) - g["mu"][c]) <= 2e-6 * abs(g["mu"][c])
        assert abs(float(rn.std) - g["sd"][c]) <= 2e-6 * abs(g["sd"][c])
        if c in (0, 1, 2, 7, 59, 60, 63):
            np.testing.assert_allclose(y.numpy()[0, ::8, ::10], g["outs"][k], rtol=0, atol=2e-5)
            k += 1
    # frozen after 60 updates
    assert g["mu"][60] == g["mu"][63] and rn.n == 60


# ------------------------------------------------------- G3 augmentation chain
@pytest.mark.parametrize("T", [101, 96])
def test_aug_chain_indices_and_views(golden, cfg_s, T):
    g = golden(f"aug_T{T}")
    np.random.seed(31)
    random.seed(31)
    tf = OA.AugmentationModule(cfg_s, 100)
    v1s, v2s = [], []
    for c in range(20):
        x = torch.from_numpy(fill.normalish((1, 64, T), 1000 + c) * 3.0 - 8.0)
        a, b = tf(x)
        v1s.append(a.numpy()[0])
        v2s.append(b.numpy()[0])
    # bit-exact crop indices, and both RNG streams left in the same state
    assert np.array_equal(np.array(tf.rrc.trace, np.int32), g["ijhw"])
    assert np.random.random() == float(g["np_state_after"])
    assert random.random() == float(g["py_state_after"])
    for k, c in enumerate(g["keep"]):
        np.testing.assert_allclose(v1s[c], g["v1"][k], rtol=0, atol=1e-5)
        np.testing.assert_allclose(v2s[c], g["v2"][k], rtol=0, atol=1e-5)
    dig = np.array([[v.sum(dtype=np.float64), np.abs(v).sum(dtype=np.float64)] for v in v1s + v2s])
    np.testing.assert_allclose(dig, g["digest"], rtol=1e-5, atol=1e-2)


def test_mixup_fifo_wraparound(golden):
    g = golden("mixup_wrap")["outs"]
    np.random.seed(5)
    mix = OA.MixupBYOLA(ratio=0.4, n_memory=8, log_mixup_exp=True)
    k = 0
    for c in range(12):
        x = torch.from_numpy(fill.normalish((1, 8, 6), 2000 + c))
        for _ in range(2):
            np.testing.assert_allclose(mix(x).numpy(), g[k], rtol=0, atol=2e-6)
            k += 1


# ---------------------------------------------------------------- G4 SpecAugment
def test_specaugment_masks(golden):
    g = golden("specaug")
    k = 0
    for seed in range(8):
        x = torch.from_numpy(fill.normalish((101, 64), 3000 + seed))
        random.seed(1234 + seed)
        y = OA.time_mask(OA.freq_mask(x, F=30, num_masks=2), T=40, num_masks=2)
        assert random.random() == g["py_state_after"][seed]
        np.testing.assert_allclose(y.numpy(), g["outs"][k], rtol=0, atol=1e-6)
        random.seed(1234 + seed)
        z = OA.time_mask(OA.freq_mask(x, F=30, num_masks=2, replace_with_zero=True), T=40, num_masks=2,
                         replace_with_zero=True)
        assert np.array_equal(z.numpy(), g["outs"][k + 1])         # zero fill: bit-exact
        k += 2


def test_specaugment_survey_example():
    """seed 1234, F=30 -> f=24, f0=28, end=31; 2nd mask f=0 -> early return."""
    random.seed(1234)
    tr = []
    OA.freq_mask(torch.zeros(101, 64), F=30, num_masks=2, trace=tr)
    assert tr == [(24, 28, 31), (0, tr[1][1], -1)]


# ------------------------------------------------------------------ G5 encoder
@pytest.mark.parametrize("T", [101, 96])
def test_encoder_forward_backward(golden, T):
    g = golden(f"encoder_T{T}")
    enc = OM.AudioNTT2020Task6(64, 2048, True)
    fill.fill_state_dict_(enc, seed=1)
    x = views(2, T, 4000 + T)
    enc.eval()
    with torch.no_grad():
        e = enc(x)
    for got, key in zip(e, ("eval_x1", "eval_x2", "eval_x3", "eval_x")):
        np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-4, atol=1e-5)
    enc.train()
    a = enc(x, drop_mask((2, T // 8, 2048), 4100 + T))
    for got, key in zip(a, ("x1", "x2", "x3", "x")):
        np.testing.assert_allclose(got.detach().numpy(), g[key], rtol=1e-4, atol=1e-5)
    r = [torch.from_numpy(fill.uniform(tuple(v.shape), 4200 + i)) for i, v in enumerate(a)]
    loss = sum((v * w).sum() for v, w in zip(a, r))
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= 1e-4 * abs(float(g["loss"])) + 1e-3
    names, norms, heads = grad_digest(enc)
    assert names == [str(s) for s in g["g_names"]]
    np.testing.assert_allclose(norms, g["g_norms"], rtol=2e-4)
    np.testing.assert_allclose(heads, g["g_heads"], rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(enc.features_1[1].running_mean.numpy(), g["bn1_rm"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(enc.features_3[1].running_var.numpy(), g["bn3_rv"], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------- G6 Barlow
@pytest.mark.parametrize("in_dim", [2048, 1024, 512])
def test_barlow_head(golden, in_dim):
    g = golden("barlow")
    p = OM.Projection(in_dim, 5e-5)
    fill.fill_state_dict_(p, seed=in_dim)
    p.train()
    y1 = torch.from_numpy(fill.uniform((8, in_dim), 5000 + in_dim, 0.0, 2.0)).requires_grad_()
    y2 = torch.from_numpy(fill.uniform((8, in_dim), 5001 + in_dim, 0.0, 2.0)).requires_grad_()
    loss = p(y1, y2)
    loss.backward()
    assert abs(float(loss) - float(g[f"loss_{in_dim}"])) <= 1e-5 * abs(float(g[f"loss_{in_dim}"]))
    np.testing.assert_allclose(y1.grad.numpy(), g[f"dy1_{in_dim}"], rtol=1e-3, atol=1e-9)
    np.testing.assert_allclose(y2.grad.numpy(), g[f"dy2_{in_dim}"], rtol=1e-3, atol=1e-9)
    _, norms, _ = grad_digest(p)
    np.testing.assert_allclose(norms, g[f"gn_{in_dim}"], rtol=1e-4)
    np.testing.assert_allclose(p.bn.running_var.numpy()[:64], g[f"bn_rv_{in_dim}"], rtol=1e-5)


# ---------------------------------------------------------------- G8 contrastive
def test_ntxent_and_cluster_loss(golden):
    g = golden("contrastive")
    for B, tau in ((8, 0.5), (16, 0.07)):
        zi = torch.nn.functional.normalize(torch.from_numpy(fill.normalish((B, 128), 8000 + B)), dim=1).requires_grad_()
        zj = torch.nn.functional.normalize(torch.from_numpy(fill.normalish((B, 128), 8001 + B)), dim=1).requires_grad_()
        loss = OM.nt_xent(zi, zj, tau)
        loss.backward()
        assert abs(float(loss) - float(g[f"nt_{B}"])) < 1e-5
        np.testing.assert_allclose(zi.grad.numpy(), g[f"nt_dzi_{B}"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(zj.grad.numpy(), g[f"nt_dzj_{B}"], rtol=1e-4, atol=1e-7)
    ci = torch.softmax(torch.from_numpy(fill.normalish((24, 16), 8100)), dim=1).requires_grad_()
    cj = torch.softmax(torch.from_numpy(fill.normalish((24, 16), 8101)), dim=1).requires_grad_()
    loss = OM.cluster_loss(ci, cj, 1.0)
    loss.backward()
    assert abs(float(loss) - float(g["cl"])) < 1e-5
    np.testing.assert_allclose(ci.grad.numpy(), g["cl_dci"], rtol=1e-4, atol=1e-7)


# ----------------------------------------------------------------------- LARS
def test_lars(golden):
    g = golden("lars")
    ps = [torch.nn.Parameter(torch.from_numpy(fill.uniform(s, 9000 + i)))
          for i, s in enumerate([(16, 8), (16,), (4, 4, 3, 3)])]
    bufs = {}
    for s in range(3):
        for i, p in enumerate(ps):
            p.grad = torch.from_numpy(fill.uniform(tuple(p.shape), 9100 + 10 * s + i))
        OM.lars_step(ps, bufs, lr=0.2, weight_decay=1.5e-6, momentum=0.9, eta=0.001,
                     weight_decay_filter=True, lars_adaptation_filter=True)
    for i, p in enumerate(ps):
        np.testing.assert_allclose(p.detach().numpy(), g[f"p{i}"], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------- G9 training steps
def test_delores_s_three_steps(golden, cfg_s):
    g = golden("step_delores_s")
    ex = OM.DeloresSExpert(cfg_s)
    fill.fill_state_dict_(ex, seed=2)
    ex.train()
    B, T, Tp = 8, 101, 12
    bufs, losses = {}, []
    params = [p for p in ex.parameters() if p.requires_grad]
    for s in range(3):
        for p in params:
            p.grad = None
        loss = ex.training_loss(views(B, T, 6000 + 2 * s), views(B, T, 6001 + 2 * s),
                                drop_mask((B, Tp, 2048), 6100 + 2 * s), drop_mask((B, Tp, 2048), 6101 + 2 * s))
        loss.backward()
        if s == 0:
            names, norms, heads = grad_digest(ex)
            assert names == [str(n) for n in g["g_names"]]
            np.testing.assert_allclose(norms, g["g_norms"], rtol=5e-4)
        OM.sgd_momentum_step(params, bufs, **{"lr": 0.03, "momentum": 0.9, "weight_decay": 1e-4})
        losses.append(float(loss))
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5)
    sd = ex.state_dict()
    np.testing.assert_allclose(sd["encoder.encoder.features_1.0.weight"].numpy().ravel(), g["w_conv1"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd["p.projector.0.weight"].numpy().ravel()[:256], g["w_p0_head"], rtol=1e-4, atol=1e-7)


def test_delores_m_three_steps(golden, cfg_m):
    g = golden("step_delores_m")
    K = 1024
    em = OM.DeloresMExpert(cfg_m, num_negatives=K)
    fill.fill_state_dict_(em, seed=3)
    for pq, pk in zip(em.encoder_q.parameters(), em.encoder_k.parameters()):
        pk.data.copy_(pq.data)
    em.queue.copy_(closed_queue(128, K))
    em.train()
    B, T, Tp = 8, 101, 12
    bufs, losses, ptrs = {}, [], []
    params = [p for p in em.parameters() if p.requires_grad]
    for s in range(3):
        for p in params:
            p.grad = None
        parts = {}
        loss = em.training_loss(views(B, T, 7000 + 2 * s), views(B, T, 7001 + 2 * s),
                                drop_mask((B, Tp, 2048), 7100 + 2 * s), drop_mask((B, Tp, 2048), 7101 + 2 * s), parts)
        loss.backward()
        if s == 0:
            np.testing.assert_allclose(parts["logits0"].numpy(), g["logits_row0"], rtol=1e-4, atol=1e-4)
            assert abs(float(parts["ce"]) - float(g["ce0"])) < 1e-4
            names, norms, _ = grad_digest(em)
            assert names == [str(n) for n in g["g_names"]]
            np.testing.assert_allclose(norms, g["g_norms"], rtol=5e-4)
        OM.sgd_momentum_step(params, bufs, lr=0.03, momentum=0.9, weight_decay=1e-4)
        losses.append(float(loss))
        ptrs.append(int(em.queue_ptr))
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5)
    assert ptrs == list(g["ptrs"])
    sd = em.state_dict()
    np.testing.assert_allclose(sd["queue"][:, :24].numpy(), g["queue_cols"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd["encoder_k.encoder.features_1.0.weight"].numpy().ravel(), g["wk_conv1"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(sd["encoder_q.encoder.features_1.0.weight"].numpy().ravel(), g["wq_conv1"], rtol=1e-4, atol=1e-6)


CFG_SL_EXTRA = dict(instance_contrastive_dim=128, cluster_contrastive_dim=128)


def slicer_cfg(cfg_s):
    import copy
    c = copy.deepcopy(cfg_s)
    c["pretrain"].update(CFG_SL_EXTRA)
    return c


def test_slicer_two_steps(golden, cfg_s):
    """SLICER (symmetric MoCo + ClusterLoss): every logged loss term, the gradient of the logged total, queue, weights."""
    g = golden("step_slicer")
    K = 256
    ex = OM.SlicerExpert(slicer_cfg(cfg_s), num_negatives=K)
    fill.fill_state_dict_(ex, seed=5)
    for pq, pk in zip(ex.encoder_q.parameters(), ex.encoder_k.parameters()):
        pk.data.copy_(pq.data)
    ex.queue.copy_(closed_queue(128, K))
    ex.train()
    B, T, Tp = 8, 101, 12
    bufs, ptrs, got = {}, [], {"returned": [], "combine": [], "sym": [], "cluster": []}
    params = [p for p in ex.parameters() if p.requires_grad]
    for s in range(2):
        for p in params:
            p.grad = None
        masks = [drop_mask((B, Tp, 2048), 8100 + 4 * s + i) for i in range(4)]     # q(v1), k(v2), q(v2), k(v1)
        parts = {}
        loss = ex.training_loss(views(B, T, 8000 + 2 * s), views(B, T, 8001 + 2 * s), masks, parts)
        loss.backward()
        if s == 0:
            names, norms, _ = grad_digest(ex)
            assert names == [str(n) for n in g["g_names"]]
            np.testing.assert_allclose(norms, g["g_norms"], rtol=5e-4)
        OM.sgd_momentum_step(params, bufs, lr=0.03, momentum=0.9, weight_decay=1e-4)
        got["returned"].append(float(parts["ce_first"]))
        got["combine"].append(float(loss))
        got["sym"].append(float(parts["sym"]))
        got["cluster"].append(float(parts["cluster"]))
        ptrs.append(int(ex.queue_ptr))
    for k, v in got.items():
        np.testing.assert_allclose(v, g[k], rtol=3e-5, err_msg=k)
    assert ptrs == list(g["ptrs"]) == [16, 32]
    sd = ex.state_dict()
    np.testing.assert_allclose(sd["queue"][:, :32].numpy(), g["queue_cols"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd["encoder_q.encoder.features_1.0.weight"].numpy().ravel(), g["wq_conv1"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd["encoder_q.cluster_projector.2.weight"].numpy().ravel()[:256], g["wq_cluster"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd["encoder_k.instance_projector.weight"].numpy().ravel()[:256], g["wk_inst"], rtol=1e-5, atol=1e-7)


def test_decar_v2_model_and_loss(golden):
    """DeepCluster-v2 model forward (both views), prototype scores, CE with ignore_index, all gradients."""
    g = golden("decar_v2_model")
    m = OM.DecarV2Model(512, n_mels=64, d=2048, nmb_prototypes=(1024,))
    assert list(m.state_dict().keys()) == [str(k) for k in g["state_keys"]]
    fill.fill_state_dict_(m, seed=6)
    m.train()
    B, T, Tp = 8, 101, 12
    emb, scores = m([views(B, T, 8400), views(B, T, 8401)], (drop_mask((B, Tp, 2048), 8500), drop_mask((B, Tp, 2048), 8501)))
    targets = torch.from_numpy(g["targets"])
    assert int(targets[3]) == -100
    loss = OM.decar_v2_loss(scores, [targets])
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 2e-5 * float(g["loss"])
    np.testing.assert_allclose(emb[:, :16].detach().numpy(), g["emb_head"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(scores[0][:, :16].detach().numpy(), g["scores_head"], rtol=1e-4, atol=1e-5)
    names, norms, _ = grad_digest(m)
    assert names == [str(n) for n in g["g_names"]]
    np.testing.assert_allclose(norms, g["g_norms"], rtol=5e-4, atol=1e-9)
    sd = m.state_dict()
    np.testing.assert_allclose(sd["projection_head.1.running_mean"][:64].numpy(), g["bn_rm"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd["projection_head.1.running_var"][:64].numpy(), g["bn_rv"], rtol=1e-4, atol=1e-6)


def test_lr_schedules_vs_reference_golden(golden):
    """`adjust_learning_rate` (extras/delores-s/multi_proc.py:45-57) and `cosine_scheduler` (extras/decar-v2/multi_proc.py:61-72):
    the oracle's restatements and the product's (`src.optim`, host logic) against values produced by the reference's functions."""
    from src import optim as P
    g = golden("schedules")
    steps_per_epoch, epochs, bs = 7, 30, 512
    for fn in (lambda s: OM.lars_lr(s, epochs, steps_per_epoch, bs), lambda s: P.lars_adjust_learning_rate(None, s, epochs, steps_per_epoch, bs)):
        got = np.array([fn(s) for s in range(epochs * steps_per_epoch)])
        np.testing.assert_allclose(got[:, 0], g["lars_lr_weights"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(got[:, 1], g["lars_lr_biases"], rtol=1e-12, atol=0)
    for mod in (OM, P):
        np.testing.assert_allclose(mod.cosine_scheduler(4.8, 0.0048, 25, 9, warmup_epochs=10, start_warmup_value=0.3), g["cosine_warm"], rtol=1e-12)
        np.testing.assert_allclose(mod.cosine_scheduler(1.0, 0.1, 6, 5), g["cosine_plain"], rtol=1e-12)
    # the schedule main.py:118-122 builds inline (no function to import): product == oracle restatement, endpoints as written
    a, b = OM.dcv2_lr_schedule(4.8, 0.0048, 15, 7), P.dcv2_lr_schedule(4.8, 0.0048, 15, 7)
    np.testing.assert_allclose(a, b, rtol=1e-12)
    assert a[0] == 0.0 and abs(a[69] - 4.8) < 1e-12 and abs(a[70] - 4.8) < 1e-12 and a[-1] > 0.0048


def test_kmix_oracle_vs_reference_golden(golden):
    """oracle.augment.Kmix against the reference's class (tests/golden/make_goldens.py g14): every index draw (argument and
    value), eight outputs, the numpy stream position."""
    g = golden("kmix")
    F, T, calls = 64, 24, 170
    km = OA.Kmix(ratio=0.4, n_memory=140, log_mixup_exp=True, top_k=16, centroids=torch.from_numpy(g["centroids"]))
    np.random.seed(77)
    ys = []
    for c in range(calls):
        x = torch.from_numpy(fill.normalish((1, F, T), 5000 + c)) * (0.5 + 0.1 * (c % 7)) + 0.2 * (c % 5)
        ys.append(km(x))
    assert km.draws == [tuple(p) for p in g["picks"].tolist() if p[1] >= 0]
    assert np.random.random() == float(g["tail"])
    for c in (0, 1, 5, 127, 128, 129, 141, 169):
        np.testing.assert_allclose(ys[c].numpy(), g[f"y{c}"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("cname", ["plain", "pooled", "transition"])
def test_mvit_block_oracle_vs_reference_block(golden, cname):
    """oracle/mvit.py against outputs of the reference's own `MultiScaleBlock` (mvit/models/attention.py:304-393): the plain
    pre-norm ViT block the AST encoder stacks, a pooled in-stage block with relative positions, and a stage transition."""
    from oracle import mvit as OM
    g = golden("mvit_block")
    P, x, kw, gsalt = OM.golden_case(cname)
    assert sorted(P) == sorted(str(k) for k in g[f"{cname}.state_keys"])
    P = {k: v.requires_grad_(True) for k, v in P.items()}
    x.requires_grad_(True)
    y, hw_out = OM.multiscale_block(P, x, OM.GOLDEN_CONFIGS[cname]["hw"], **kw)
    assert list(hw_out) == list(g[f"{cname}.hw_out"])
    np.testing.assert_allclose(y.detach().numpy(), g[f"{cname}.y"], rtol=1e-4, atol=2e-5)
    (y * torch.from_numpy(fill.uniform(tuple(y.shape), gsalt))).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g[f"{cname}.dx"], rtol=1e-3, atol=2e-5)
    for n, norm, head in zip(g[f"{cname}.g_names"], g[f"{cname}.g_norms"], g[f"{cname}.g_heads"]):
        gr = P[str(n)].grad
        assert abs(float(gr.norm()) - norm) <= 1e-4 * norm + 1e-7, n
        h = gr.flatten()[:8].numpy()
        np.testing.assert_allclose(h, head[:h.size], rtol=2e-3, atol=1e-5)


def test_vit_oracle_block_is_the_unpooled_reference_block(golden):
    """oracle/vit.py's transformer block (the AST-base encoder's) against the reference's `MultiScaleBlock` run without
    pooling / relative positions: pins the block arithmetic that timm (absent here) would otherwise have to vouch for."""
    from oracle import mvit as OM
    from oracle import vit as OV
    g = golden("mvit_block")
    P, x, kw, gsalt = OM.golden_case("plain")
    blk = OV.Block(kw["dim"], kw["heads"], 4.0, 1e-6)
    blk.load_state_dict(P)
    x.requires_grad_(True)
    y = blk(x)
    np.testing.assert_allclose(y.detach().numpy(), g["plain.y"], rtol=1e-4, atol=2e-5)
    (y * torch.from_numpy(fill.uniform(tuple(y.shape), gsalt))).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["plain.dx"], rtol=1e-3, atol=2e-5)
    grads = dict(blk.named_parameters())
    for n, norm in zip(g["plain.g_names"], g["plain.g_norms"]):
        assert abs(float(grads[str(n)].grad.norm()) - norm) <= 1e-4 * norm + 1e-7, n


@pytest.mark.parametrize("case", ["a", "b"])
def test_kmeans_oracle_vs_reference_cluster_memory(golden, case):
    """oracle/kmeans.py against the reference's own `cluster_memory` (extras/decar-v2/utils.py:276-346, run on the CPU by
    make_goldens.py g16 with a no-op `.cuda` and a 1-process gloo group): same seeds -> the centroids after the last M step
    and the assignments in data-set order, including the clusters that stay empty in case b (they keep their seed)."""
    from oracle import kmeans as OK
    g = golden("kmeans_ref")
    N, size_dataset, d, K, iters, seed = (int(v) for v in g[f"{case}.dims"])
    mem = torch.from_numpy(g[f"{case}.mem"])
    torch.manual_seed(seed)
    seed_idx = torch.randperm(N)[:K]                            # the draw the reference makes on rank 0
    assert torch.equal(seed_idx, torch.from_numpy(g[f"{case}.seed_idx"]))
    cent, assign = OK.cluster_memory(mem, mem[seed_idx], n_iters=iters)
    want_c = torch.from_numpy(g[f"{case}.centroids"][-1])
    np.testing.assert_allclose(cent.numpy(), want_c.numpy(), rtol=1e-5, atol=1e-6)
    out = torch.full((size_dataset,), -100, dtype=torch.int64)
    out[torch.from_numpy(g[f"{case}.index"])] = assign
    want_a = torch.from_numpy(g[f"{case}.assignments"])
    assert torch.equal(out == -100, want_a == -100)
    assert float((out == want_a).float().mean()) >= 0.999
    if case == "b":
        counts = np.bincount(assign.numpy(), minlength=K)
        empty = np.nonzero(counts == 0)[0]
        assert len(empty) >= 12
        np.testing.assert_allclose(cent[empty].numpy(), mem[seed_idx][empty].numpy(), rtol=1e-6, atol=1e-7)
