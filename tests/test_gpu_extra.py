"""GPU: the remaining SURVEY 8(a) heads - NT-Xent / ClusterLoss (a18), DeepCluster-v2 k-means + prototype CE (a19),
LARS (a21) - against the reference's goldens and the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import fill
from oracle import kmeans as OK
from oracle import model as OM
from helpers import rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_ntxent_instance_loss_vs_reference_golden(golden, prec):
    from src.upstream.slicer.losses import InstanceLoss
    g = golden("contrastive")
    td = torch.float32 if prec == "fp32" else torch.bfloat16
    for B, tau in ((8, 0.5), (16, 0.07)):
        zi = torch.nn.functional.normalize(torch.from_numpy(fill.normalish((B, 128), 8000 + B)), dim=1).cuda().to(td).requires_grad_()
        zj = torch.nn.functional.normalize(torch.from_numpy(fill.normalish((B, 128), 8001 + B)), dim=1).cuda().to(td).requires_grad_()
        loss = InstanceLoss(B, tau, "cuda")(zi, zj)
        loss.backward()
        tol = 1e-5 if prec == "fp32" else 3e-2
        assert abs(float(loss) - float(g[f"nt_{B}"])) < tol * max(1.0, float(g[f"nt_{B}"]))
        assert rel_l2(zi.grad.float().cpu(), g[f"nt_dzi_{B}"]) < (1e-4 if prec == "fp32" else 5e-2)
        assert rel_l2(zj.grad.float().cpu(), g[f"nt_dzj_{B}"]) < (1e-4 if prec == "fp32" else 5e-2)


def test_cluster_loss_vs_reference_golden(golden):
    from src.upstream.slicer.losses import ClusterLoss
    g = golden("contrastive")
    ci = torch.softmax(torch.from_numpy(fill.normalish((24, 16), 8100)), dim=1).cuda().requires_grad_()
    cj = torch.softmax(torch.from_numpy(fill.normalish((24, 16), 8101)), dim=1).cuda().requires_grad_()
    loss = ClusterLoss(16, 1.0, "cuda")(ci, cj)
    loss.backward()
    assert abs(float(loss) - float(g["cl"])) < 1e-5
    assert rel_l2(ci.grad.cpu(), g["cl_dci"]) < 1e-4


def test_ntxent_larger_vs_oracle():
    from src.upstream.slicer.losses import InstanceLoss
    B = 512
    zi = torch.nn.functional.normalize(torch.from_numpy(fill.normalish((B, 128), 1)), dim=1)
    zj = torch.nn.functional.normalize(torch.from_numpy(fill.normalish((B, 128), 2)), dim=1)
    a, b = zi.clone().requires_grad_(), zj.clone().requires_grad_()
    want = OM.nt_xent(a, b, 0.07)
    want.backward()
    x, y = zi.cuda().requires_grad_(), zj.cuda().requires_grad_()
    loss = InstanceLoss(B, 0.07, "cuda")(x, y)
    loss.backward()
    assert abs(float(loss) - float(want)) < 1e-4
    assert rel_l2(x.grad.cpu(), a.grad) < 1e-4


def test_lars_vs_reference_golden(golden):
    from src.flat import FlatGroup
    from src.optim import HipLARS
    g = golden("lars")
    ps = [torch.nn.Parameter(torch.from_numpy(fill.uniform(s, 9000 + i)).cuda()) for i, s in enumerate([(16, 8), (16,), (4, 4, 3, 3)])]
    fg = FlatGroup([(f"p{i}", p) for i, p in enumerate(ps)])
    opt = HipLARS([fg], ps, lr_weights=0.2, lr_biases=0.2, weight_decay=1.5e-6, momentum=0.9, eta=0.001,
                  weight_decay_filter=True, lars_adaptation_filter=True)
    for s in range(3):
        fg.zero_grad()
        for i, p in enumerate(ps):
            fg.grad_view(i).copy_(torch.from_numpy(fill.uniform(tuple(p.shape), 9100 + 10 * s + i)))
        opt.step()
    for i, p in enumerate(ps):
        np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"p{i}"], rtol=2e-6, atol=2e-7)


def test_spherical_kmeans_vs_oracle():
    from src.upstream.decar_v2.kmeans import cluster_memory, spherical_kmeans
    Nn, D, K = 4096, 512, 64
    mem = torch.nn.functional.normalize(torch.from_numpy(fill.normalish((Nn, D), 11)).abs() + 0.05 *
                                        torch.from_numpy(fill.normalish((Nn, D), 12)), dim=1)
    init = mem[torch.arange(K) * 7].clone()
    init[5] = -torch.ones(D) / D ** 0.5                       # negative dot with every (mostly positive) point: stays empty
    c_ref, a_ref = OK.cluster_memory(mem, init, n_iters=10)
    c, a = spherical_kmeans(mem.cuda(), K, 10, centroids=init.cuda())
    agree = float((a.cpu() == a_ref).float().mean())
    assert agree >= 0.999, agree                               # >= 99.9 % identical (ties)
    assert rel_l2(c.cpu(), c_ref) < 1e-3
    assert int((a.cpu() == 5).sum()) == 0 and torch.allclose(c[5].cpu(), init[5], atol=1e-6)   # empty cluster untouched
    # dataset-order scatter with unseen entries left at -100
    idx = torch.arange(Nn).cuda() * 2
    out, _ = cluster_memory(mem.cuda(), idx, 2 * Nn, K, 2)
    assert int((out == -100).sum()) == Nn and out[0] >= 0


@pytest.mark.parametrize("case", ["a", "b"])
def test_spherical_kmeans_vs_reference_cluster_memory(golden, case):
    """The HIP k-means (GEMM + row_argmax E step, kmeans_accumulate / kmeans_update M step) against the fixture produced by the
    reference's own `cluster_memory` (extras/decar-v2/utils.py:276-346; tests/golden/make_goldens.py g16): seeded exactly like
    it (`torch.randperm` on the host generator, rank 0), centroids after every iteration, assignments in data-set order >= 99.9 %
    identical (ties), unseen entries -100, empty clusters keep their centroid (case b)."""
    from src.upstream.decar_v2.kmeans import cluster_memory, spherical_kmeans
    g = golden("kmeans_ref")
    N, size_dataset, d, K, iters, seed = (int(v) for v in g[f"{case}.dims"])
    mem = torch.from_numpy(g[f"{case}.mem"]).cuda()
    index = torch.from_numpy(g[f"{case}.index"]).cuda()
    want_a = torch.from_numpy(g[f"{case}.assignments"])
    for it in (1, iters):                                        # the centroid trajectory, first and last iteration
        c, _ = spherical_kmeans(mem, K, it, centroids=mem[torch.from_numpy(g[f"{case}.seed_idx"]).cuda()])
        np.testing.assert_allclose(c.cpu().numpy(), g[f"{case}.centroids"][it - 1], rtol=2e-5, atol=2e-6)
    torch.manual_seed(seed)                                      # cluster_memory draws its own seeds, as the reference does
    out, cent = cluster_memory(mem, index, size_dataset, K, iters)
    out = out.cpu()
    assert torch.equal(out == -100, want_a == -100) and int((want_a == -100).sum()) == size_dataset - N
    np.testing.assert_allclose(cent.cpu().numpy(), g[f"{case}.centroids"][-1], rtol=2e-5, atol=2e-6)
    if case == "a":
        assert float((out == want_a).float().mean()) >= 0.999
    else:
        # case b is degenerate by construction (512 rows = copies of 20 directions, duplicate seeds): a point has the SAME dot
        # product, to the last ulp or not, with the centroid of its copies and with every unused duplicate seed, so the label of
        # the final arg-max is a coin toss between equal centroids.  What must hold: every point sits on a centroid that is
        # optimal for it (dot within 2e-6 of the best), and the clusters the REFERENCE left empty still carry their seed.
        ref_c = torch.from_numpy(g[f"{case}.centroids"][-1]).double()
        dots = mem.cpu().double() @ ref_c.T
        mine = out[index.cpu()]
        got = dots[torch.arange(N), mine]
        assert float((dots.max(1).values - got).max()) <= 2e-6
        ref_seen = want_a[want_a >= 0]
        empty = np.nonzero(np.bincount(ref_seen.numpy(), minlength=K) == 0)[0]
        assert len(empty) >= 12
        seeds = mem[torch.from_numpy(g[f"{case}.seed_idx"]).cuda()].cpu().numpy()
        np.testing.assert_allclose(cent.cpu().numpy()[empty], seeds[empty], rtol=1e-6, atol=1e-7)


def test_prototype_cross_entropy():
    from src.upstream.decar_v2.kmeans import prototype_cross_entropy
    B, K = 96, 1024
    logits = torch.from_numpy(fill.normalish((B, K), 21) * 3).requires_grad_()
    tgt = torch.from_numpy((fill.uniform01((B,), 22) * K).astype(np.int64))
    tgt[::7] = -100
    want = torch.nn.functional.cross_entropy(logits, tgt, ignore_index=-100)
    want.backward()
    x = logits.detach().cuda().requires_grad_()
    loss = prototype_cross_entropy(x, tgt.cuda())
    loss.backward()
    assert abs(float(loss) - float(want)) < 1e-5
    assert rel_l2(x.grad.cpu(), logits.grad) < 1e-5


# ------------------------------------------------------------------------------------------------ DeepCluster-v2 model
def _decar_model(prec):
    import types
    from src import _native as N
    from src.upstream.decar_v2.model import AudioNTT2020
    args = types.SimpleNamespace(nmb_prototypes=[1024], crops_for_assign=[0])
    m = AudioNTT2020(args, 512, n_mels=64, d=2048, nmb_prototypes=[1024])
    m.precision = {"fp32": N.F32, "bf16": N.BF16}[prec]
    return m


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_decar_v2_model_vs_reference_golden(golden, prec):
    """`extras/decar-v2/models_delores.py` AudioNTT2020 + prototype CE: state_dict keys, embedding, scores, loss, every
    gradient norm and the BatchNorm1d running buffers against numbers produced by the reference classes."""
    from helpers import drop_mask, grad_digest, views
    from src.upstream.decar_v2.kmeans import prototype_cross_entropy
    g = golden("decar_v2_model")
    m = _decar_model(prec)
    assert list(m.state_dict().keys()) == [str(k) for k in g["state_keys"]]
    fill.fill_state_dict_(m, seed=6)
    m = m.cuda().train()
    B, T, Tp = 8, 101, 12
    m.dropout_masks.queue = [drop_mask((B, Tp, 2048), 8500), drop_mask((B, Tp, 2048), 8501)]
    emb, out = m([views(B, T, 8400).cuda(), views(B, T, 8401).cuda()])
    targets = torch.from_numpy(g["targets"]).cuda()
    loss = prototype_cross_entropy(out[0], targets)
    loss.backward()
    tol = {"fp32": 2e-4, "bf16": 2e-2}[prec]
    assert abs(float(loss) - float(g["loss"])) <= tol * float(g["loss"])
    assert rel_l2(emb[:, :16].float().cpu(), g["emb_head"]) < tol * 2
    assert rel_l2(out[0][:, :16].float().cpu(), g["scores_head"]) < tol * 2
    names, norms, _ = grad_digest(m)
    assert names == [str(n) for n in g["g_names"]]
    gt = {"fp32": 2e-3, "bf16": 1e-1}[prec]
    for n, got, want in zip(names, norms, g["g_norms"]):
        if n in ("features.0.bias", "features.4.bias", "features.8.bias", "projection_head.0.bias"):
            continue                  # a bias in front of a BatchNorm cancels: its exact gradient is 0, both sides hold noise
        assert abs(got - want) <= gt * want + 1e-6 * float(np.max(g["g_norms"])), (n, got, want)
    sd = m.state_dict()
    assert rel_l2(sd["projection_head.1.running_mean"][:64].cpu(), g["bn_rm"]) < tol * 2
    assert rel_l2(sd["projection_head.1.running_var"][:64].cpu(), g["bn_rv"]) < tol * 2


def test_decar_v2_epoch_end_to_end():
    """init_memory -> distributed k-means (world 1) -> two training steps: assignments cover every clip, prototypes equal
    the centroids and stay frozen, encoder weights move, memory bank rows are overwritten by the new embeddings."""
    from helpers import views
    from src.optim import HipSGD
    from src.upstream.decar_v2.train import DeepClusterState, cluster_epoch, init_memory, train_step
    m = _decar_model("bf16")
    fill.fill_state_dict_(m, seed=7)
    m = m.cuda().train()
    n_clips, B, T, K = 64, 16, 96, 1024
    st = DeepClusterState(m, HipSGD, 512, n_clips, lr=0.005, momentum=0.9, weight_decay=1e-6)
    data = [(torch.arange(i * B, (i + 1) * B), [views(B, T, 9500 + i), views(B, T, 9600 + i)]) for i in range(n_clips // B)]
    init_memory(st, data)
    assert float(st.local_memory_embeddings.abs().sum()) > 0 and st.local_memory_index.tolist() == list(range(n_clips))
    # the reference L2-normalises nothing before k-means either; K must not exceed the bank: use a small K on this toy bank
    torch.manual_seed(3)
    with pytest.raises(AssertionError):
        cluster_epoch(st, n_clips, (K,))                                         # 1024 centroids from 64 clips: refused
    m.prototypes.prototypes0 = torch.nn.Linear(512, 32, bias=False).cuda()
    st = DeepClusterState(m, HipSGD, 512, n_clips, lr=0.005, momentum=0.9, weight_decay=1e-6)
    init_memory(st, data)
    assign = cluster_epoch(st, n_clips, (32,))
    assert assign.shape == (1, n_clips) and int(assign.min()) >= 0 and int(assign.max()) < 32
    w0 = m.prototypes.prototypes0.weight.detach().clone()
    np.testing.assert_allclose(w0.norm(dim=1).cpu().numpy(), 1.0, rtol=1e-4)     # centroids are L2-normalised
    enc0 = m.fc[3].weight.detach().clone()
    mem0 = st.local_memory_embeddings.clone()
    start, losses = 0, []
    for idx, inputs in data[:2]:
        loss, start = train_step(st, idx, [x.cuda() for x in inputs], assign, start)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and start == 2 * B
    assert torch.equal(m.prototypes.prototypes0.weight.detach(), w0)             # frozen prototypes
    assert not torch.equal(m.fc[3].weight.detach(), enc0)
    assert not torch.equal(st.local_memory_embeddings[0, :2 * B], mem0[0, :2 * B])
    assert torch.equal(st.local_memory_embeddings[0, 2 * B:], mem0[0, 2 * B:])


# ------------------------------------------------------------------------------- 16-byte-store forms of the small kernels
def test_vectorised_elementwise_kernels_match_scalar_definitions():
    """dropout_mask / cast / ema_update / moco_ce_bwd write 16 bytes per lane where alignment allows; lengths that are not a
    multiple of the vector width and misaligned bases exercise the tails and the scalar fall-backs."""
    from src import _native as N
    # dropout mask = splitmix64 finaliser of (seed + 0x9E37..15 * (i + 1)), keep iff high word >= p * 2^32
    def ref_mask(n, seed, p):
        M = (1 << 64) - 1
        i = np.arange(1, n + 1, dtype=np.uint64)
        with np.errstate(over="ignore"):
            z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * i) & np.uint64(M)
            z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(M)
            z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(M)
            z = z ^ (z >> np.uint64(31))
        thr = np.uint64(min(int(np.float32(p) * np.float32(4294967296.0)), 4294967295))
        return ((z >> np.uint64(32)) >= thr).astype(np.uint8)
    for n, off in ((4099, 0), (4099, 1), (37, 0)):
        buf = torch.zeros(n + off + 16, dtype=torch.uint8, device="cuda")
        N.call("dropout_mask", buf[off:off + n], n, 12345, 0.3, None)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(buf[off:off + n].cpu().numpy(), ref_mask(n, 12345, 0.3))
        assert int(buf[off + n:].sum()) == 0 and int(buf[:off].sum()) == 0
    g = torch.Generator().manual_seed(3)
    for n, off in ((1003, 0), (1003, 1)):
        src = torch.randn(n + off, generator=g).cuda()
        dst = torch.zeros(n + off, dtype=torch.bfloat16, device="cuda")
        N.call("cast", 1, src[off:], dst[off:], n)
        pk, pq = torch.randn(n + off, generator=g).cuda(), torch.randn(n + off, generator=g).cuda()
        want = pk[off:] * 0.99 + pq[off:] * (1.0 - 0.99)
        N.call("ema_update", pk[off:], pq[off:], n, 0.99, None)
        torch.cuda.synchronize()
        assert torch.equal(dst[off:], src[off:].bfloat16())
        np.testing.assert_allclose(pk[off:].cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-7)
        sh = torch.zeros(n + off + 8, dtype=torch.bfloat16, device="cuda")      # the same pass also leaves the bf16 copy
        pk2 = pk.clone()
        N.call("ema_update", pk2[off:], pq[off:], n, 0.5, sh[off:off + n])
        torch.cuda.synchronize()
        assert torch.equal(sh[off:off + n], pk2[off:].bfloat16()) and float(sh[off + n:].abs().sum()) == 0.0
    # moco_prep = l2norm_fwd(q), l2norm_fwd(k), rowdot in one launch: bit-identical
    B, D = 37, 128
    q, k = torch.randn(B, D, generator=g).cuda() * 3, torch.randn(B, D, generator=g).cuda()
    o1 = [torch.empty(B, D, device="cuda", dtype=torch.bfloat16), torch.empty(B, D, device="cuda"), torch.empty(B, device="cuda"),
          torch.empty(B, D, device="cuda", dtype=torch.bfloat16), torch.empty(B, D, device="cuda"), torch.empty(B, device="cuda"),
          torch.empty(B, device="cuda")]
    o2 = [torch.empty_like(t) for t in o1]
    N.call("l2norm_fwd", 1, q, B, D, o1[0], o1[1], o1[2])
    N.call("l2norm_fwd", 1, k, B, D, o1[3], o1[4], o1[5])
    N.call("rowdot", o1[1], o1[4], B, D, 1.0 / 0.2, o1[6])
    N.call("moco_prep", 1, q, k, B, D, 1.0 / 0.2, *o2)
    torch.cuda.synchronize()
    for u, v in zip(o1[:6], o2[:6]):
        assert torch.equal(u, v)
    np.testing.assert_allclose(o2[6].cpu().numpy(), o1[6].cpu().numpy(), rtol=1e-6, atol=1e-7)      # sum order / contraction only
    for K in (64, 100):                                            # K % 8 != 0 takes the scalar path
        B = 5
        lpos, lneg = torch.randn(B, generator=g).cuda(), torch.randn(B, K, generator=g).cuda()
        lse = torch.logsumexp(torch.cat([lpos[:, None], lneg], 1), 1).contiguous()
        P = torch.empty(B, K, dtype=torch.bfloat16, device="cuda")
        dl = torch.empty(B, device="cuda")
        N.call("moco_ce_bwd", 1, lpos, lneg, lse, B, K, 0.25, P, dl)
        torch.cuda.synchronize()
        np.testing.assert_allclose(P.float().cpu().numpy(), (torch.exp(lneg - lse[:, None]) * 0.25).cpu().numpy(), rtol=1e-2, atol=1e-6)
        np.testing.assert_allclose(dl.cpu().numpy(), ((torch.exp(lpos - lse) - 1) * 0.25).cpu().numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("B,K,dim", [(40, 1000, 128), (512, 65536, 128), (33, 136, 64), (130, 4096, 256)])
def test_moco_fused_softmax_epilogue_vs_materialised_logits(B, K, dim):
    """InfoNCE with the row soft-max folded into the logits GEMM (moco_logits mode 1 + moco_lse_merge, mode 2) against the
    same quantities from logits written out in fp32 and torch's log-sum-exp in fp64 (`delores_m/upstream_expert.py:250-264`);
    K = 65,536 is BASELINE config 2, the odd sizes end inside tiles and slabs."""
    from src import _native as N
    g = torch.Generator().manual_seed(B + K)
    qn = torch.nn.functional.normalize(torch.randn(B, dim, generator=g), dim=1).cuda().bfloat16()
    queue = torch.nn.functional.normalize(torch.randn(dim, K, generator=g), dim=0).cuda().bfloat16()
    lpos = (torch.rand(B, generator=g) * 10.0).cuda()
    T = 0.07
    gscale = 1.0 / (B * T)
    nslot = (K + 63) // 64
    part = torch.full((B, nslot, 2), float("nan"), device="cuda")
    lse, dlpos, loss = torch.empty(B, device="cuda"), torch.empty(B, device="cuda"), torch.zeros(1, device="cuda")
    N.call("moco_logits", 1, qn, queue, B, K, dim, 1.0 / T, part, None, 0.0, None)
    N.call("moco_lse_merge", lpos, part, B, nslot, gscale, lse, loss, dlpos)
    P = torch.full((B, K), float("nan"), device="cuda", dtype=torch.bfloat16)
    N.call("moco_logits", 2, qn, queue, B, K, dim, 1.0 / T, None, lse, gscale, P)
    torch.cuda.synchronize()
    logits = torch.cat([lpos.double()[:, None], qn.double() @ queue.double() / T], 1)
    want_lse = torch.logsumexp(logits, 1)
    assert float((lse.double() - want_lse).abs().max()) < 2e-5 * float(want_lse.abs().max())
    assert abs(float(loss) - float((want_lse - lpos.double()).mean())) < 1e-5 * float(want_lse.abs().max())
    sm = torch.softmax(logits, 1)
    np.testing.assert_allclose(dlpos.cpu().numpy(), ((sm[:, 0] - 1.0) * gscale).cpu().numpy(), rtol=1e-4, atol=1e-7)
    assert rel_l2(P.float().cpu(), (sm[:, 1:] * gscale).cpu()) < 4e-3          # one bf16 rounding
    assert bool(torch.isfinite(P.float()).all())


def test_larc_sgd_vs_oracle():
    """HipLARC (apex LARC(clip=False) around SGD, extras/decar-v2/main.py:92-97, 111) on a flat group: 4 steps against the
    oracle's restatement of apex's published algorithm (apex itself is absent: parity unpinned), incl. a zero-gradient tensor
    (no adaptive rate, no decay), a tensor skipped as "no gradient" (frozen prototypes) and the clip=True form."""
    from src.flat import FlatGroup
    from src.optim import HipLARC
    shapes = [(16, 8), (16,), (4, 4, 3, 3), (8, 8), (5,)]
    for clip in (False, True):
        ref = [torch.nn.Parameter(torch.from_numpy(fill.uniform(s, 9300 + i))) for i, s in enumerate(shapes)]
        ps = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref]
        fg = FlatGroup([(f"p{i}", p) for i, p in enumerate(ps)])
        opt = HipLARC([fg], ps, lr=0.6, momentum=0.9, weight_decay=1e-3, trust_coefficient=0.02, clip=clip)
        bufs = {}
        for s in range(4):
            fg.zero_grad()
            for i, p in enumerate(ref):
                g = torch.from_numpy(fill.uniform(tuple(p.shape), 9400 + 10 * s + i))
                if i == 3:
                    g = torch.zeros_like(g)                      # |g| = 0: plain SGD on a zero gradient
                p.grad = None if i == 4 else g                   # tensor 4: no gradient at all
                fg.grad_view(i).copy_(g)
            opt.step(skip=(4,))
            OM.larc_sgd_step(ref, bufs, 0.6, weight_decay=1e-3, momentum=0.9, trust_coefficient=0.02, clip=clip)
        for i, (p, r) in enumerate(zip(ps, ref)):
            np.testing.assert_allclose(p.detach().cpu().numpy(), r.detach().numpy(), rtol=3e-6, atol=3e-7, err_msg=f"tensor {i} clip {clip}")


def test_offline_pseudolabeler_vs_oracle(tmp_path):
    """`extras/decar-v2/clustering.py` Kmeans(k).cluster(features) + the label CSV of store_clusters.py: PCA-whitening and Lloyd's
    k-means on the GPU against the numpy restatement from the same initial points (faiss absent: parity unpinned)."""
    from src.upstream.decar_v2 import clustering as CL
    n, d, k = 3000, 512, 24
    g = np.random.RandomState(3)
    blobs = g.randn(k, d) * 2.0
    x = (blobs[g.randint(k, size=n)] + g.randn(n, d) * 0.7).astype(np.float32)
    xb = CL.preprocess_features(x, pca=128)
    want = OK.pca_whiten_l2(x, 128)
    # eigenvectors are defined up to sign: compare the Gram matrices (what every distance depends on)
    assert rel_l2((xb @ xb.T).cpu()[:200, :200], torch.from_numpy(want @ want.T)[:200, :200]) < 2e-3
    np.testing.assert_allclose(xb.norm(dim=1).cpu().numpy(), 1.0, rtol=1e-5)
    ids, loss = CL.run_kmeans(xb, k, seed=11)
    init = xb[torch.from_numpy(np.random.RandomState(11).permutation(n)[:k]).cuda()].cpu().numpy()
    a_ref, loss_ref, _ = OK.lloyd(xb.cpu().numpy(), init, 20)
    assert float((np.array(ids) == a_ref).mean()) >= 0.999
    assert abs(loss - loss_ref) <= 1e-3 * loss_ref
    km = CL.Kmeans(k)
    np.random.seed(4)
    km.cluster(x)
    assert sum(len(l) for l in km.images_lists) == n
    labels = CL.write_pseudolabel_csv([f"clip{i}.wav" for i in range(n)], km, str(tmp_path / "labels.csv"))
    lines = open(tmp_path / "labels.csv").read().split()
    assert len(lines) == n and lines[5] == f"clip5.wav,{int(labels[5])}" and labels[km.images_lists[3][0]] == 3


def test_sgd_step_tail_clears_only_what_is_not_stored():
    """`HipSGD.step_tail(stored=...)` (fused tail on): parameters, momentum and bf16 shadow equal the plain fused tail bit for
    bit; afterwards the gradients of the stored tensors are untouched and every other gradient of the slice is zero."""
    from src import _native as N
    from src.flat import FlatGroup
    from src.optim import HipSGD
    shapes = ((64, 9), (256, 128), (256,), (256,), (256, 256), (256,), (256,), (128, 256))
    names = ["enc.w", "h.0.weight", "h.1.weight", "h.1.bias", "h.3.weight", "h.4.weight", "h.4.bias", "h.6.weight"]
    stored = ("h.0.weight", "h.3.weight", "h.6.weight")
    outs = []
    for use_stored in (False, True):
        g = torch.Generator().manual_seed(3)
        ps = [torch.nn.Parameter(torch.randn(*sh, generator=g).cuda()) for sh in shapes]
        fg = FlatGroup(list(zip(names, ps)))
        opt = HipSGD([fg], ps, 0.05, momentum=0.9, weight_decay=1e-3)
        opt.fused_refresh = True
        fg.refresh_shadow(N.BF16)
        fg.momentum = torch.randn(fg.numel, generator=g).cuda()
        fg.grad.copy_(torch.randn(fg.numel, generator=g).cuda())
        g_before = fg.grad.clone()
        start = fg.offsets[1]
        assert opt.step_tail(fg, start, stored=stored if use_stored else ())
        opt.mark_early(fg, start)
        opt.step()
        torch.cuda.synchronize()
        assert fg._fresh_grad == ("partial" if use_stored else True)
        outs.append((fg.data.clone(), fg.momentum.clone(), fg._shadow.clone()))
        for n, p, o in zip(fg.names, fg.params, fg.offsets):
            got = fg.grad[o:o + p.numel()]
            if use_stored and n in stored:
                assert torch.equal(got, g_before[o:o + p.numel()]), n
            else:
                assert float(got.abs().max()) == 0.0, n
    for a, b in zip(*outs):
        assert torch.equal(a, b)

