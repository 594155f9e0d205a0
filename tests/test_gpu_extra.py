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
