"""Diagnostic (GPU box): per-parameter gradient error of the fused delores_m step vs the CPU oracle."""
import copy, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from oracle import fill, model as OM
from helpers import closed_queue, drop_mask, rel_l2, views
import conftest
from src.encoder import AudioNTT2020Task6
from src.upstream.delores_m.upstream_expert import Upstream_Expert

B, T, Tp, K = int(os.environ.get("B", 32)), 96, 12, 2048
ref = OM.DeloresMExpert(copy.deepcopy(conftest.CFG_M), num_negatives=K)
fill.fill_state_dict_(ref, seed=9)
for pq, pk in zip(ref.encoder_q.parameters(), ref.encoder_k.parameters()):
    pk.data.copy_(pq.data)
ref.queue.copy_(closed_queue(128, K)); ref.train()
a, b = views(B, T, 8800), views(B, T, 8801)
mq, mk = drop_mask((B, Tp, 2048), 8802), drop_mask((B, Tp, 2048), 8803)
parts = {}
lr = ref.training_loss(a, b, mq, mk, parts); lr.backward()
for prec in ("fp32", "bf16", "bf16_hp"):
    cfg = copy.deepcopy(conftest.CFG_M); cfg["run"]["precision"] = prec
    em = Upstream_Expert(cfg, base_encoder=AudioNTT2020Task6, num_negatives=K)
    fill.fill_state_dict_(em, seed=9)
    for pq, pk in zip(em.encoder_q.parameters(), em.encoder_k.parameters()):
        pk.data.copy_(pq.data)
    em.queue.copy_(closed_queue(128, K)); em = em.cuda().train()
    em.encoder_q.encoder.dropout_masks.queue = [mq]; em.encoder_k.encoder.dropout_masks.queue = [mk]
    gp = {}
    loss = em.fused_loss(a.cuda(), b.cuda(), True, gp); em.flat.attach_grads()
    print(prec, "loss", float(loss), float(lr), gp["losses"].cpu().numpy(), [float(parts[k]) for k in ("ce","b1","b2","b3")])
    rp = dict(ref.named_parameters())
    for n, p in em.named_parameters():
        if p.grad is None: continue
        g, r = p.grad.float().cpu(), rp[n].grad
        print(f"  {n:45s} |g|={float(r.norm()):.3e} rel_l2={rel_l2(g, r):.2e}")
