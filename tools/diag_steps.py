"""Diagnostic (GPU box): per-step gradient / weight drift of delores_s fp32 vs the CPU oracle."""
import copy, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from oracle import fill, model as OM
from helpers import drop_mask, rel_l2, views
import conftest
from src.encoder import AudioNTT2020Task6
from src.upstream.delores_s.upstream_expert import Upstream_Expert

B, T, Tp = int(os.environ.get("B", 8)), 101, 12
ref = OM.DeloresSExpert(copy.deepcopy(conftest.CFG_S)); fill.fill_state_dict_(ref, seed=2); ref.train()
cfg = copy.deepcopy(conftest.CFG_S); cfg["run"]["precision"] = os.environ.get("PREC", "fp32")
ex = Upstream_Expert(cfg, base_encoder=AudioNTT2020Task6); fill.fill_state_dict_(ex, seed=2); ex = ex.cuda().train()
opt = ex.configure_optimizers()
bufs = {}
rparams = [p for p in ref.parameters() if p.requires_grad]
for s in range(4):
    a, b = views(B, T, 6000 + 2 * s), views(B, T, 6001 + 2 * s)
    m1, m2 = drop_mask((B, Tp, 2048), 6100 + 2 * s), drop_mask((B, Tp, 2048), 6101 + 2 * s)
    for p in rparams: p.grad = None
    lr_ = ref.training_loss(a, b, m1, m2); lr_.backward()
    ex.encoder.encoder.dropout_masks.queue = [m1, m2]
    opt.zero_grad(); loss = ex.training_step((a.cuda(), b.cuda()), s); loss.backward()
    rp = dict(ref.named_parameters())
    worst = sorted(((rel_l2(p.grad.float().cpu(), rp[n].grad), n) for n, p in ex.named_parameters() if rp[n].grad.norm() > 1e-6), reverse=True)[:4]
    print(f"step {s} loss {float(loss):.6f} ref {float(lr_):.6f}  worst grad rel_l2:", [(f"{e:.1e}", n) for e, n in worst])
    OM.sgd_momentum_step(rparams, bufs, 0.03, 0.9, 1e-4); opt.step()
    worstw = sorted(((rel_l2(p.data.float().cpu(), rp[n].data), float((p.data.float().cpu() - rp[n].data).abs().max()), n) for n, p in ex.named_parameters()), reverse=True)[:3]
    print("        worst weight rel_l2 / maxabs:", [(f"{e:.1e}", f"{m:.1e}", n) for e, m, n in worstw])
