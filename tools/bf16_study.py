"""CPU numerics study: which bf16 roundings of the Barlow projector chain cost gradient accuracy?
Emulates the engine's projector forward/backward (src/engine.py projector_forward/backward) in float64 with a bf16
rounding switched on at one class of tensors at a time; inputs are the pooled features of the oracle encoder."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd"), os.path.join(ROOT, "tests")]
import torch
from oracle import fill, model as OM
from helpers import views
import conftest

torch.set_num_threads(8)
B, T = int(os.environ.get("B", 64)), 96
ref = OM.DeloresMExpert(copy.deepcopy(conftest.CFG_M), num_negatives=256)
fill.fill_state_dict_(ref, seed=9)
ref.train()
with torch.no_grad():
    _, q1, q2, q3 = ref.encoder_q(views(B, T, 8800), None)
    _, k1, k2, k3 = ref.encoder_k(views(B, T, 8801), None)
print("pooled q1: |col mean| / col std median", float((q1.mean(0).abs() / q1.std(0).clamp_min(1e-9)).median()))


def Q(x, on):
    return x.float().bfloat16().double() if on else x


def bn_fwd(a, gamma, beta):
    mu, var = a.mean(0), a.var(0, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    xhat = (a - mu) * rstd
    return xhat * gamma + beta, xhat, rstd


def bn_bwd(g, xhat, rstd, gamma):
    return gamma * rstd * (g - g.mean(0) - xhat * (g * xhat).mean(0))


def run(P, y1, y2, t):
    """t: set of switched-on roundings"""
    W = [Q(P[f"projector.{i}.weight"].double(), "w" in t) for i in (0, 3, 6)]
    g1, b1 = P["projector.1.weight"].double(), P["projector.1.bias"].double()
    g4, b4 = P["projector.4.weight"].double(), P["projector.4.bias"].double()
    outs = []
    ctx = []
    for y in (y1, y2):
        y = y.double()
        yc = Q(y - y.mean(0) if "center" in t else y, "in" in t)
        a1 = Q(yc @ W[0].T, "a" in t)
        o1, xh1, rs1 = bn_fwd(a1, g1, b1); h1 = Q(o1.clamp_min(0), "h" in t)
        a2 = Q(h1 @ W[1].T, "a" in t)
        o2, xh2, rs2 = bn_fwd(a2, g4, b4); h2 = Q(o2.clamp_min(0), "h" in t)
        z = Q(h2 @ W[2].T, "a" in t)
        o0, xh0, rs0 = bn_fwd(z, 1.0, 0.0); zn = Q(o0, "h" in t)
        ctx.append((yc, a1, o1, xh1, rs1, h1, a2, o2, xh2, rs2, h2, z, xh0, rs0, zn))
    Bn = y1.shape[0]
    zn1, zn2 = ctx[0][-1], ctx[1][-1]
    c = zn1.T @ zn2 / Bn
    coef = 5e-5 / 32
    loss = coef * ((c - torch.eye(c.shape[0], dtype=c.dtype)) ** 2).sum()
    dc = Q(2 * coef / Bn * (c - torch.eye(c.shape[0], dtype=c.dtype)), "dc" in t)
    dzn = [zn2 @ dc.T, zn1 @ dc]
    dW = [0, 0, 0]
    dY = None
    for v, (yc, a1, o1, xh1, rs1, h1, a2, o2, xh2, rs2, h2, z, xh0, rs0, zn) in enumerate(ctx):
        dz = Q(bn_bwd(dzn[v], xh0, rs0, 1.0), "g" in t)
        dW[2] = dW[2] + dz.T @ h2
        dh2 = dz @ W[2]
        da2 = Q(bn_bwd(dh2 * (o2 > 0), xh2, rs2, g4), "g" in t)
        dW[1] = dW[1] + da2.T @ h1
        dh1 = da2 @ W[1]
        da1 = Q(bn_bwd(dh1 * (o1 > 0), xh1, rs1, g1), "g" in t)
        dW[0] = dW[0] + da1.T @ yc
        if v == 0:
            dY = da1 @ W[0]
    return float(loss), dW, dY


def rel(a, b):
    return float((a - b).norm() / b.norm())


for name, P, y1, y2 in (("p1", ref.p1, q1, k1), ("p3", ref.p3, q3, k3)):
    P = dict(P.named_parameters())
    l0, dW0, dY0 = run(P, y1, y2, set())
    print(f"--- {name}: loss {l0:.6f}")
    for t in ({"in"}, {"in", "center"}, {"w"}, {"a"}, {"h"}, {"dc"}, {"g"}, {"in", "center", "w", "h", "dc", "g"},
              {"in", "center", "w", "a", "h", "dc", "g"}, {"in", "center", "w", "h", "dc"}, {"in", "center", "w", "h", "g"},
              {"in", "center", "w", "dc", "g"}, {"in", "center", "h", "dc", "g"}):
        l, dW, dY = run(P, y1, y2, t)
        print(f"  {'+'.join(sorted(t)):28s} loss rel {abs(l - l0) / l0:.1e}  dW0 {rel(dW[0], dW0[0]):.3f} dW3 {rel(dW[1], dW0[1]):.3f} "
              f"dW6 {rel(dW[2], dW0[2]):.3f} dY {rel(dY, dY0):.3f}")
