"""GPU: the transformer kernels of the AST / MAST encoder (attention, LayerNorm, GELU, patch unfold, AdamW) through the C ABI
against plain PyTorch fp32 references of the same ops on the same seeded inputs.  Tolerances are those of bf16 MFMA operands
with fp32 accumulation: outputs rel-L2 <= 1e-2, gradients rel-L2 <= 2e-2."""
import math

import numpy as np
import pytest
import torch

from oracle import fill
from helpers import rel_l2

pytestmark = pytest.mark.gpu


def _t(shape, salt, lo=-1.0, hi=1.0):
    return torch.from_numpy(fill.uniform(shape, salt, lo, hi))


@pytest.mark.parametrize("B,S,H", [(2, 108, 12), (3, 128, 2), (1, 37, 4), (2, 1, 1),
                                   (2, 129, 2), (1, 256, 3), (2, 300, 2), (1, 1212, 2)])
def test_attention_forward_backward(B, S, H):
    """S <= 128: one tile per (clip, head); longer (up to the 1,212 patches of a 10 s clip): 128-token blocks, two-sweep
    forward, dQ and dK/dV kernels walking the other axis - block boundaries inside (129, 300, 1212) and at the end (256)."""
    from src import _native as N
    C = H * 64
    qkv = _t((B * S, 3 * C), 100 + S, -1.5, 1.5).cuda().bfloat16()
    dout = _t((B * S, C), 200 + S).cuda().bfloat16()
    out = torch.empty(B * S, C, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B * H, S, dtype=torch.float32, device="cuda")
    dqkv = torch.full((B * S, 3 * C), float("nan"), dtype=torch.bfloat16, device="cuda")
    scale = 1.0 / math.sqrt(64)
    N.call("attn_fwd", qkv, out, lse, B, S, H, scale)
    N.call("attn_bwd", qkv, out, dout, lse, dqkv, B, S, H, scale)
    torch.cuda.synchronize()
    x = qkv.float().cpu().requires_grad_(True)
    q, k, v = (x[:, i * C:(i + 1) * C].view(B, S, H, 64).permute(0, 2, 1, 3) for i in range(3))
    s = (q @ k.transpose(-1, -2)) * scale
    ref = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * S, C)
    ref.backward(dout.float().cpu())
    assert rel_l2(out.float().cpu(), ref.detach()) < 1e-2
    np.testing.assert_allclose(lse.cpu().numpy(), torch.logsumexp(s, dim=-1).reshape(B * H, S).detach().numpy(), rtol=2e-3, atol=2e-3)
    for i, name in enumerate("qkv"):
        assert rel_l2(dqkv[:, i * C:(i + 1) * C].float().cpu(), x.grad[:, i * C:(i + 1) * C]) < 2e-2, name


def test_attention_long_backward_needs_forward_output():
    from src import _native as N
    z = torch.zeros(129 * 192, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError, match="EINVAL"):
        N.call("attn_bwd", z, None, z, torch.zeros(129, device="cuda"), z, 1, 129, 1, 0.125)


@pytest.mark.parametrize("M,C", [(216, 768), (5, 64), (130, 1024), (60, 96), (18, 192), (7, 40)])
def test_layernorm_forward_backward(M, C):
    from src import _native as N
    x = (_t((M, C), 300 + M, -2, 2) + 0.5).cuda()
    g = _t((C,), 301, 0.5, 1.5).cuda()
    b = _t((C,), 302, -0.2, 0.2).cuda()
    dy = _t((M, C), 303).cuda()
    y = torch.empty(M, C, dtype=torch.bfloat16, device="cuda")
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    y32 = torch.empty(M, C, device="cuda")
    N.call("layernorm_fwd", x, g, b, y, y32, mean, rstd, M, C, 1e-6)
    base = _t((M, C), 304).cuda()                       # the residual-stream gradient the LN gradient is added to
    dres = base.clone()
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    N.call("layernorm_bwd", dy, x, mean, rstd, g, dres, dg, db, M, C)
    torch.cuda.synchronize()
    xr, gr, br = x.cpu().requires_grad_(True), g.cpu().requires_grad_(True), b.cpu().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (C,), gr, br, 1e-6)
    ref.backward(dy.cpu())
    assert rel_l2(y.float().cpu(), ref.detach()) < 5e-3                       # bf16 output rounding
    assert rel_l2(y32.cpu(), ref.detach()) < 1e-6                             # the optional fp32 copy of the output
    np.testing.assert_allclose(mean.cpu().numpy(), xr.detach().mean(1).numpy(), rtol=1e-5, atol=1e-6)
    assert rel_l2((dres - base).cpu(), xr.grad) < 1e-5
    assert rel_l2(dg.cpu(), gr.grad) < 1e-5 and rel_l2(db.cpu(), br.grad) < 1e-5


def test_gelu_forward_backward():
    from src import _native as N
    n = 8 * 1000
    a = _t((n,), 400, -4, 4).cuda().bfloat16()
    dh = _t((n,), 401).cuda().bfloat16()
    h, da = torch.empty_like(a), torch.empty_like(a)
    N.call("gelu_fwd", a, h, n)
    N.call("gelu_bwd", a, dh, da, n)
    torch.cuda.synchronize()
    ar = a.float().cpu().requires_grad_(True)
    ref = torch.nn.functional.gelu(ar)
    ref.backward(dh.float().cpu())
    assert rel_l2(h.float().cpu(), ref.detach()) < 5e-3
    assert rel_l2(da.float().cpu(), ar.grad) < 5e-3


@pytest.mark.parametrize("F,T,fs,ts", [(128, 101, 10, 10), (64, 96, 16, 16)])
def test_patch_unfold_matches_conv2d(F, T, fs, ts):
    from src import _native as N
    B = 3
    x = _t((B, 1, F, T), 500 + F).cuda()
    nf, nt = (F - 16) // fs + 1, (T - 16) // ts + 1
    rows = torch.empty(B * nf * nt, 256, dtype=torch.bfloat16, device="cuda")
    N.call("patch_unfold", x, rows, B, F, T, fs, ts)
    torch.cuda.synchronize()
    w = _t((32, 1, 16, 16), 501)
    ref = torch.nn.functional.conv2d(x.cpu().bfloat16().float(), w, stride=(fs, ts)).flatten(2).transpose(1, 2).reshape(B * nf * nt, 32)
    got = rows.float().cpu() @ w.view(32, 256).T
    assert (nf, nt) == ((12, 9) if F == 128 else (4, 6))
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)


def test_adamw_matches_torch():
    from src import _native as N
    n = 4 * 1000 + 3
    p0, grads = _t((n,), 600), [_t((n,), 601 + i, -0.1, 0.1) for i in range(3)]
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref_p], lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    pad = (-n) % 4
    p = torch.nn.functional.pad(p0, (0, pad)).cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    for g in grads:
        ref_p.grad = g.clone()
        opt.step()
        step.add_(1)
        N.call("adamw", p, torch.nn.functional.pad(g, (0, pad)).cuda(), m, v, n, 3e-4, 0.9, 0.999, 1e-8, 0.05, 1.0, step)
    torch.cuda.synchronize()
    np.testing.assert_allclose(p[:n].cpu().numpy(), ref_p.detach().numpy(), rtol=2e-5, atol=1e-7)


# ------------------------------------------------------------------------------------------------ one block vs the reference
def test_transformer_block_vs_reference_multiscale_block(golden):
    """The HIP block launch sequence (LayerNorm, qkv GEMM, attention, proj + residual, LayerNorm, MLP + residual) against
    outputs of the reference's own `MultiScaleBlock` without pooling (`mvit/models/attention.py:304-393`, fixture
    `mvit_block.npz` / `plain`): output, input gradient and every parameter gradient.  bf16 MFMA operands: 1e-2 / 3e-2."""
    from oracle import mvit as OM
    from src import _native as N
    from src import vit_engine as VE
    g = golden("mvit_block")
    Pc, x, kw, gsalt = OM.golden_case("plain")
    C, H = kw["dim"], kw["heads"]
    B, Ntok = x.shape[0], x.shape[1]
    P = {k: v.cuda().contiguous() for k, v in Pc.items()}
    W = {k: v.bfloat16().contiguous() for k, v in P.items() if v.dim() == 2}
    X = x.reshape(B * Ntok, C).cuda().contiguous()
    Y, ctx = VE.block_forward(P, W, "", X, B, Ntok, C, H, 1e-6)
    torch.cuda.synchronize()
    want = torch.from_numpy(g["plain.y"]).reshape(B * Ntok, C)
    assert rel_l2(Y.cpu(), want) < 1e-2
    G = {k: torch.zeros_like(v) for k, v in P.items()}
    dX = torch.from_numpy(fill.uniform((B, Ntok, C), gsalt)).reshape(B * Ntok, C).cuda().contiguous()
    VE.block_backward(ctx, P, W, G, "", dX, B, Ntok, C, H)
    torch.cuda.synchronize()
    assert rel_l2(dX.cpu(), torch.from_numpy(g["plain.dx"]).reshape(B * Ntok, C)) < 3e-2
    for n, norm, head in zip(g["plain.g_names"], g["plain.g_norms"], g["plain.g_heads"]):
        gr = G[str(n)].cpu()
        assert abs(float(gr.norm()) - norm) <= 3e-2 * norm, (n, float(gr.norm()), norm)
        h = gr.flatten()[:8].numpy()
        assert np.abs(h - head[:h.size]).max() <= 3e-2 * max(np.abs(head).max(), norm / np.sqrt(gr.numel())) + 1e-6, n


@pytest.mark.parametrize("cname", ["plain", "pooled", "transition"])
def test_mvit_pooling_attention_block_vs_reference_multiscale_block(golden, cname):
    """The MViTv2 block launch sequence (`src/mvit_engine.py`: qkv GEMM, depthwise-conv pooling + LayerNorm of q / k / v,
    attention with decomposed relative positions and residual pooling, max-pooled / projected skip path, MLP with the width
    change) against the reference's own `MultiScaleBlock` (`mvit/models/attention.py:304-393`, fixture `mvit_block.npz`) in all
    three configurations: without pooling, pooled inside a stage, and the stage transition (width x2, query stride 2).
    Output 1e-2, input gradient and every parameter gradient 3e-2 (bf16 MFMA operands in the GEMMs)."""
    from oracle import mvit as OM
    from src import mvit_engine as ME
    g = golden("mvit_block")
    Pc, x, kw, gsalt = OM.golden_case(cname)
    cfg = ME.BlockCfg(hw=OM.GOLDEN_CONFIGS[cname]["hw"], **kw)
    B, L = x.shape[0], x.shape[1]
    P = {k: v.cuda().contiguous() for k, v in Pc.items()}
    W = {k: v.bfloat16().contiguous() for k, v in P.items() if v.dim() == 2 and "rel_pos" not in k}
    X = x.reshape(B * L, cfg.dim).cuda().contiguous()
    Y, hw_out, ctx = ME.block_forward(P, W, "", X, B, cfg)
    torch.cuda.synchronize()
    assert list(hw_out) == list(g[f"{cname}.hw_out"])
    want = torch.from_numpy(g[f"{cname}.y"])
    assert tuple(Y.shape) == (B * want.shape[1], want.shape[2])
    assert rel_l2(Y.cpu(), want.reshape(Y.shape)) < 1e-2
    G = {k: torch.zeros_like(v) for k, v in P.items()}
    dY = torch.from_numpy(fill.uniform(tuple(want.shape), gsalt)).reshape(Y.shape).cuda().contiguous()
    dX = ME.block_backward(ctx, P, W, G, "", dY, B, cfg)
    torch.cuda.synchronize()
    assert rel_l2(dX.cpu(), torch.from_numpy(g[f"{cname}.dx"]).reshape(B * L, cfg.dim)) < 3e-2
    for n, norm, head in zip(g[f"{cname}.g_names"], g[f"{cname}.g_norms"], g[f"{cname}.g_heads"]):
        gr = G[str(n)].cpu()
        if str(n) == "attn.norm_k.bias":
            # soft-max is invariant to a constant added to every key: this gradient is zero in exact arithmetic (3e-6 of rounding
            # residue in the reference's fp32 run against gradients of 10 - 1,000 everywhere else)
            assert float(gr.norm()) < 1e-3 and norm < 1e-4
            continue
        assert abs(float(gr.norm()) - norm) <= 3e-2 * norm, (n, float(gr.norm()), norm)
        h = gr.flatten()[:8].numpy()
        assert np.abs(h - head[:h.size]).max() <= 3e-2 * max(np.abs(head).max(), norm / np.sqrt(gr.numel())) + 1e-6, n


@pytest.mark.parametrize("d,qhw,khw,rel,residual", [(96, (12, 9), (3, 3), True, True), (64, (5, 7), (5, 7), True, False),
                                                    (128, (9, 8), (3, 2), False, True), (96, (2, 2), (3, 3), True, True)])
def test_mvit_attention_with_relative_positions_vs_torch(d, qhw, khw, rel, residual):
    """mvit_attn_fwd / _bwd alone against the fp32 torch statement of `MultiScaleAttention`'s core (`attention.py:262-288`,
    `cal_rel_pos_spatial` :44-90): head dims 64 / 96 / 128, more than 64 queries (two workgroups per head), query grids smaller
    and larger than the key grid, with and without tables / residual pooling.  fp32 kernels: 1e-4 (outputs are stored in bf16)."""
    from src import _native as N
    from src import mvit_engine as ME
    B, heads = 2, 3
    (qh, qw), (kh, kw) = qhw, khw
    Lq, Lk = qh * qw, kh * kw
    q = _t((B, heads, Lq, d), 900 + d).cuda()
    k = _t((B, heads, Lk, d), 901 + d).cuda()
    v = _t((B, heads, Lk, d), 902 + d).cuda()
    nr = 2 * max(qh, qw, kh, kw) - 1
    rh, rw = _t((nr, d), 903, -0.3, 0.3).cuda(), _t((nr, d), 904, -0.3, 0.3).cuda()
    ih, iw = ME.rel_index(qh, kh).cuda(), ME.rel_index(qw, kw).cuda()
    dout = _t((B * Lq, heads * d), 905).cuda().bfloat16()
    out = torch.empty(B * Lq, heads * d, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B * heads * Lq, device="cuda")
    scale = d ** -0.5
    tabs = (rh, rw, ih, iw) if rel else (None, None, None, None)
    N.call("mvit_attn_fwd", q, k, v, *tabs, out, lse, B, heads, d, qh, qw, kh, kw, nr if rel else 0, nr if rel else 0, int(residual), scale)
    dq = torch.full_like(q, float("nan"))
    dk, dv, drh, drw = torch.zeros_like(k), torch.zeros_like(v), torch.zeros_like(rh), torch.zeros_like(rw)
    N.call("mvit_attn_bwd", q, k, v, *tabs, dout, lse, dq, dk, dv, drh if rel else None, drw if rel else None, B, heads, d, qh, qw, kh, kw,
           nr if rel else 0, nr if rel else 0, int(residual), scale)
    torch.cuda.synchronize()
    qr, kr, vr, rhr, rwr = (t.double().cpu().requires_grad_(True) for t in (q, k, v, rh, rw))
    s = (qr * scale) @ kr.transpose(-1, -2)
    if rel:
        Rh, Rw = rhr[ih.cpu().long()], rwr[iw.cpu().long()]
        qg = qr.reshape(B, heads, qh, qw, d)
        s = (s.reshape(B, heads, qh, qw, kh, kw) + torch.einsum("bnhwc,hkc->bnhwk", qg, Rh)[..., :, None] +
             torch.einsum("bnhwc,wkc->bnhwk", qg, Rw)[..., None, :]).reshape(B, heads, Lq, Lk)
    y = torch.softmax(s, -1) @ vr
    if residual:
        y = y + qr
    ref = y.transpose(1, 2).reshape(B * Lq, heads * d)
    ref.backward(dout.double().cpu())
    assert rel_l2(out.double().cpu(), ref.detach()) < 4e-3                     # bf16 storage of the output
    np.testing.assert_allclose(lse.cpu().numpy(), torch.logsumexp(s, -1).reshape(-1).detach().numpy(), rtol=1e-4, atol=1e-4)
    for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        assert rel_l2(got.double().cpu(), want) < 2e-4, name
    if rel:
        assert rel_l2(drh.double().cpu(), rhr.grad) < 2e-4 and rel_l2(drw.double().cpu(), rwr.grad) < 2e-4


@pytest.mark.parametrize("d,hw,stride", [(96, (12, 9), (4, 4)), (64, (8, 8), (2, 2)), (96, (6, 5), (1, 1)), (128, (3, 3), (2, 2)), (96, (5, 4), ())])
def test_mvit_pooling_and_token_maxpool_vs_torch(d, hw, stride):
    """mvit_pool_fwd / _bwd (depthwise 3x3 conv over the token grid + LayerNorm per head; head split alone for stride ()) and
    tokpool_max_fwd / _bwd against torch (`attention_pool`, attention.py:12-41; the skip path's MaxPool2d, :343-350)."""
    from src import _native as N
    import torch.nn.functional as F
    B, heads = 2, 2
    H, W = hw
    L, att = H * W, heads * d
    qkv = _t((B * L, 3 * att), 950 + d).cuda().bfloat16()
    which = 1
    pooled = len(stride) > 0
    Ho, Wo = ((H - 1) // stride[0] + 1, (W - 1) // stride[1] + 1) if pooled else (H, W)
    w = _t((d, 1, 3, 3), 951, -0.5, 0.5).cuda()
    gam, bet = _t((d,), 952, 0.5, 1.5).cuda(), _t((d,), 953, -0.2, 0.2).cuda()
    out = torch.empty(B, heads, Ho * Wo, d, device="cuda")
    z, mean, rstd = torch.empty_like(out), torch.empty(B * heads * Ho * Wo, device="cuda"), torch.empty(B * heads * Ho * Wo, device="cuda")
    sh, sw = stride if pooled else (1, 1)
    N.call("mvit_pool_fwd", qkv, 3 * att, which * att, w if pooled else None, gam if pooled else None, bet if pooled else None, out,
           z if pooled else None, mean if pooled else None, rstd if pooled else None, B, heads, d, H, W, Ho, Wo, sh, sw, 1e-6)
    dout = _t((B, heads, Ho * Wo, d), 954).cuda()
    dqkv = torch.zeros(B * L, 3 * att, dtype=torch.bfloat16, device="cuda")
    dw, dg, db, dz = torch.zeros_like(w), torch.zeros_like(gam), torch.zeros_like(bet), torch.empty_like(out)
    N.call("mvit_pool_bwd", qkv, 3 * att, which * att, w if pooled else None, gam if pooled else None, dout, z if pooled else None,
           mean if pooled else None, rstd if pooled else None, dz if pooled else None, dw if pooled else None, dg if pooled else None,
           db if pooled else None, dqkv, B, heads, d, H, W, Ho, Wo, sh, sw)
    torch.cuda.synchronize()
    x = qkv.double().cpu().requires_grad_(True)
    wr, gr, br = (t.double().cpu().requires_grad_(True) for t in (w, gam, bet))
    t = x[:, which * att:(which + 1) * att].reshape(B, L, heads, d).permute(0, 2, 1, 3)
    if pooled:
        img = t.reshape(B * heads, H, W, d).permute(0, 3, 1, 2)
        img = F.conv2d(img, wr, None, stride=stride, padding=1, groups=d)
        t = F.layer_norm(img.reshape(B, heads, d, Ho * Wo).transpose(2, 3), (d,), gr, br, 1e-6)
    t.backward(dout.double().cpu())
    assert rel_l2(out.double().cpu(), t.detach()) < 1e-5
    got = dqkv[:, which * att:(which + 1) * att].double().cpu()
    assert rel_l2(got, x.grad[:, which * att:(which + 1) * att]) < 4e-3       # bf16 storage
    assert float(dqkv[:, :att].abs().max()) == 0 and float(dqkv[:, 2 * att:].abs().max()) == 0
    if pooled:
        assert rel_l2(dw.double().cpu(), wr.grad) < 1e-4 and rel_l2(dg.double().cpu(), gr.grad) < 1e-4 and rel_l2(db.double().cpu(), br.grad) < 1e-4
        # skip path: MaxPool2d(stride + 1 where stride > 1)
        ks = tuple(s + 1 if s > 1 else s for s in stride)
        C = 2 * d
        xs = _t((B, L, C), 960).cuda()
        xs[0, 0] = xs[0, 1]                                        # a tie: the first maximum wins, as in torch
        Lo = ((H + 2 * (ks[0] // 2) - ks[0]) // stride[0] + 1) * ((W + 2 * (ks[1] // 2) - ks[1]) // stride[1] + 1)
        ys, arg = torch.empty(B, Lo, C, device="cuda"), torch.empty(B, Lo, C, dtype=torch.uint8, device="cuda")
        N.call("tokpool_max_fwd", xs, ys, arg, B, H, W, C, ks[0], ks[1], stride[0], stride[1])
        dys = _t((B, Lo, C), 961).cuda()
        dxs = torch.empty_like(xs)
        N.call("tokpool_max_bwd", dys, arg, dxs, B, H, W, C, ks[0], ks[1], stride[0], stride[1])
        torch.cuda.synchronize()
        xr = xs.double().cpu().requires_grad_(True)
        yr = F.max_pool2d(xr.reshape(B, H, W, C).permute(0, 3, 1, 2), ks, stride, [k_ // 2 for k_ in ks])
        assert yr.shape[-2] * yr.shape[-1] == Lo
        yr = yr.reshape(B, C, Lo).transpose(1, 2)
        yr.backward(dys.double().cpu())
        assert torch.equal(ys.double().cpu(), yr.detach())
        assert rel_l2(dxs.double().cpu(), xr.grad) < 1e-6


@pytest.mark.parametrize("depth,B,T", [(24, 4, 101), (6, 2, 301)])
def test_mvit_encoder_forward_backward_vs_oracle(depth, B, T):
    """`ASTModel(model_size='mvit')` - the MViTv2-B layout of configs/MVITv2_B.yaml (96 -> 768 wide, 1 -> 8 heads of 96, 24 blocks,
    query stride 2 at blocks 2 / 5 / 21, adaptive kv stride from (4, 4)) on the 12 x 9 patch grid of a 1 s clip (and a shallower
    one on a 3 s clip) - against `oracle/mvit.py: mvit_encoder` built from the reference-pinned `multiscale_block`: embedding and
    EVERY parameter gradient."""
    from oracle import mvit as OM
    from helpers import views
    from src.encoder import ASTModel
    F_ = 128
    mv = {} if depth == 24 else dict(depth=depth, dim_mul=((1, 2.0), (3, 2.0)), head_mul=((1, 2.0), (3, 2.0)),
                                     q_strides=((1, 2, 2), (3, 2, 2)), kv_stride_adaptive=(4, 4))
    m = ASTModel(label_dim=256, input_fdim=F_, input_tdim=T, model_size="mvit", mvit=mv)
    fill.fill_state_dict_(m, seed=60 + depth)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "rel_pos" in n:
                p.copy_(_t(tuple(p.shape), fill.salt_of(n), -0.2, 0.2))
    P = {n: p.detach().clone().requires_grad_(True) for n, p in m.named_parameters()}
    blocks = [dict(dim=c.dim, dim_out=c.dim_out, heads=c.heads, hw=c.hw, stride_q=c.stride_q, stride_kv=c.stride_kv, rel_pos=c.rel_pos,
                   residual_pooling=c.residual_pooling, dim_mul_in_att=c.dim_mul_in_att) for c in m.cfg["blocks"]]
    x = torch.cat([views(B, T, 720 + i) for i in range(2)], dim=2).contiguous()
    assert x.shape == (B, 1, F_, T)
    dout = _t((B, 256), 722, -1, 1)
    out_ref = OM.mvit_encoder(P, x, blocks)
    out_ref.backward(dout)
    # the bar for the gradients, as for the DeLoRes-M step: bf16 operands through 24 blocks and three
    # arg-max poolings cost what they cost - the restatement of the REFERENCE under torch.autocast(bfloat16) is 8 % off its own
    # fp32 gradients at the median parameter.  Every HIP gradient must be within 6e-2 of the fp32 oracle, or at least as close
    # to it as that autocast run (10 % slack).
    Pa = {n: p.detach().clone().requires_grad_(True) for n, p in m.named_parameters()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out_ac = OM.mvit_encoder(Pa, x, blocks)
    out_ac.float().backward(dout)
    m = m.cuda().train()
    out = m(x.cuda())
    out.backward(dout.cuda())
    torch.cuda.synchronize()
    assert rel_l2(out.detach().float().cpu(), out_ref.detach()) < 2e-2
    checked = 0
    for n, p in m.named_parameters():
        if n.endswith("attn.norm_k.bias"):                      # zero in exact arithmetic (soft-max ignores a constant on the keys)
            assert float(p.grad.abs().max()) < 1e-3 * float(dict(m.named_parameters())[n.replace("norm_k.bias", "norm_v.bias")].grad.abs().max())
            continue
        e = rel_l2(p.grad.float().cpu(), P[n].grad)
        bar = max(6e-2, 1.1 * rel_l2(Pa[n].grad.float(), P[n].grad))
        assert e < bar, (n, e, bar)
        checked += 1
    assert checked > 100
    with torch.no_grad():
        out2 = m(x.cuda())
    assert rel_l2(out2.float().cpu(), out.detach().float().cpu()) < 1e-6


# ------------------------------------------------------------------------------------------------ whole encoder
@pytest.mark.parametrize("depth,B,F,T", [(2, 4, 128, 101), (12, 2, 128, 101), (2, 8, 64, 96), (2, 2, 128, 1001)])
def test_ast_encoder_forward_backward_vs_oracle(depth, B, F, T):
    """ASTModel (HIP launch sequence) against the CPU restatement: embedding and EVERY parameter gradient."""
    from oracle import vit as OV
    from helpers import views
    from src.encoder import ASTModel
    kw = dict(label_dim=256, fstride=10, tstride=10, input_fdim=F, input_tdim=T, depth=depth)
    ref = OV.ASTModel(**kw)
    fill.fill_state_dict_(ref, seed=40 + depth)
    with torch.no_grad():
        ref.v.pos_embed.copy_(_t(tuple(ref.v.pos_embed.shape), 41, -0.05, 0.05))
    m = ASTModel(**kw)
    assert [k for k, _ in m.named_parameters()] == [k for k, _ in ref.named_parameters()]
    m.load_state_dict(ref.state_dict())
    m = m.cuda().train()
    x = torch.cat([views(B, T, 700 + i)[:, :, :min(F, 64)] for i in range((F + 63) // 64)], dim=2)[:, :, :F].contiguous()
    assert x.shape == (B, 1, F, T)
    dout = _t((B, 256), 702, -1, 1)
    out_ref = ref(x)
    out_ref.backward(dout)
    out = m(x.cuda())
    out.backward(dout.cuda())
    torch.cuda.synchronize()
    assert rel_l2(out.detach().float().cpu(), out_ref.detach()) < 2e-2
    worst = 0.0
    for (n, p), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        e = rel_l2(p.grad.float().cpu(), pr.grad)
        worst = max(worst, e)
        assert e < 6e-2, (n, e)
    assert worst > 0.0
    # inference path (no autograd) gives the same embedding
    with torch.no_grad():
        out2 = m(x.cuda())
    assert rel_l2(out2.float().cpu(), out.detach().float().cpu()) < 1e-6


# ------------------------------------------------------------------------------------------------ SS-MAST expert
MAST_CFG = {"run": {"batch_size": 8, "precision": "bf16"},
            "pretrain": {"base_encoder": {"type": "MAST", "output_dim": 768, "depth": 2, "num_heads": 12, "fstride": 10, "tstride": 10,
                                          "return_all_layers": False},
                         "normalization": "mean_var",
                         "input": {"type": "raw_wav", "sampling_rate": 16000, "length_wave": 1.0, "n_mels": 128}}}


def _mel128(B, T, salt):
    from helpers import views
    return torch.cat([views(B, T, salt), views(B, T, salt + 5000)], dim=2).contiguous()          # [B, 1, 128, T]


def test_ssmast_step_vs_oracle():
    """Symmetric MoCo on the transformer: both cross-entropies, every gradient of the step, queue, EMA of the key encoder."""
    import copy
    from oracle import vit as OV
    from helpers import closed_queue
    from src.upstream.ssmast.upstream_expert import Upstream_Expert, adjust_moco_momentum
    B, T, K = 8, 101, 64
    ref = OV.SSMastExpert(emb_dim=256, num_negatives=K, input_fdim=128, input_tdim=T, depth=2)
    fill.fill_state_dict_(ref, seed=50)
    with torch.no_grad():
        ref.encoder_q.v.pos_embed.copy_(_t(tuple(ref.encoder_q.v.pos_embed.shape), 51, -0.05, 0.05))
        for pq, pk in zip(ref.encoder_q.parameters(), ref.encoder_k.parameters()):
            pk.copy_(pq)
        ref.queue.copy_(closed_queue(256, K))
    ex = Upstream_Expert(copy.deepcopy(MAST_CFG), num_negatives=K)
    ex.load_state_dict(ref.state_dict())
    ex = ex.cuda().train()
    a, b = _mel128(B, T, 9700), _mel128(B, T, 9701)
    loss_ref = ref.training_loss(a, b, epoch=3)
    loss_ref.backward()
    ex.current_epoch = 3
    opt = ex.configure_optimizers()
    wk0 = ex.encoder_k.fc.weight.detach().clone()
    opt.zero_grad()
    loss = ex.training_step((a.cuda(), b.cuda()), 0)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 2e-2 * float(loss_ref)
    for (n, p), (_, pr) in zip(ex.encoder_q.named_parameters(), ref.encoder_q.named_parameters()):
        assert rel_l2(p.grad.float().cpu(), pr.grad) < 8e-2, n
    assert int(ex.queue_ptr[0]) == int(ref.queue_ptr[0]) == (2 * B) % K
    assert rel_l2(ex.queue.cpu(), ref.queue) < 2e-2
    # key encoder: two EMA updates with m(epoch + 1), no gradient
    m = adjust_moco_momentum(4)
    assert 0.99 < m < 1.0
    want = wk0.cpu() * m * m + ex.encoder_q.fc.weight.detach().cpu() * (1 - m) * (1 + m)
    np.testing.assert_allclose(ex.encoder_k.fc.weight.detach().cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-7)
    assert rel_l2(ex.encoder_k.fc.weight.detach().cpu(), ref.encoder_k.fc.weight.detach()) < 1e-5
    w0 = ex.encoder_q.fc.weight.detach().clone()
    opt.step()                                                              # AdamW over the flat buffer
    assert int(opt.step_count[0]) == 1 and not torch.equal(ex.encoder_q.fc.weight.detach(), w0)
    assert float((ex.encoder_q.fc.weight.detach() - w0).abs().max()) <= 3e-4 * 1.001 + 1e-9      # |first Adam update| <= lr


def test_ssmast_graph_replay_matches_eager():
    """Graph-replayed steps (AdamW step counter, queue pointer and EMA inside the graph) against an eagerly stepped twin."""
    import copy
    from src.upstream.ssmast.upstream_expert import Upstream_Expert
    twins = []
    for _ in range(2):
        torch.manual_seed(0)
        twins.append(Upstream_Expert(copy.deepcopy(MAST_CFG), num_negatives=64).cuda().train())
    eager, graphed = twins
    opt_e, opt_g = eager.configure_optimizers(), graphed.configure_optimizers()
    step = graphed.graphed_step(opt_g, eager_steps=1)
    le, lg = [], []
    for s in range(4):
        a, b = _mel128(8, 101, 9800 + 2 * s).cuda(), _mel128(8, 101, 9801 + 2 * s).cuda()
        opt_e.zero_grad()
        loss = eager.training_step((a, b), s)
        loss.backward()
        opt_e.step()
        le.append(float(loss))
        lg.append(float(step(a, b)))
    assert step.replays == 3 and all(np.isfinite(lg))
    np.testing.assert_allclose(lg, le, rtol=2e-2)
    assert int(opt_g.step_count[0]) == int(opt_e.step_count[0]) == 4
    assert int(graphed.queue_ptr[0]) == int(eager.queue_ptr[0]) == (4 * 16) % 64
