"""Launch sequences of the MViTv2 pooling-attention block and of the MViTv2 encoder on one GPU.

The block is `MultiScaleBlock` of `extras/mast_new/mast/mvit/models/attention.py:304-393` (attention `:93-302`) in the form the
shipped configurations use (`configs/MVITv2_B.yaml`: mode "conv", pool_first False, no class token, relative positions, residual
pooling, width change inside the attention); the stage layout is `mvit/models/mvit_model.py:165-220, 280-317`.  Per block:

  layernorm -> GEMM qkv -> mvit_pool (q, k, v: depthwise 3x3 + LayerNorm_d per head, or the head split alone)
            -> mvit_attn (scores + decomposed relative positions + softmax + PV + residual pooling) -> GEMM proj (+bias +skip)
     skip  =  x, or GEMM proj(LN1 x) when the width changes inside the attention, max-pooled when the queries are strided
  layernorm -> GEMM fc1 -> GELU -> GEMM fc2 (+bias + x, or + GEMM proj(LN2 x) when the width changes after the attention)

Conventions are those of `vit_engine`: fp32 residual stream, bf16 MFMA operands, fp32 gradients into every LayerNorm backward,
parameter gradients ACCUMULATED into the tensors handed in.  The pooled q / k / v and their gradients are fp32.
"""
import math

import torch

from src import _native as N
from src import engine as E
from src.vit_engine import ViTCtx, _cast, _ln, _wgrad

BF = N.BF16


def rel_index(nq, nk):
    """Row of the relative-position table for (query coordinate, key coordinate): the arithmetic of `cal_rel_pos_spatial`
    (`attention.py:59-71`, float32 distances truncated by `.long()`), int32 [nq, nk]."""
    rq, rk = max(nk / nq, 1.0), max(nq / nk, 1.0)
    dist = torch.arange(nq)[:, None] * rq - torch.arange(nk)[None, :] * rk
    dist += (nk - 1) * rk
    return dist.long().to(torch.int32)


def pooled_hw(hw, stride):
    """Output grid of the 3x3 / pad 1 pooling convolution (and of the skip path's MaxPool2d(s + 1, s, pad s // 2 ... ))."""
    return ((hw[0] - 1) // stride[0] + 1, (hw[1] - 1) // stride[1] + 1) if len(stride) else tuple(hw)


class BlockCfg:
    def __init__(self, dim, dim_out, heads, hw, stride_q=(), stride_kv=(), rel_pos=False, residual_pooling=True,
                 dim_mul_in_att=False, eps=1e-6, **_):
        self.dim, self.dim_out, self.heads, self.hw = dim, dim_out, heads, tuple(hw)
        self.stride_q, self.stride_kv = tuple(stride_q), tuple(stride_kv)
        self.rel_pos, self.residual_pooling, self.dim_mul_in_att, self.eps = rel_pos, residual_pooling, dim_mul_in_att, eps
        self.att = dim_out if dim_mul_in_att else dim
        self.d = self.att // heads
        self.q_hw, self.k_hw = pooled_hw(hw, self.stride_q), pooled_hw(hw, self.stride_kv)
        self._idx, self._checked = {}, set()

    def index_tables(self, device):
        key = str(device)
        if key not in self._idx:
            self._idx[key] = (rel_index(self.q_hw[0], self.k_hw[0]).to(device).contiguous(),
                              rel_index(self.q_hw[1], self.k_hw[1]).to(device).contiguous())
        return self._idx[key]


def _pool(QKV, which, cfg, P, p, B, stride, name):
    """One of q / k / v: [B*L, 3*att] bf16 -> (pooled fp32 [B, heads, Lo, d], ctx for the backward)."""
    H, W = cfg.hw
    Ho, Wo = pooled_hw(cfg.hw, stride)
    dev = QKV.device
    out = torch.empty(B, cfg.heads, Ho * Wo, cfg.d, dtype=torch.float32, device=dev)
    k = ViTCtx()
    k.which, k.stride, k.Ho, k.Wo, k.name = which, stride, Ho, Wo, name
    if len(stride):
        k.z = torch.empty_like(out)
        k.mean = torch.empty(B * cfg.heads * Ho * Wo, dtype=torch.float32, device=dev)
        k.rstd = torch.empty_like(k.mean)
        N.call("mvit_pool_fwd", QKV, 3 * cfg.att, which * cfg.att, P[p + f"attn.pool_{name}.weight"], P[p + f"attn.norm_{name}.weight"],
               P[p + f"attn.norm_{name}.bias"], out, k.z, k.mean, k.rstd, B, cfg.heads, cfg.d, H, W, Ho, Wo, stride[0], stride[1], cfg.eps)
    else:
        N.call("mvit_pool_fwd", QKV, 3 * cfg.att, which * cfg.att, None, None, None, out, None, None, None, B, cfg.heads, cfg.d, H, W,
               Ho, Wo, 1, 1, cfg.eps)
    return out, k


def _pool_bwd(k, dOut, QKV, dQKV, cfg, P, G, p, B):
    H, W = cfg.hw
    name = k.name
    if len(k.stride):
        dz = torch.empty_like(dOut)
        N.call("mvit_pool_bwd", QKV, 3 * cfg.att, k.which * cfg.att, P[p + f"attn.pool_{name}.weight"], P[p + f"attn.norm_{name}.weight"],
               dOut, k.z, k.mean, k.rstd, dz, G[p + f"attn.pool_{name}.weight"], G[p + f"attn.norm_{name}.weight"],
               G[p + f"attn.norm_{name}.bias"], dQKV, B, cfg.heads, cfg.d, H, W, k.Ho, k.Wo, k.stride[0], k.stride[1])
    else:
        N.call("mvit_pool_bwd", QKV, 3 * cfg.att, k.which * cfg.att, None, None, dOut, None, None, None, None, None, None, None, dQKV,
               B, cfg.heads, cfg.d, H, W, k.Ho, k.Wo, 1, 1)


def _skip_kernel(stride_q):
    return tuple(s + 1 if s > 1 else s for s in stride_q)


def block_forward(P, W, p, X, B, cfg):
    """X fp32 [B*L, dim] -> (X' fp32 [B*Lq, dim_out], (qh, qw), ctx).  P: fp32 parameters, W: bf16 copies of the 2-D weights,
    both keyed `p + <name of the reference block's parameter>`."""
    H, Wd = cfg.hw
    L, dim, att, dout = H * Wd, cfg.dim, cfg.att, cfg.dim_out
    M, dev = B * L, X.device
    k = ViTCtx()
    k.X1 = X
    widen_in = cfg.dim_mul_in_att and dim != dout
    # a block that widens inside the attention max-pools proj(LN1 x) on its skip path: that projection runs on the exact-fp32
    # MFMA path from an fp32 copy of the LayerNorm output, because the ARG-MAX of the pooling decides where the gradient goes -
    # with bf16 operands ~2 % of the windows pick another token and the block's input gradient is 5 % off the reference's
    Y1f = torch.empty(M, dim, dtype=torch.float32, device=dev) if widen_in else None
    k.Y1, k.mu1, k.rs1 = _ln(X, P[p + "norm1.weight"], P[p + "norm1.bias"], M, dim, cfg.eps, y32=Y1f)
    k.QKV = torch.empty(M, 3 * att, dtype=torch.bfloat16, device=dev)
    E.gemm(BF, 0, 0, M, 3 * att, dim, k.Y1, dim, W[p + "attn.qkv.weight"], dim, k.QKV, 3 * att, bias=P[p + "attn.qkv.bias"])
    k.q, k.cq = _pool(k.QKV, 0, cfg, P, p, B, cfg.stride_q, "q")
    k.k, k.ck = _pool(k.QKV, 1, cfg, P, p, B, cfg.stride_kv, "k")
    k.v, k.cv = _pool(k.QKV, 2, cfg, P, p, B, cfg.stride_kv, "v")
    (qh, qw), (kh, kw) = cfg.q_hw, cfg.k_hw
    Lq = qh * qw
    Mq = B * Lq
    k.A = torch.empty(Mq, att, dtype=torch.bfloat16, device=dev)
    k.lse = torch.empty(B * cfg.heads * Lq, dtype=torch.float32, device=dev)
    rh = rw = ih = iw = None
    nrh = nrw = 0
    if cfg.rel_pos:
        rh, rw = P[p + "attn.rel_pos_h"], P[p + "attn.rel_pos_w"]
        ih, iw = cfg.index_tables(dev)
        nrh, nrw = rh.shape[0], rw.shape[0]
        if (nrh, nrw) not in cfg._checked:                   # host-side check of the table heights, once per shape (it synchronises)
            cfg._checked.add((nrh, nrw))
            ok = int(ih.max()) < nrh and int(iw.max()) < nrw
        else:
            ok = True
        if not ok:
            raise ValueError(f"{p}: relative-position tables with {nrh} / {nrw} rows are too short for grids {cfg.q_hw} x {cfg.k_hw}")
    k.rel = (rh, rw, ih, iw, nrh, nrw)
    scale = cfg.d ** -0.5
    N.call("mvit_attn_fwd", k.q, k.k, k.v, rh, rw, ih, iw, k.A, k.lse, B, cfg.heads, cfg.d, qh, qw, kh, kw, nrh, nrw,
           int(cfg.residual_pooling), scale)
    # ---- skip path
    if widen_in:
        S = torch.empty(M, dout, dtype=torch.float32, device=dev)
        E.gemm(N.F32, 0, 0, M, dout, dim, Y1f, dim, P[p + "proj.weight"], dim, S, dout, bias=P[p + "proj.bias"], out_f32=1)
    else:
        S = X
    k.pooled_skip = len(cfg.stride_q) > 0 and math.prod(cfg.stride_q) > 1
    if k.pooled_skip:
        ks = _skip_kernel(cfg.stride_q)
        C = S.shape[1]
        Sp = torch.empty(Mq, C, dtype=torch.float32, device=dev)
        k.arg = torch.empty(Mq, C, dtype=torch.uint8, device=dev)
        N.call("tokpool_max_fwd", S, Sp, k.arg, B, H, Wd, C, ks[0], ks[1], cfg.stride_q[0], cfg.stride_q[1])
        S = Sp
    X2 = torch.empty(Mq, att, dtype=torch.float32, device=dev)
    E.gemm(BF, 0, 0, Mq, att, att, k.A, att, W[p + "attn.proj.weight"], att, X2, att, bias=P[p + "attn.proj.bias"], out_f32=1,
           resid=S, ldr=att)
    k.X2 = X2
    k.Y2, k.mu2, k.rs2 = _ln(X2, P[p + "norm2.weight"], P[p + "norm2.bias"], Mq, att, cfg.eps)
    Hd = W[p + "mlp.fc1.weight"].shape[0]
    k.A1 = torch.empty(Mq, Hd, dtype=torch.bfloat16, device=dev)
    E.gemm(BF, 0, 0, Mq, Hd, att, k.Y2, att, W[p + "mlp.fc1.weight"], att, k.A1, Hd, bias=P[p + "mlp.fc1.bias"])
    k.H1 = torch.empty(Mq, Hd, dtype=torch.bfloat16, device=dev)
    N.call("gelu_fwd", k.A1, k.H1, Mq * Hd)
    k.widen_out = (not cfg.dim_mul_in_att) and dim != dout
    if k.widen_out:
        R = torch.empty(Mq, dout, dtype=torch.float32, device=dev)
        E.gemm(BF, 0, 0, Mq, dout, att, k.Y2, att, W[p + "proj.weight"], att, R, dout, bias=P[p + "proj.bias"], out_f32=1)
    else:
        R = X2
    X3 = torch.empty(Mq, dout, dtype=torch.float32, device=dev)
    E.gemm(BF, 0, 0, Mq, dout, Hd, k.H1, Hd, W[p + "mlp.fc2.weight"], Hd, X3, dout, bias=P[p + "mlp.fc2.bias"], out_f32=1,
           resid=R, ldr=dout)
    return X3, (qh, qw), k


def block_backward(k, P, W, G, p, dX3, B, cfg):
    """dX3 fp32 [B*Lq, dim_out] (gradient of the block's output; consumed) -> gradient of the block's input fp32 [B*L, dim];
    parameter gradients are accumulated into G."""
    H, Wd = cfg.hw
    L, dim, att, dout = H * Wd, cfg.dim, cfg.att, cfg.dim_out
    (qh, qw), (kh, kw) = cfg.q_hw, cfg.k_hw
    Lq, Lk = qh * qw, kh * kw
    M, Mq, dev = B * L, B * Lq, dX3.device
    Hd = W[p + "mlp.fc1.weight"].shape[0]
    # ---- MLP branch: X3 = R + fc2(gelu(fc1(Y2))),  R = X2 or proj(Y2)
    dXb = _cast(dX3, Mq, dout)
    _wgrad(dXb, k.H1, G[p + "mlp.fc2.weight"], Mq, dout, Hd)
    E.colsum_add(N.F32, dX3, Mq, dout, G[p + "mlp.fc2.bias"])
    dH1 = E.linear_bwd_x(BF, dXb, W[p + "mlp.fc2.weight"], Mq, dout, Hd)
    dA1 = torch.empty(Mq, Hd, dtype=torch.bfloat16, device=dev)
    N.call("gelu_bwd", k.A1, dH1, dA1, Mq * Hd)
    _wgrad(dA1, k.Y2, G[p + "mlp.fc1.weight"], Mq, Hd, att)
    E.colsum_add(BF, dA1, Mq, Hd, G[p + "mlp.fc1.bias"])
    dY2 = E.linear_bwd_x(BF, dA1, W[p + "mlp.fc1.weight"], Mq, Hd, att, out_f32=1)
    if k.widen_out:
        _wgrad(dXb, k.Y2, G[p + "proj.weight"], Mq, dout, att)
        E.colsum_add(N.F32, dX3, Mq, dout, G[p + "proj.bias"])
        dY2 += E.linear_bwd_x(BF, dXb, W[p + "proj.weight"], Mq, dout, att, out_f32=1)
        dX2 = torch.zeros(Mq, att, dtype=torch.float32, device=dev)
    else:
        dX2 = dX3                                           # the residual connection: the gradient passes through
    N.call("layernorm_bwd", dY2, k.X2, k.mu2, k.rs2, P[p + "norm2.weight"], dX2, G[p + "norm2.weight"], G[p + "norm2.bias"], Mq, att)
    # ---- attention branch: X2 = skip + proj(A)
    dXb = _cast(dX2, Mq, att)
    _wgrad(dXb, k.A, G[p + "attn.proj.weight"], Mq, att, att)
    E.colsum_add(N.F32, dX2, Mq, att, G[p + "attn.proj.bias"])
    dA = E.linear_bwd_x(BF, dXb, W[p + "attn.proj.weight"], Mq, att, att)                   # bf16 [Mq, att]
    rh, rw, ih, iw, nrh, nrw = k.rel
    dq = torch.empty_like(k.q)
    dk, dv = torch.zeros_like(k.k), torch.zeros_like(k.v)
    N.call("mvit_attn_bwd", k.q, k.k, k.v, rh, rw, ih, iw, dA, k.lse, dq, dk, dv,
           G[p + "attn.rel_pos_h"] if cfg.rel_pos else None, G[p + "attn.rel_pos_w"] if cfg.rel_pos else None,
           B, cfg.heads, cfg.d, qh, qw, kh, kw, nrh, nrw, int(cfg.residual_pooling), cfg.d ** -0.5)
    dQKV = torch.empty(M, 3 * att, dtype=torch.bfloat16, device=dev)
    _pool_bwd(k.cq, dq, k.QKV, dQKV, cfg, P, G, p, B)
    _pool_bwd(k.ck, dk, k.QKV, dQKV, cfg, P, G, p, B)
    _pool_bwd(k.cv, dv, k.QKV, dQKV, cfg, P, G, p, B)
    _wgrad(dQKV, k.Y1, G[p + "attn.qkv.weight"], M, 3 * att, dim)
    E.colsum_add(BF, dQKV, M, 3 * att, G[p + "attn.qkv.bias"])
    dY1 = E.linear_bwd_x(BF, dQKV, W[p + "attn.qkv.weight"], M, 3 * att, dim, out_f32=1)
    # ---- skip path: dS = dX2 (through the max pooling, through proj(Y1) when the width changes inside the attention)
    dS = dX2
    if k.pooled_skip:
        ks = _skip_kernel(cfg.stride_q)
        C = dS.shape[1]
        dSf = torch.empty(M, C, dtype=torch.float32, device=dev)
        N.call("tokpool_max_bwd", dS, k.arg, dSf, B, H, Wd, C, ks[0], ks[1], cfg.stride_q[0], cfg.stride_q[1])
        dS = dSf
    if cfg.dim_mul_in_att and dim != dout:
        dSb = _cast(dS, M, dout)
        _wgrad(dSb, k.Y1, G[p + "proj.weight"], M, dout, dim)
        E.colsum_add(N.F32, dS, M, dout, G[p + "proj.bias"])
        dY1 += E.linear_bwd_x(BF, dSb, W[p + "proj.weight"], M, dout, dim, out_f32=1)
        dX = torch.zeros(M, dim, dtype=torch.float32, device=dev)
    else:
        dX = dS
    N.call("layernorm_bwd", dY1, k.X1, k.mu1, k.rs1, P[p + "norm1.weight"], dX, G[p + "norm1.weight"], G[p + "norm1.bias"], M, dim)
    return dX


# ------------------------------------------------------------------------------------------------------------ the encoder
def stage_layout(hw, embed_dim=96, depth=24, num_heads=1, dim_mul=((2, 2.0), (5, 2.0), (21, 2.0)), head_mul=((2, 2.0), (5, 2.0), (21, 2.0)),
                 q_strides=((2, 2, 2), (5, 2, 2), (21, 2, 2)), kv_stride_adaptive=(4, 4), pool_all_q=True, rel_pos=True,
                 residual_pooling=True, dim_mul_in_att=True, eps=1e-6):
    """Per-block configurations of MViTv2 as `MViT.__init__` + `_prepare_mvit_configs` derive them (`mvit_model.py:165-220,
    280-317`; defaults = configs/MVITv2_B.yaml: every block pools q with stride 1 except the listed transitions, kv stride
    adaptive from (4, 4), width / heads x2 at blocks 2, 5, 21, width change inside the attention).  hw = grid of the patch
    embedding; every block's grid is what the previous block's query pooling really produced."""
    dm, hm = {int(i): float(m) for i, m in dim_mul}, {int(i): float(m) for i, m in head_mul}
    sq = {int(i): (int(a), int(b)) for i, a, b in q_strides}
    skv = tuple(kv_stride_adaptive)
    cfgs, dim, heads, hw = [], embed_dim, num_heads, tuple(hw)
    for i in range(depth):
        heads = int(round(heads * hm.get(i, 1.0)))
        stride_q = sq.get(i, (1, 1) if pool_all_q else ())
        if i in sq:
            skv = tuple(max(s // q, 1) for s, q in zip(skv, sq[i]))
        if dim_mul_in_att:
            dim_out = int(round(dim * dm.get(i, 1.0)))
        else:
            dim_out = int(round(dim * dm.get(i + 1, 1.0)))
        c = BlockCfg(dim, dim_out, heads, hw, stride_q=stride_q, stride_kv=skv, rel_pos=rel_pos, residual_pooling=residual_pooling,
                     dim_mul_in_att=dim_mul_in_att, eps=eps)
        cfgs.append(c)
        hw, dim = c.q_hw, dim_out
    return cfgs


def mvit_forward(P, W, x, cfg, need_ctx=True):
    """x [B, 1, F, T] fp32 log-mel -> (embedding [B, out_dim] fp32, ctx).  cfg: dict(blocks=[BlockCfg...], embed_dim, fstride,
    tstride, final_norm, eps)."""
    B, _, F, T = x.shape
    C0, fs, ts = cfg["embed_dim"], cfg["fstride"], cfg["tstride"]
    nf, nt = (F - 16) // fs + 1, (T - 16) // ts + 1
    M = B * nf * nt
    dev = x.device
    c = ViTCtx()
    c.B, c.cfg = B, cfg
    U = torch.empty(M, 256, dtype=torch.bfloat16, device=dev)
    N.call("patch_unfold", x.contiguous(), U, B, F, T, fs, ts)
    X = torch.empty(M, C0, dtype=torch.float32, device=dev)
    E.gemm(BF, 0, 0, M, C0, 256, U, 256, W["v.patch_embed.proj.weight"], 256, X, C0, bias=P["v.patch_embed.proj.bias"], out_f32=1)
    c.U, c.blocks = U, []
    hw = (nf, nt)
    for i, bc in enumerate(cfg["blocks"]):
        if tuple(bc.hw) != tuple(hw):
            raise ValueError(f"block {i} was laid out for a {bc.hw} grid, the input gives {hw}")
        X, hw, k = block_forward(P, W, f"v.blocks.{i}.", X, B, bc)
        c.blocks.append(k)
    Cl = cfg["blocks"][-1].dim_out
    Ntok = hw[0] * hw[1]
    Ml = B * Ntok
    c.XL, c.Ntok, c.Cl = X, Ntok, Cl
    if cfg["final_norm"]:
        c.Yf, c.muf, c.rsf = _ln(X, P["v.norm.weight"], P["v.norm.bias"], Ml, Cl, cfg["eps"])
    else:
        c.Yf = torch.empty(Ml, Cl, dtype=torch.bfloat16, device=dev)
        N.call("cast", BF, X, c.Yf, Ml * Cl)
    c.pooled = torch.empty(B, Cl, dtype=torch.bfloat16, device=dev)
    N.call("tmean_fwd", BF, 0, c.Yf, c.pooled, B, Ntok, Cl // 64)
    out = E.linear_fwd(BF, c.pooled, W["fc.weight"], B, W["fc.weight"].shape[0], Cl, bias=P["fc.bias"], out_f32=1)
    return out, (c if need_ctx else None)


def mvit_backward(c, P, W, G, dout):
    cfg, B = c.cfg, c.B
    Ntok, Cl = c.Ntok, c.Cl
    Ml = B * Ntok
    dev = dout.device
    out_dim = W["fc.weight"].shape[0]
    Bp = (B + 7) // 8 * 8
    dob = torch.zeros(Bp, out_dim, dtype=torch.bfloat16, device=dev)
    dob[:B].copy_(dout)
    pooled = c.pooled
    if Bp != B:
        pooled = torch.zeros(Bp, Cl, dtype=torch.bfloat16, device=dev)
        pooled[:B].copy_(c.pooled)
    _wgrad(dob, pooled, G["fc.weight"], Bp, out_dim, Cl)
    E.colsum_add(N.F32, dout.float().contiguous(), B, out_dim, G["fc.bias"])
    dpool = E.linear_bwd_x(BF, dob, W["fc.weight"], Bp, out_dim, Cl, out_f32=1)
    dYf = torch.empty(Ml, Cl, dtype=torch.float32, device=dev)
    N.call("tile_rows", dpool, dYf, Ml, B, Ntok, 1.0 / Ntok, Cl)
    if cfg["final_norm"]:
        dX = torch.zeros(Ml, Cl, dtype=torch.float32, device=dev)
        N.call("layernorm_bwd", dYf, c.XL, c.muf, c.rsf, P["v.norm.weight"], dX, G["v.norm.weight"], G["v.norm.bias"], Ml, Cl)
    else:
        dX = dYf
    for i in reversed(range(len(cfg["blocks"]))):
        dX = block_backward(c.blocks[i], P, W, G, f"v.blocks.{i}.", dX, B, cfg["blocks"][i])
    C0 = cfg["embed_dim"]
    M = dX.shape[0]
    dXb = _cast(dX, M, C0)
    _wgrad(dXb, c.U, G["v.patch_embed.proj.weight"].view(C0, 256), M, C0, 256)
    E.colsum_add(N.F32, dX, M, C0, G["v.patch_embed.proj.bias"])
