"""CPU restatement of the MViTv2 pooling-attention block (test infrastructure only - never imported by the product).

What it restates: `MultiScaleBlock` / `MultiScaleAttention` of `extras/mast_new/mast/mvit/models/attention.py:93-393`
(the block `models_msn.py:147` -> `ASTModel(model_size='mvit')` stacks), in the `mode="conv"`, `pool_first=False` form the
shipped configs use (`configs/MVITv2_*.yaml`).  Written functionally over a flat name -> tensor dict with the reference's
parameter names, so a state_dict of the reference block can be fed in directly.

PINNED by `tests/golden/mvit_block.npz` (outputs and gradient digests produced by running the reference class itself,
`tests/golden/make_goldens.py: g15_mvit_block`), checked in `tests/test_oracle_golden.py`.

    x_n   = LN1(x)
    q,k,v = split(qkv(x_n)) per head                                   [B, heads, L, d]
    q,k,v = LN_d(depthwise_conv3x3(. as [B*heads, d, H, W], stride))    (pooling; per-head channels share ONE filter bank)
    a     = softmax(scale q k^T + rel_h[q, kh] + rel_w[q, kw])          (decomposed relative position terms, :44-90)
    y     = a v (+ q: "residual pooling")  -> proj
    skip  = maxpool(x or proj(x_n))  when the query stride > 1 / the width changes inside the attention
    x     = skip + y ;  x = x (or proj(LN2 x) when the width changes after the attention) + mlp(LN2 x)
"""
import math

import torch
import torch.nn.functional as F


def _pool_tokens(t, hw, weight, stride, ln_w, ln_b, eps):
    """t [B, heads, H*W, d] -> depthwise 3x3 conv (pad k//2, given stride) over the H x W grid, LayerNorm over d."""
    if weight is None:
        return t, hw
    B, nh, L, d = t.shape
    H, W = hw
    kh, kw = weight.shape[-2:]
    img = t.reshape(B * nh, H, W, d).permute(0, 3, 1, 2)
    img = F.conv2d(img, weight, None, stride=stride, padding=(kh // 2, kw // 2), groups=d)
    Ho, Wo = img.shape[-2:]
    out = img.reshape(B, nh, d, Ho * Wo).transpose(2, 3)
    return F.layer_norm(out, (d,), ln_w, ln_b, eps), (Ho, Wo)


def rel_pos_index(nq, nk):
    """Row table of the decomposed relative-position lookup: index[i, j] into the (2 max(nq, nk) - 1)-row embedding for query
    coordinate i and key coordinate j when the two grids differ by an integer factor (`attention.py:59-71`)."""
    rq, rk = max(nk / nq, 1.0), max(nq / nk, 1.0)
    i = torch.arange(nq, dtype=torch.float32)[:, None] * rq
    j = torch.arange(nk, dtype=torch.float32)[None, :] * rk
    return (i - j + (nk - 1) * rk).long()


def multiscale_block(P, x, hw, *, dim, dim_out, heads, stride_q=(), stride_kv=(), kernel=(3, 3), rel_pos=False,
                     residual_pooling=True, dim_mul_in_att=False, eps=1e-6, prefix=""):
    """P: name -> tensor with the reference block's parameter names (optionally under `prefix`).  x [B, L, dim], hw = (H, W)
    with L = H*W (no class token).  stride_* = () means "no pooling on that path".  -> (y [B, L', dim_out], (H', W'))."""
    g = lambda n: P.get(prefix + n)
    att_dim = dim_out if dim_mul_in_att else dim
    d = att_dim // heads
    B, L, _ = x.shape
    xn = F.layer_norm(x, (dim,), g("norm1.weight"), g("norm1.bias"), eps)
    qkv = F.linear(xn, g("attn.qkv.weight"), g("attn.qkv.bias")).reshape(B, L, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, q_hw = _pool_tokens(qkv[0], hw, g("attn.pool_q.weight") if len(stride_q) else None, tuple(stride_q),
                           g("attn.norm_q.weight"), g("attn.norm_q.bias"), eps)
    k, k_hw = _pool_tokens(qkv[1], hw, g("attn.pool_k.weight") if len(stride_kv) else None, tuple(stride_kv),
                           g("attn.norm_k.weight"), g("attn.norm_k.bias"), eps)
    v, _ = _pool_tokens(qkv[2], hw, g("attn.pool_v.weight") if len(stride_kv) else None, tuple(stride_kv),
                        g("attn.norm_v.weight"), g("attn.norm_v.bias"), eps)
    s = (q * d ** -0.5) @ k.transpose(-1, -2)                                          # [B, heads, Lq, Lk]
    if rel_pos:
        (qh, qw), (kh_, kw_) = q_hw, k_hw
        Rh = g("attn.rel_pos_h")[rel_pos_index(qh, kh_)]                               # [qh, kh, d]
        Rw = g("attn.rel_pos_w")[rel_pos_index(qw, kw_)]                               # [qw, kw, d]
        qg = q.reshape(B, heads, qh, qw, d)                                            # the UNSCALED pooled query
        bh = torch.einsum("bnhwc,hkc->bnhwk", qg, Rh)                                  # depends on (query, key row)
        bw = torch.einsum("bnhwc,wkc->bnhwk", qg, Rw)                                  # depends on (query, key column)
        s = (s.reshape(B, heads, qh, qw, kh_, kw_) + bh[..., :, None] + bw[..., None, :]).reshape(B, heads, qh * qw, kh_ * kw_)
    y = torch.softmax(s, dim=-1) @ v
    if residual_pooling:
        y = y + q
    y = F.linear(y.transpose(1, 2).reshape(B, -1, att_dim), g("attn.proj.weight"), g("attn.proj.bias"))
    skip = xn if (dim_mul_in_att and dim != dim_out) else x
    if dim_mul_in_att and dim != dim_out:
        skip = F.linear(skip, g("proj.weight"), g("proj.bias"))
    if len(stride_q) and math.prod(stride_q) > 1:
        ks = [s_ + 1 if s_ > 1 else s_ for s_ in stride_q]
        H, W = hw
        C = skip.shape[-1]
        img = skip.reshape(B, H, W, C).permute(0, 3, 1, 2)
        img = F.max_pool2d(img, ks, tuple(stride_q), [k_ // 2 for k_ in ks])
        skip = img.reshape(B, C, -1).transpose(1, 2)
    x = skip + y
    xn2 = F.layer_norm(x, (att_dim,), g("norm2.weight"), g("norm2.bias"), eps)
    m = F.linear(F.gelu(F.linear(xn2, g("mlp.fc1.weight"), g("mlp.fc1.bias"))), g("mlp.fc2.weight"), g("mlp.fc2.bias"))
    if not dim_mul_in_att and dim != dim_out:
        x = F.linear(xn2, g("proj.weight"), g("proj.bias"))
    return x + m, q_hw


# the block configurations the golden fixture covers (name -> keyword arguments of `multiscale_block` + the input grid)
GOLDEN_CONFIGS = {
    # no pooling, no relative positions, no residual pooling: the plain pre-norm ViT block (what the AST-base encoder stacks)
    "plain": dict(dim=128, dim_out=128, heads=2, hw=(12, 9), rel_pos=False, residual_pooling=False),
    # inside a stage: queries pooled at stride 1, keys / values at stride 2, relative positions, residual pooling
    "pooled": dict(dim=128, dim_out=128, heads=2, hw=(8, 8), stride_q=(1, 1), stride_kv=(2, 2), rel_pos=True,
                   residual_pooling=True, dim_mul_in_att=True),
    # stage transition: width x2 inside the attention, heads x2, queries pooled at stride 2 (skip path max-pooled)
    "transition": dict(dim=128, dim_out=256, heads=4, hw=(8, 8), stride_q=(2, 2), stride_kv=(1, 1), rel_pos=True,
                       residual_pooling=True, dim_mul_in_att=True),
}


def block_shapes(dim, dim_out, heads, hw, stride_q=(), stride_kv=(), rel_pos=False, dim_mul_in_att=False, mlp_ratio=4.0, **_):
    """Parameter names -> shapes of one block, in the reference module's state_dict order."""
    att = dim_out if dim_mul_in_att else dim
    d = att // heads
    sh = {"norm1.weight": (dim,), "norm1.bias": (dim,), "attn.qkv.weight": (3 * att, dim), "attn.qkv.bias": (3 * att,),
          "attn.proj.weight": (att, att), "attn.proj.bias": (att,)}
    if rel_pos:
        size = hw[0]
        qs = size // stride_q[1] if len(stride_q) else size
        ks = size // stride_kv[1] if len(stride_kv) else size
        sh["attn.rel_pos_h"] = sh["attn.rel_pos_w"] = (2 * max(qs, ks) - 1, d)
    for nm, on in (("q", len(stride_q)), ("k", len(stride_kv)), ("v", len(stride_kv))):
        if on:
            sh[f"attn.pool_{nm}.weight"] = (d, 1, 3, 3)
            sh[f"attn.norm_{nm}.weight"] = sh[f"attn.norm_{nm}.bias"] = (d,)
    hid = int(att * mlp_ratio)
    sh.update({"norm2.weight": (att,), "norm2.bias": (att,), "mlp.fc1.weight": (hid, att), "mlp.fc1.bias": (hid,),
               "mlp.fc2.weight": (dim_out, hid), "mlp.fc2.bias": (dim_out,)})
    if dim != dim_out:
        sh["proj.weight"], sh["proj.bias"] = (dim_out, dim), (dim_out,)
    return sh


def golden_case(cname):
    """(parameters, input x, output gradient seed) of a fixture configuration, exactly as the generator filled them."""
    import numpy as np
    from oracle import fill
    cfg = GOLDEN_CONFIGS[cname]
    P = fill.fill_shapes(block_shapes(**cfg), seed=150 + len(cname))
    for n in list(P):
        if "rel_pos" in n:
            P[n] = torch.from_numpy(fill.uniform(tuple(P[n].shape), fill.salt_of(cname + n), -0.2, 0.2))
    L = cfg["hw"][0] * cfg["hw"][1]
    x = torch.from_numpy(fill.normalish((2, L, cfg["dim"]), 1500 + len(cname)))
    kw = {k: v for k, v in cfg.items() if k != "hw"}
    return P, x, kw, 1501 + len(cname)


def mvit_encoder(P, x, blocks, fstride=10, tstride=10, final_norm=False, eps=1e-6):
    """The encoder `ASTModel(model_size='mvit')` runs (`models/ast_work.py:101, 183-230`: 16 x 16 patch embedding with strides
    (fstride, tstride), the MultiScaleBlocks, mean over the tokens; no position embedding, no final norm in the shipped forward)
    plus the Linear the MoCo wrapper adds.  P: name -> tensor with the product's parameter names (`v.patch_embed.proj.*`,
    `v.blocks.<i>.*`, `fc.*`); blocks: list of dicts with the keyword arguments of `multiscale_block` + `hw`.
    x [B, 1, F, T] -> [B, out_dim].  UNPINNED as a whole (the reference builds this shell with timm, absent here): the blocks are
    the pinned `multiscale_block`, the shell follows the call sites in ast_work.py / moco_model.py."""
    t = F.conv2d(x, P["v.patch_embed.proj.weight"], P["v.patch_embed.proj.bias"], stride=(fstride, tstride))
    hw = tuple(t.shape[-2:])
    t = t.flatten(2).transpose(1, 2)
    for i, b in enumerate(blocks):
        kw = {k: v for k, v in b.items() if k != "hw"}
        assert tuple(b["hw"]) == tuple(hw), (i, b["hw"], hw)
        t, hw = multiscale_block(P, t, hw, prefix=f"v.blocks.{i}.", eps=eps, **kw)
    if final_norm:
        t = F.layer_norm(t, (t.shape[-1],), P["v.norm.weight"], P["v.norm.bias"], eps)
    return F.linear(t.mean(1), P["fc.weight"], P["fc.bias"])
