"""Launch sequences of the AST / MAST transformer encoder on one GPU (host orchestration of libaudiossl_hip.so).

Forward and backward of the ViT that `ASTModel` wraps (`extras/mast_new/mast/models/ast_work.py:70-81, 101, 183-230`:
timm DeiT-base = 12 pre-norm blocks x 768 x 12 heads) as sequences of C-ABI calls:

  patch_unfold -> GEMM(+bias, + position embedding as the residual operand)              tokens [B*N, 768] fp32
  per block:  layernorm -> GEMM qkv -> attention (one workgroup per clip, head and 128-token block) -> GEMM proj (+bias +residual)
              layernorm -> GEMM fc1 (+bias) -> GELU -> GEMM fc2 (+bias +residual)
  layernorm -> mean over tokens -> GEMM head

The residual stream is fp32 and every block writes a NEW tensor for it (the GEMM epilogue adds the residual), so the
LayerNorm inputs the backward needs are simply kept - no snapshot copies.  MFMA operands are bf16; gradients entering a
LayerNorm backward and the residual-stream gradient are fp32.  Parameter gradients are ACCUMULATED into the tensors
handed in (flat-buffer views in training).
"""
import math

import torch

from src import _native as N
from src import engine as E

BF = N.BF16


class ViTCtx:
    """Everything the backward needs from one forward call."""
    pass


def _ln(x, g, b, M, C, eps, y32=None):
    y = torch.empty(M, C, dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    N.call("layernorm_fwd", x, g, b, y, y32, mean, rstd, M, C, eps)
    return y, mean, rstd


def block_forward(P, W, p, X, B, Ntok, C, H, eps):
    """One pre-norm transformer block (timm `Block`; equally the reference's `MultiScaleBlock` without pooling,
    `mvit/models/attention.py:304-393`): X fp32 [B*Ntok, C] -> (X' fp32, ctx).  Parameters under prefix `p`."""
    M, dev = B * Ntok, X.device
    scale = 1.0 / math.sqrt(C // H)
    k = ViTCtx()
    k.X1 = X
    k.Y1, k.mu1, k.rs1 = _ln(X, P[p + "norm1.weight"], P[p + "norm1.bias"], M, C, eps)
    k.QKV = torch.empty(M, 3 * C, dtype=torch.bfloat16, device=dev)
    E.gemm(BF, 0, 0, M, 3 * C, C, k.Y1, C, W[p + "attn.qkv.weight"], C, k.QKV, 3 * C, bias=P[p + "attn.qkv.bias"])
    k.A = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    k.lse = torch.empty(B * H, Ntok, dtype=torch.float32, device=dev)
    N.call("attn_fwd", k.QKV, k.A, k.lse, B, Ntok, H, scale)
    X2 = torch.empty(M, C, dtype=torch.float32, device=dev)
    E.gemm(BF, 0, 0, M, C, C, k.A, C, W[p + "attn.proj.weight"], C, X2, C, bias=P[p + "attn.proj.bias"], out_f32=1,
           resid=X, ldr=C)
    k.X2 = X2
    k.Y2, k.mu2, k.rs2 = _ln(X2, P[p + "norm2.weight"], P[p + "norm2.bias"], M, C, eps)
    Hd = W[p + "mlp.fc1.weight"].shape[0]
    k.A1 = torch.empty(M, Hd, dtype=torch.bfloat16, device=dev)
    E.gemm(BF, 0, 0, M, Hd, C, k.Y2, C, W[p + "mlp.fc1.weight"], C, k.A1, Hd, bias=P[p + "mlp.fc1.bias"])
    k.H1 = torch.empty(M, Hd, dtype=torch.bfloat16, device=dev)
    N.call("gelu_fwd", k.A1, k.H1, M * Hd)
    X3 = torch.empty(M, C, dtype=torch.float32, device=dev)
    E.gemm(BF, 0, 0, M, C, Hd, k.H1, Hd, W[p + "mlp.fc2.weight"], Hd, X3, C, bias=P[p + "mlp.fc2.bias"], out_f32=1,
           resid=X2, ldr=C)
    return X3, k


def vit_forward(P, W, x, cfg, need_ctx=True):
    """P: name -> fp32 parameter (keys of `src.encoder.mast.ASTModel`, prefix stripped); W: name -> bf16 copy of every
    2-D weight.  x [B, 1, F, T] fp32 log-mel.  Returns (embedding [B, out_dim] fp32, ctx)."""
    B, _, F, T = x.shape
    C, H, depth, eps = cfg["embed_dim"], cfg["num_heads"], cfg["depth"], cfg["eps"]
    fs, ts = cfg["fstride"], cfg["tstride"]
    nf, nt = (F - 16) // fs + 1, (T - 16) // ts + 1
    Ntok = nf * nt
    M = B * Ntok
    dev = x.device
    if B * H > 65535:
        raise ValueError("batch * heads must stay within 65,535 (attention grid)")
    if M % 8:
        raise ValueError("batch * patches must be a multiple of 8 (weight-gradient GEMMs contract over the token rows)")
    c = ViTCtx()
    c.B, c.Ntok, c.M, c.cfg = B, Ntok, M, cfg
    U = torch.empty(M, 256, dtype=torch.bfloat16, device=dev)
    N.call("patch_unfold", x.contiguous(), U, B, F, T, fs, ts)
    X = torch.empty(M, C, dtype=torch.float32, device=dev)
    if cfg["use_pos_embed"]:
        pos = torch.empty(M, C, dtype=torch.float32, device=dev)
        N.call("tile_rows", P["v.pos_embed"], pos, M, Ntok, 1, 1.0, C)
        E.gemm(BF, 0, 0, M, C, 256, U, 256, W["v.patch_embed.proj.weight"], 256, X, C, bias=P["v.patch_embed.proj.bias"],
               out_f32=1, resid=pos, ldr=C)
    else:
        E.gemm(BF, 0, 0, M, C, 256, U, 256, W["v.patch_embed.proj.weight"], 256, X, C, bias=P["v.patch_embed.proj.bias"], out_f32=1)
    c.U = U
    c.blocks = []
    scale = 1.0 / math.sqrt(C // H)
    for i in range(depth):
        X, k = block_forward(P, W, f"v.blocks.{i}.", X, B, Ntok, C, H, eps)
        c.blocks.append(k)
    c.XL = X
    if cfg["final_norm"]:
        c.Yf, c.muf, c.rsf = _ln(X, P["v.norm.weight"], P["v.norm.bias"], M, C, eps)
    else:
        c.Yf = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        N.call("cast", BF, X, c.Yf, M * C)
    c.pooled = torch.empty(B, C, dtype=torch.bfloat16, device=dev)
    N.call("tmean_fwd", BF, 0, c.Yf, c.pooled, B, Ntok, C // 64)
    out = E.linear_fwd(BF, c.pooled, W["fc.weight"], B, W["fc.weight"].shape[0], C, bias=P["fc.bias"], out_f32=1)
    if not need_ctx:
        return out, None
    return out, c


def _cast(x32, M, C):
    y = torch.empty(M, C, dtype=torch.bfloat16, device=x32.device)
    N.call("cast", BF, x32, y, M * C)
    return y


def _wgrad(dYb, Xb, dW, M, Nout, K):
    """dW[Nout, K] += dY[M, Nout]^T X[M, K]"""
    E.linear_bwd_w(BF, dYb, Xb, dW, M, Nout, K)


def block_backward(k, P, W, G, p, dX, B, Ntok, C, H):
    """Backward of `block_forward`: dX fp32 [B*Ntok, C] is the gradient of the block's output on entry and of its input on
    return (updated in place); parameter gradients are accumulated into G."""
    M, dev = B * Ntok, dX.device
    scale = 1.0 / math.sqrt(C // H)
    Hd = W[p + "mlp.fc1.weight"].shape[0]
    # ---- MLP branch: X3 = X2 + fc2(gelu(fc1(LN2(X2))))
    dXb = _cast(dX, M, C)
    _wgrad(dXb, k.H1, G[p + "mlp.fc2.weight"], M, C, Hd)
    E.colsum_add(N.F32, dX, M, C, G[p + "mlp.fc2.bias"])
    dH1 = E.linear_bwd_x(BF, dXb, W[p + "mlp.fc2.weight"], M, C, Hd)                   # bf16 [M, Hd]
    dA1 = torch.empty(M, Hd, dtype=torch.bfloat16, device=dev)
    N.call("gelu_bwd", k.A1, dH1, dA1, M * Hd)
    _wgrad(dA1, k.Y2, G[p + "mlp.fc1.weight"], M, Hd, C)
    E.colsum_add(BF, dA1, M, Hd, G[p + "mlp.fc1.bias"])
    dY2 = E.linear_bwd_x(BF, dA1, W[p + "mlp.fc1.weight"], M, Hd, C, out_f32=1)
    N.call("layernorm_bwd", dY2, k.X2, k.mu2, k.rs2, P[p + "norm2.weight"], dX, G[p + "norm2.weight"], G[p + "norm2.bias"], M, C)
    # ---- attention branch: X2 = X1 + proj(attn(qkv(LN1(X1))))
    dXb = _cast(dX, M, C)
    _wgrad(dXb, k.A, G[p + "attn.proj.weight"], M, C, C)
    E.colsum_add(N.F32, dX, M, C, G[p + "attn.proj.bias"])
    dA = E.linear_bwd_x(BF, dXb, W[p + "attn.proj.weight"], M, C, C)                   # bf16 [M, C]
    dQKV = torch.empty(M, 3 * C, dtype=torch.bfloat16, device=dev)
    N.call("attn_bwd", k.QKV, k.A, dA, k.lse, dQKV, B, Ntok, H, scale)
    _wgrad(dQKV, k.Y1, G[p + "attn.qkv.weight"], M, 3 * C, C)
    E.colsum_add(BF, dQKV, M, 3 * C, G[p + "attn.qkv.bias"])
    dY1 = E.linear_bwd_x(BF, dQKV, W[p + "attn.qkv.weight"], M, 3 * C, C, out_f32=1)
    N.call("layernorm_bwd", dY1, k.X1, k.mu1, k.rs1, P[p + "norm1.weight"], dX, G[p + "norm1.weight"], G[p + "norm1.bias"], M, C)


def vit_backward(c, P, W, G, dout):
    """dout [B, out_dim] fp32 -> accumulates every parameter gradient into G (fp32, keyed like P)."""
    cfg = c.cfg
    B, Ntok, M = c.B, c.Ntok, c.M
    C, H, depth = cfg["embed_dim"], cfg["num_heads"], cfg["depth"]
    dev = dout.device
    scale = 1.0 / math.sqrt(C // H)
    out_dim = W["fc.weight"].shape[0]
    Bp = (B + 7) // 8 * 8                                    # the weight-gradient GEMM contracts over rows: multiple of 8
    dob = torch.zeros(Bp, out_dim, dtype=torch.bfloat16, device=dev)
    dob[:B].copy_(dout)
    pooled = c.pooled
    if Bp != B:
        pooled = torch.zeros(Bp, C, dtype=torch.bfloat16, device=dev)
        pooled[:B].copy_(c.pooled)
    _wgrad(dob, pooled, G["fc.weight"], Bp, out_dim, C)
    E.colsum_add(N.F32, dout.float().contiguous(), B, out_dim, G["fc.bias"])
    dpool = E.linear_bwd_x(BF, dob, W["fc.weight"], Bp, out_dim, C, out_f32=1)             # [Bp, C] fp32
    dYf = torch.empty(M, C, dtype=torch.float32, device=dev)         # mean over tokens backward: each token row gets dpool / Ntok
    N.call("tile_rows", dpool, dYf, M, B, Ntok, 1.0 / Ntok, C)
    dX = torch.zeros(M, C, dtype=torch.float32, device=dev)
    if cfg["final_norm"]:
        N.call("layernorm_bwd", dYf, c.XL, c.muf, c.rsf, P["v.norm.weight"], dX, G["v.norm.weight"], G["v.norm.bias"], M, C)
    else:
        dX = dYf
    for i in reversed(range(depth)):
        block_backward(c.blocks[i], P, W, G, f"v.blocks.{i}.", dX, B, Ntok, C, H)
    # ---- patch embedding (the input is data: no gradient beyond the projection) and the position embedding
    dXb = _cast(dX, M, C)
    _wgrad(dXb, c.U, G["v.patch_embed.proj.weight"].view(C, 256), M, C, 256)
    E.colsum_add(N.F32, dX, M, C, G["v.patch_embed.proj.bias"])
    if cfg["use_pos_embed"]:
        E.colsum_add(N.F32, dX.view(B, Ntok * C), B, Ntok * C, G["v.pos_embed"].view(-1))
