"""CPU restatement of the AST-base transformer encoder (test infrastructure only - never imported by the product).

Follows `extras/mast_new/mast/models/ast_work.py:41-230` (`ASTModel`, `model_size='base224'`) and the published ViT / DeiT
block that timm's `VisionTransformer` implements (timm is not importable here: PARITY UNPINNED against timm itself; the
block maths below is the standard pre-norm formulation - LayerNorm(eps 1e-6), fused qkv Linear, softmax(q k^T / sqrt(d)) v,
proj, exact-erf GELU MLP - with timm's parameter names):

    tokens = Conv2d(1, 768, 16x16, stride (fstride, tstride))(x).flatten(2).transpose(1, 2) [+ pos_embed]
    for blk:  x = x + proj(attn(norm1(x)));  x = x + fc2(gelu(fc1(norm2(x))))
    x = norm(x) ; x = x.mean(1) ; out = fc(x)
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, eps):
        super().__init__()
        self.heads = heads
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(dim, 3 * dim)
        self.attn.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(dim, int(dim * mlp_ratio))
        self.mlp.fc2 = nn.Linear(int(dim * mlp_ratio), dim)

    def forward(self, x):
        B, S, C = x.shape
        h = self.heads
        qkv = self.attn.qkv(self.norm1(x)).reshape(B, S, 3, h, C // h).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = torch.softmax((q @ k.transpose(-2, -1)) / math.sqrt(C // h), dim=-1)
        x = x + self.attn.proj((a @ v).transpose(1, 2).reshape(B, S, C))
        return x + self.mlp.fc2(F.gelu(self.mlp.fc1(self.norm2(x))))


class ASTModel(nn.Module):
    def __init__(self, label_dim=256, fstride=10, tstride=10, input_fdim=128, input_tdim=101, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, use_pos_embed=True, final_norm=True):
        super().__init__()
        f_dim, t_dim = (input_fdim - 16) // fstride + 1, (input_tdim - 16) // tstride + 1
        self.v = nn.Module()
        self.v.patch_embed = nn.Module()
        self.v.patch_embed.proj = nn.Conv2d(1, embed_dim, kernel_size=(16, 16), stride=(fstride, tstride))
        self.v.pos_embed = nn.Parameter(torch.zeros(1, f_dim * t_dim, embed_dim))
        self.v.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, 1e-6) for _ in range(depth)])
        self.v.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.fc = nn.Linear(embed_dim, label_dim)
        self.use_pos_embed, self.final_norm = use_pos_embed, final_norm

    def forward(self, x):
        """x [B, 1, F, T] -> [B, label_dim]"""
        x = self.v.patch_embed.proj(x).flatten(2).transpose(1, 2)
        if self.use_pos_embed:
            x = x + self.v.pos_embed
        for blk in self.v.blocks:
            x = blk(x)
        if self.final_norm:
            x = self.v.norm(x)
        return self.fc(x.mean(1))


def adjust_moco_momentum(epoch, epochs=200, base=0.99):
    """`extras/mast_new/mast/utils.py:55-57`"""
    return 1. - 0.5 * (1. + math.cos(math.pi * epoch / epochs)) * (1. - base)


class SSMastExpert(nn.Module):
    """`Moco_v2` of `extras/mast_new/mast/moco_model.py:60-379`, single process: symmetric InfoNCE with a queue, the key
    encoder updated by EMA (cosine momentum schedule, evaluated at epoch + 1) inside EACH of the two forward calls."""

    def __init__(self, emb_dim=256, num_negatives=65536, softmax_temperature=0.07, **ast_kwargs):
        super().__init__()
        self.encoder_q = ASTModel(label_dim=emb_dim, **ast_kwargs)
        self.encoder_k = ASTModel(label_dim=emb_dim, **ast_kwargs)
        for pq, pk in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            pk.data.copy_(pq.data)
            pk.requires_grad = False
        self.register_buffer("queue", F.normalize(torch.randn(emb_dim, num_negatives), dim=0))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        self.T, self.K = softmax_temperature, num_negatives

    def one_direction(self, img_q, img_k, epoch):
        q = F.normalize(self.encoder_q(img_q), dim=1)
        with torch.no_grad():
            em = adjust_moco_momentum(epoch + 1)
            for pq, pk in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
                pk.data = pk.data * em + pq.data * (1. - em)
            k = F.normalize(self.encoder_k(img_k), dim=1)
        l_pos = torch.einsum('nc,nc->n', [q, k]).unsqueeze(-1)
        l_neg = torch.einsum('nc,ck->nk', [q, self.queue.clone().detach()])
        logits = torch.cat([l_pos, l_neg], dim=1) / self.T
        with torch.no_grad():
            ptr, b = int(self.queue_ptr), k.shape[0]
            assert self.K % b == 0
            self.queue[:, ptr:ptr + b] = k.T
            self.queue_ptr[0] = (ptr + b) % self.K
        return F.cross_entropy(logits.float(), torch.zeros(logits.shape[0], dtype=torch.long))

    def training_loss(self, img_1, img_2, epoch=0):
        return self.one_direction(img_1, img_2, epoch) + self.one_direction(img_2, img_1, epoch)
