"""AST / MAST transformer encoder on MI355X (BASELINE config 4: "AST-base 12x768").

`ASTModel` follows `extras/mast_new/mast/models/ast_work.py:41-230` of the reference (the `model_size='base224'` branch:
timm's DeiT-base, 12 pre-norm blocks x 768 x 12 heads): 16x16 patches of the [B, 1, F, T] log-mel with strides
(fstride, tstride), a learned position embedding per patch (`:126-130`), the blocks, mean over the patch tokens (`:229`),
plus the Linear(768, out_dim) the MoCo wrapper puts on top (`moco_model.py:148-149`: 256-d embedding).  Parameter names
are timm's (`v.patch_embed.proj`, `v.pos_embed`, `v.blocks.<i>.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}`, `v.norm`).
The shipped forward comments out the position-embedding add and the final norm (it was written around the MViT variant,
which carries its own relative positions); both are on by default here and can be switched off.
timm is not importable in this image, so parity is pinned against the CPU restatement `oracle/vit.py` only
("parity unpinned" against timm itself).
"""
import torch
from torch import nn

from src import _native as N
from src import vit_engine as VE


class _Block(nn.Module):
    def __init__(self, dim, mlp_ratio, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(dim, 3 * dim)
        self.attn.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(dim, int(dim * mlp_ratio))
        self.mlp.fc2 = nn.Linear(int(dim * mlp_ratio), dim)


class _MViTBlock(nn.Module):
    """Parameter container of one `MultiScaleBlock` (`mvit/models/attention.py:304-393`), names and shapes as the reference's."""

    def __init__(self, c, rel_rows, mlp_ratio):
        super().__init__()
        dim, dout, att, d = c.dim, c.dim_out, c.att, c.d
        self.norm1 = nn.LayerNorm(dim, eps=c.eps)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(dim, 3 * att)
        self.attn.proj = nn.Linear(att, att)
        for nm, stride in (("q", c.stride_q), ("k", c.stride_kv), ("v", c.stride_kv)):
            if len(stride):
                setattr(self.attn, f"pool_{nm}", nn.Conv2d(d, d, 3, stride=stride, padding=1, groups=d, bias=False))
                setattr(self.attn, f"norm_{nm}", nn.LayerNorm(d, eps=c.eps))
        if c.rel_pos:
            self.attn.rel_pos_h = nn.Parameter(torch.zeros(rel_rows, d))
            self.attn.rel_pos_w = nn.Parameter(torch.zeros(rel_rows, d))
            nn.init.trunc_normal_(self.attn.rel_pos_h, std=0.02)
            nn.init.trunc_normal_(self.attn.rel_pos_w, std=0.02)
        self.norm2 = nn.LayerNorm(att, eps=c.eps)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(att, int(att * mlp_ratio))
        self.mlp.fc2 = nn.Linear(int(att * mlp_ratio), dout)
        if dim != dout:
            self.proj = nn.Linear(dim, dout)


class _ViTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, *params):
        P, W = mod.param_dict(), mod.weight_shadow()
        out, c = mod.engine_forward(P, W, x, mod.cfg, need_ctx=True)
        ctx.mod, ctx.c = mod, c
        return out

    @staticmethod
    def backward(ctx, dout):
        mod, c = ctx.mod, ctx.c
        P, W = mod.param_dict(), mod.weight_shadow(refresh=False)
        G = {n: torch.zeros_like(p, dtype=torch.float32) for n, p in mod.named_parameters()}
        mod.engine_backward(c, P, W, G, dout.float().contiguous())
        return (None, None) + tuple(G[n] for n, _ in mod.named_parameters())


class ASTModel(nn.Module):
    def __init__(self, label_dim=256, fstride=10, tstride=10, input_fdim=128, input_tdim=101, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, use_pos_embed=True, final_norm=True, model_size="base224", mvit=None):
        super().__init__()
        self.engine_forward, self.engine_backward = VE.vit_forward, VE.vit_backward
        if model_size == "mvit":
            self._init_mvit(label_dim, fstride, tstride, input_fdim, input_tdim, mlp_ratio, mvit or {})
            return
        if model_size != "base224":
            raise NotImplementedError("the HIP path implements the AST-base (DeiT-base 12 x 768) and the MViTv2 ('mvit') encoders")
        if embed_dim != num_heads * 64:
            raise NotImplementedError("the attention kernel is built for 64-wide heads")
        self.f_dim, self.t_dim = self.get_shape(fstride, tstride, input_fdim, input_tdim)
        num_patches = self.f_dim * self.t_dim
        eps = 1e-6
        self.v = nn.Module()
        self.v.patch_embed = nn.Module()
        self.v.patch_embed.proj = nn.Conv2d(1, embed_dim, kernel_size=(16, 16), stride=(fstride, tstride))
        self.v.patch_embed.num_patches = num_patches
        self.v.pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim))
        nn.init.trunc_normal_(self.v.pos_embed, std=.02)
        self.v.blocks = nn.ModuleList([_Block(embed_dim, mlp_ratio, eps) for _ in range(depth)])
        self.v.norm = nn.LayerNorm(embed_dim, eps=eps)
        self.fc = nn.Linear(embed_dim, label_dim)
        self.cfg = dict(embed_dim=embed_dim, num_heads=num_heads, depth=depth, eps=eps, fstride=fstride, tstride=tstride,
                        use_pos_embed=use_pos_embed, final_norm=final_norm)
        self._shadow = None

    def _init_mvit(self, label_dim, fstride, tstride, input_fdim, input_tdim, mlp_ratio, mv):
        """`ASTModel(model_size='mvit')` - what `models_msn.py:147` instantiates for SS-MAST: the 16 x 16 / stride-10 patch
        embedding of AST (`ast_work.py:101`) feeding the MViTv2 blocks; the shipped forward (`ast_work.py:183-230`) applies no
        position embedding and no final norm, takes the mean over the tokens, and the MoCo wrapper adds Linear(768, 256).
        `mv`: overrides of `mvit_engine.stage_layout`'s defaults (= configs/MVITv2_B.yaml: embed 96, depth 24, x2 at 2 / 5 / 21).
        Relative-position tables have 2 max(q, k) - 1 rows for the LARGER of the grid's two sides (the vendored constructor sizes
        them from the first side after floor divisions, `attention.py:153-160`, which is too short for grids like 12 x 9 -> 2 x 2
        whose strided convolutions round up)."""
        from src import mvit_engine as ME
        self.engine_forward, self.engine_backward = ME.mvit_forward, ME.mvit_backward
        self.f_dim, self.t_dim = self.get_shape(fstride, tstride, input_fdim, input_tdim)
        blocks = ME.stage_layout((self.f_dim, self.t_dim), **{k: v for k, v in mv.items() if k != "final_norm"})
        eps = blocks[0].eps
        self.v = nn.Module()
        self.v.patch_embed = nn.Module()
        self.v.patch_embed.proj = nn.Conv2d(1, blocks[0].dim, kernel_size=(16, 16), stride=(fstride, tstride))
        self.v.patch_embed.num_patches = self.f_dim * self.t_dim
        mods = []
        for c in blocks:
            rows = 2 * max(max(c.q_hw), max(c.k_hw)) - 1
            mods.append(_MViTBlock(c, rows, mlp_ratio))
        self.v.blocks = nn.ModuleList(mods)
        width = blocks[-1].dim_out
        final_norm = bool(mv.get("final_norm", False))
        if final_norm:
            self.v.norm = nn.LayerNorm(width, eps=eps)
        self.fc = nn.Linear(width, label_dim)
        self.cfg = dict(blocks=blocks, embed_dim=blocks[0].dim, fstride=fstride, tstride=tstride, final_norm=final_norm, eps=eps)
        self._shadow = None
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)            # `MViT._init_weights`, mvit_model.py:233-242
                nn.init.zeros_(m.bias)

    @staticmethod
    def get_shape(fstride, tstride, input_fdim=128, input_tdim=1024):
        return (input_fdim - 16) // fstride + 1, (input_tdim - 16) // tstride + 1

    def param_dict(self):
        from src.flat import cached_param_dict
        return cached_param_dict(self)

    def weight_shadow(self, refresh=True):
        """bf16 copies of every GEMM weight (the conv kernel flattened to [768, 256]); rebuilt once per forward."""
        if refresh or self._shadow is None:
            W = {}
            for n, p in self.named_parameters():
                if p.dim() >= 2 and n != "v.pos_embed" and "rel_pos" not in n and ".pool_" not in n:
                    w = torch.empty(p.shape[0], p.numel() // p.shape[0], dtype=torch.bfloat16, device=p.device)
                    N.call("cast", N.BF16, p.data, w, p.numel())
                    W[n] = w
            self._shadow = W
        return self._shadow

    def forward(self, x, patch_drop=0.0):
        """x [B, 1, F, T] log-mel (the layout the front end produces; the reference transposes its [B, 1, T, F] input
        to this before the patch embedding, `ast_work.py:190`) -> [B, label_dim] fp32."""
        if not x.is_cuda:
            raise RuntimeError("ASTModel (HIP) needs a GPU tensor - there is no CPU fallback")
        x = x.float().contiguous()
        params = tuple(p for _, p in self.named_parameters())
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _ViTFn.apply(self, x, *params)
        return self.engine_forward(self.param_dict(), self.weight_shadow(), x, self.cfg, need_ctx=False)[0]

    def __repr__(self):
        return "ASTModel"


MAST = ASTModel        # the name `src/encoder/__init__.py:20-24` of the reference registers for this backbone
