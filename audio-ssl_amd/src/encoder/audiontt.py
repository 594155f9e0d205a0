"""AudioNTT2020Task6 on MI355X - same constructor, parameter names and outputs as
`src/encoder/audiontt.py:37-107` of the reference; the arithmetic is `src.engine.encoder_forward/backward`
(hand-written HIP kernels).  The torch layers below are parameter containers only (state_dict keys
`features_1.0.weight` ... `fc.3.bias` match the reference, default init included) - they are never called.
"""
import logging
import os
import re
from pathlib import Path

import torch
from torch import nn

from src import _native as N
from src import engine as E

# 1 (default): the dropout keep mask of the fc block is drawn inside the first fc GEMM's epilogue (bf16 path); 0: written out by
# `dropout_mask` and read back (12.6 MB each way per encoder pass at B = 512)
_DROPOUT_IN_GEMM = os.environ.get("AUDIOSSL_DROPOUT_IN_GEMM", "1") != "0"


def default_precision():
    return {"fp32": N.F32, "bf16": N.BF16}[os.environ.get("AUDIOSSL_PRECISION", "bf16")]


class NetworkCommonMixIn():
    """Weight-file helpers of the reference (`audiontt.py:9-34`)."""

    def load_weight(self, weight_file, device):
        state_dict = torch.load(weight_file, map_location=device, weights_only=True)
        if 'state_dict' in state_dict:
            state_dict = state_dict['state_dict']
        weights = {}
        for k in state_dict:
            m = re.search(r'(^fc\.|\.fc\.|^features\.|\.features\.)', k)
            if m is None:
                continue
            new_k = k[m.start():]
            weights[new_k[1:] if new_k[0] == '.' else new_k] = state_dict[k]
        self.load_state_dict(weights)
        self.eval()
        logging.info(f'Using audio embbeding network pretrained weight: {Path(weight_file).name}')
        return self

    def set_trainable(self, trainable=False):
        for p in self.parameters():
            p.requires_grad = trainable


class DropoutMasks:
    """Keep-masks for fc.2 (Dropout p=0.3).  Device counter-based RNG by default; tests may queue explicit masks
    (the reference draws from torch's CPU generator, which a GPU cannot reproduce - masks are inputs of parity tests)."""

    def __init__(self, p=0.3, seed=0x5EED):
        self.p, self.seed, self.queue = p, seed, []
        self.counter = None          # device int64: masks drawn so far (kept on the GPU so a hipGraph replay advances it)

    @property
    def calls(self):
        return 0 if self.counter is None else int(self.counter[0])

    def next(self, M, d, device, virtual=False):
        """The next keep mask [M, d] (uint8): a queued one (tests inject the oracle's masks), else drawn by the counter hash.
        virtual=True: no tensor - a `VirtualKeep` that tells the fc GEMM to draw exactly that mask for its own elements in the
        epilogue (`audiossl_gemm_dropout`); `.materialise()` writes it out if somebody needs the tensor after all."""
        if self.queue:
            m = self.queue.pop(0)
            return m.reshape(M, d).to(device=device, dtype=torch.uint8).contiguous()
        if self.counter is None or self.counter.device != torch.device(device):
            self.counter = torch.zeros(1, dtype=torch.int64, device=device)
        self.counter.add_(1)
        vk = VirtualKeep((self.seed * 0x9E3779B1) & 0xFFFFFFFFFFFF, self.p, self.counter, M, d)
        return vk if virtual else vk.materialise()


class VirtualKeep:
    """A dropout keep mask that exists as (seed, p, device counter) only.  Valid until the counter is advanced again ON THE DEVICE,
    i.e. it must be consumed by a launch issued before the next `DropoutMasks.next()` of the same encoder in stream order."""

    def __init__(self, seed, p, counter, M, d):
        self.seed, self.p, self.counter, self.M, self.d = seed, p, counter, M, d

    def materialise(self):
        keep = torch.empty(self.M, self.d, dtype=torch.uint8, device=self.counter.device)
        N.call("dropout_mask", keep, self.M * self.d, self.seed, self.p, self.counter)
        return keep


class _EncoderFn(torch.autograd.Function):
    """Module-level autograd bridge: forward and backward both run the HIP launch sequences."""

    @staticmethod
    def forward(ctx, mod, x, keep, *params):
        """params: the encoder's parameters in the order of `mod.engine_param_names()` (autograd hands their gradients
        back in that order).  `mod.param_dict()` is keyed the way the launch sequences expect (`features_1.0.weight` ...)."""
        P = mod.param_dict()
        x1, x2, x3, h, c = E.encoder_forward(P, x, mod.precision, keep=keep, p_drop=mod.fc[2].p, train=mod.training)
        ctx.mod, ctx.c = mod, c
        return x1, x2, x3, h

    @staticmethod
    def backward(ctx, g1, g2, g3, gh):
        mod, c = ctx.mod, ctx.c
        td = N.torch_dtype(c.dtype)
        P = mod.param_dict()
        names = mod.engine_param_names()
        G = {n: torch.zeros_like(P[n], dtype=torch.float32) for n in names}

        def prep(g, like):
            return None if g is None else g.float().contiguous()         # layer-mean gradients enter pool backwards: fp32
        gh = torch.zeros_like(c.H2.view(c.N, -1, c.d)) if gh is None else gh.to(td).contiguous()
        E.encoder_backward(c, G, dH2=gh.view(c.M, c.d), dx1=prep(g1, None), dx2=prep(g2, None), dx3=prep(g3, None))
        return (None, None, None) + tuple(G[n] for n in names)


class AudioNTT2020Task6(nn.Module, NetworkCommonMixIn):
    """DCASE2020 Task6 NTT audio embedding network (BYOL-A encoder)."""

    def __init__(self, n_mels, d, return_all_layers):
        super().__init__()
        self.return_all_layers = return_all_layers

        def block(cin):
            return nn.Sequential(nn.Conv2d(cin, 64, 3, stride=1, padding=1), nn.BatchNorm2d(64), nn.ReLU(),
                                 nn.MaxPool2d(2, stride=2))
        self.features_1 = block(1)
        self.features_2 = block(64)
        self.features_3 = block(64)
        self.fc = nn.Sequential(nn.Linear(64 * (n_mels // (2 ** 3)), d), nn.ReLU(), nn.Dropout(p=0.3), nn.Linear(d, d),
                                nn.ReLU())
        self.d = d
        self.n_mels = n_mels
        self.precision = default_precision()
        self.dropout_masks = DropoutMasks(0.3)

    def param_dict(self):
        """Reference-keyed fp32 tensors (parameters and BatchNorm buffers)."""
        from src.flat import cached_param_dict
        return cached_param_dict(self)

    def engine_param_names(self):
        return [n for n, _ in self.named_parameters()]

    def next_keep_mask(self, n_img, T):
        if not self.training:
            return None
        return self.dropout_masks.next(n_img * (T // 8), self.d, self.fc[0].weight.device, virtual=_DROPOUT_IN_GEMM)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("AudioNTT2020Task6 (HIP) needs a GPU tensor - there is no CPU fallback")
        x = x.float().contiguous()
        keep = self.next_keep_mask(x.shape[0], x.shape[-1])
        params = tuple(p for _, p in self.named_parameters())
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            x1, x2, x3, h = _EncoderFn.apply(self, x, keep, *params)
        else:
            x1, x2, x3, h, _ = E.encoder_forward(self.param_dict(), x, self.precision, keep=keep, p_drop=self.fc[2].p,
                                                 train=self.training)
        if self.return_all_layers:
            return x1, x2, x3, h
        return h

    def __repr__(self):
        return "AudioNTT2020Task6"
