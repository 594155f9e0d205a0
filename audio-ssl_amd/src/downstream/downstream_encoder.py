"""Downstream (linear-probe / fine-tune) encoder on MI355X - `src/downstream/downstream_encoder.py:8-45` of the
reference: base encoder -> mean over time -> Linear(output_dim, classes).  The frozen-encoder probe
(BASELINE config 5) is the inference path of the same HIP encoder kernels; `load_pretrained_encoder` copies an
upstream checkpoint's `encoder_q` weights into it (state_dict keys are the reference's)."""
import torch
from torch import nn

from src import _native as N
from src.functional import LinearFn


class MeanTFn(torch.autograd.Function):
    """[N, T, D] -> mean over T (tmean kernel); backward broadcasts g / T."""

    @staticmethod
    def forward(ctx, h):
        h = h.contiguous()
        dt = N.F32 if h.dtype == torch.float32 else N.BF16
        n, T, D = h.shape
        y = torch.empty(n, D, dtype=h.dtype, device=h.device)
        N.call("tmean_fwd", dt, 0, h, y, n, T, D // 64)
        ctx.shape = (n, T, D, dt)
        return y

    @staticmethod
    def backward(ctx, gy):
        n, T, D, dt = ctx.shape
        td = N.torch_dtype(dt)
        g = torch.empty(n * T, D, dtype=td, device=gy.device)
        never = torch.full((n, D), 255, dtype=torch.uint8, device=gy.device)       # no arg-max term: pure g / T
        ones = torch.ones(n, T, D, dtype=td, device=gy.device)
        N.call("maxmean_bwd", dt, N.F32, gy.float().contiguous(), never, ones, g, n, T, D)
        return g.view(n, T, D)


class DownstreamEncoder(nn.Module):
    def __init__(self, config, args, base_encoder, no_of_classes):
        super().__init__()
        self.config = config
        ds = config["downstream"]
        self.output_layer = ds['finetune_layer']
        self.return_all_layers = ds["base_encoder"]['return_all_layers']
        self.interim_layer_output_shapes = ds["base_encoder"].get('interim_layer_output_shapes', [])
        self.output_dim = ds["base_encoder"]["output_dim"]
        self.encoder = base_encoder(ds["input"]["n_mels"], self.output_dim, self.return_all_layers)
        if self.output_layer == -1:
            self.final = nn.Linear(self.output_dim, no_of_classes)
        else:
            if self.return_all_layers is False:
                raise Exception("Please set return_all_layers=True in config for taking representations from any intermediate layer")
            elif len(self.interim_layer_output_shapes) < self.output_layer:
                raise Exception("Number of layers exceed number of intemediate layers")
            self.final = nn.Linear(self.interim_layer_output_shapes[self.output_layer], no_of_classes)

    def forward(self, x):
        if repr(self.encoder) != "AudioNTT2020Task6":
            raise NotImplementedError("Downstream currently supports just AudioNTT2020Task6 encoder")
        x = self.encoder(x)
        if self.return_all_layers:
            x = x[self.output_layer]
        if x.dim() == 3:
            x = MeanTFn.apply(x)                       # time pooling of the [N, T, d] embedding
        return LinearFn.apply(x, self.final.weight, self.final.bias, False)
