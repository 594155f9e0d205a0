"""SLICER encoder wrapper (`src/upstream/slicer/upstream_encoder.py:4-35` of the reference): base encoder ->
max + mean over time -> instance projector Linear(d, 128) and cluster projector Linear(d, d)-ReLU-Linear(d, K)-Softmax.
Same attribute / state_dict names (`encoder`, `instance_projector`, `cluster_projector.{0,2}`); every op runs through
the C ABI (encoder launch sequences, MFMA GEMMs with fused bias/ReLU, the row-softmax kernel)."""
import torch
from torch import nn

from src.functional import LinearFn, MaxMeanFn, SoftmaxRowsFn


class SLICER(nn.Module):
    def __init__(self, config, base_encoder):
        super().__init__()
        pre = config["pretrain"]
        d = pre["base_encoder"]["output_dim"]
        self.encoder = base_encoder(pre["input"]["n_mels"], d, pre["base_encoder"]["return_all_layers"])
        self.instance_projector = nn.Linear(d, pre["instance_contrastive_dim"])
        self.cluster_projector = nn.Sequential(nn.Linear(d, d), nn.ReLU(), nn.Linear(d, pre["cluster_contrastive_dim"]),
                                               nn.Softmax(dim=1))

    def forward(self, x):
        if repr(self.encoder) != "AudioNTT2020Task6":
            raise NotImplementedError("SLICER currently supports just AudioNTT2020Task6 encoder")
        h = self.encoder(x)
        if isinstance(h, tuple):
            h = h[-1]
        y = MaxMeanFn.apply(h)                                       # [B, d], activation dtype
        act = y.dtype
        x_instance = LinearFn.apply(y, self.instance_projector.weight, self.instance_projector.bias, False)
        c0, c2 = self.cluster_projector[0], self.cluster_projector[2]
        hid = LinearFn.apply(y, c0.weight, c0.bias, True)            # Linear + ReLU fused in the GEMM epilogue
        logits = LinearFn.apply(hid.to(act), c2.weight, c2.bias, False)
        return x_instance, SoftmaxRowsFn.apply(logits)
