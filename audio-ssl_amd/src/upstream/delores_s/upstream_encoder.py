"""DELORES_S encoder wrapper (`src/upstream/delores_s/upstream_encoder.py:4-30` of the reference):
base encoder -> max over time + mean over time.  Module-level (autograd) path; the expert's fused step uses the
same kernels through `src.engine`."""
from torch import nn

from src.functional import MaxMeanFn


class DELORES_S(nn.Module):
    def __init__(self, config, base_encoder):
        super().__init__()
        be = config["pretrain"]["base_encoder"]
        self.return_all_layers = be["return_all_layers"]
        self.encoder = base_encoder(config["pretrain"]["input"]["n_mels"], be["output_dim"], self.return_all_layers)

    def forward(self, x):
        if repr(self.encoder) != "AudioNTT2020Task6":
            raise NotImplementedError("DELORES_S currently supports just AudioNTT2020Task6 encoder")
        x = self.encoder(x)
        if self.return_all_layers:
            x = x[-1]
        return MaxMeanFn.apply(x)
