"""DELORES_M encoder wrapper (`src/upstream/delores_m/upstream_encoder.py:4-36` of the reference):
base encoder (all layers) -> max+mean over time -> Linear(output_dim, contrastive_dim); passes l1, l2, l3 through."""
from torch import nn

from src.functional import LinearFn, MaxMeanFn


class DELORES_M(nn.Module):
    def __init__(self, config, base_encoder):
        super().__init__()
        be = config["pretrain"]["base_encoder"]
        self.return_all_layers = be["return_all_layers"]
        self.encoder = base_encoder(config["pretrain"]["input"]["n_mels"], be["output_dim"], self.return_all_layers)
        self.fc = nn.Linear(be["output_dim"], config["pretrain"]["contrastive_dim"])

    def forward(self, x):
        if repr(self.encoder) != "AudioNTT2020Task6":
            raise NotImplementedError("DELORES_M currently supports just AudioNTT2020Task6 encoder")
        if self.return_all_layers is False:
            raise NotImplementedError("DELORES_M need return_all_layers = True to be set in the config!")
        l1, l2, l3, x = self.encoder(x)
        x = LinearFn.apply(MaxMeanFn.apply(x), self.fc.weight, self.fc.bias, False)
        return x, l1, l2, l3
