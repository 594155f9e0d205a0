"""DeepCluster-v2 (DECAR-v2) model on MI355X - `extras/decar-v2/models_delores.py:33-122` and `MultiPrototypes`
(`extras/decar-v2/utils.py:134-145`) of the reference.

Same constructor, `forward(batch) -> (embedding of view 1, [prototype scores of view 2])` and state_dict keys
(`features.{0,1,4,5,8,9}.*`, `fc.{0,3}.*`, `projection_head.{0,1,3}.*`, `prototypes.prototypes<i>.weight`), so the
reference's checkpoints load.  The encoder is the same HIP launch sequence as `src.encoder.AudioNTT2020Task6` (the
reference writes it as one `features` Sequential here; a key map bridges the two namings); projection head and
prototypes are MFMA GEMMs with the BatchNorm1d + ReLU kernels in between.
"""
import torch
from torch import nn

from src import _native as N
from src import engine as E
from src.encoder.audiontt import DropoutMasks, _EncoderFn, default_precision
from src.functional import BatchNorm1dFn, LinearFn, MaxMeanFn

_TO_ENGINE = {"features.0.": "features_1.0.", "features.1.": "features_1.1.", "features.4.": "features_2.0.",
              "features.5.": "features_2.1.", "features.8.": "features_3.0.", "features.9.": "features_3.1.", "fc.": "fc."}


class MultiPrototypes(nn.Module):
    def __init__(self, output_dim, nmb_prototypes):
        super().__init__()
        self.nmb_heads = len(nmb_prototypes)
        for i, k in enumerate(nmb_prototypes):
            self.add_module("prototypes" + str(i), nn.Linear(output_dim, k, bias=False))

    def forward(self, x):
        return [LinearFn.apply(x, getattr(self, "prototypes" + str(i)).weight, None, False) for i in range(self.nmb_heads)]


class AudioNTT2020(nn.Module):
    """BYOL-A encoder + projection head + prototypes (the reference subclasses its Task6 network)."""

    def __init__(self, args, out_dim, n_mels=64, d=512, nmb_prototypes=3000):
        super().__init__()
        self.args = args
        layers = []
        for cin in (1, 64, 64):
            layers += [nn.Conv2d(cin, 64, 3, stride=1, padding=1), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(2, stride=2)]
        self.features = nn.Sequential(*layers)
        self.fc = nn.Sequential(nn.Linear(64 * (n_mels // (2 ** 3)), d), nn.ReLU(), nn.Dropout(p=0.3), nn.Linear(d, d), nn.ReLU())
        self.d = d
        self.projection_head = nn.Sequential(nn.Linear(d, 2048), nn.BatchNorm1d(2048), nn.ReLU(inplace=True),
                                             nn.Linear(2048, out_dim))
        if isinstance(args.nmb_prototypes, list):
            # the reference hard-codes one head of 1,024 prototypes here whatever the list says; `prototype_sizes` (not in
            # the reference) lets small test banks use fewer
            self.prototypes = MultiPrototypes(out_dim, list(getattr(args, "prototype_sizes", None) or [1024]))
        elif args.nmb_prototypes > 0:
            self.prototypes = nn.Linear(out_dim, 1024, bias=False)
        self.precision = default_precision()
        self.dropout_masks = DropoutMasks(0.3)

    # ---- bridge to the encoder launch sequences (src/engine.py keys parameters `features_1.0.weight` ...)
    def _encoder_items(self):
        out = []
        for n, t in list(self.named_parameters()) + list(self.named_buffers()):
            for a, b in _TO_ENGINE.items():
                if n.startswith(a):
                    out.append((b + n[len(a):], n, t))
        return out

    def param_dict(self):
        p0 = self.features[0].weight
        c = self.__dict__.get("_enc_cache")
        if c is None or c[0] != p0.data_ptr():
            c = self.__dict__["_enc_cache"] = (p0.data_ptr(), {k: (t.data if isinstance(t, nn.Parameter) else t)
                                                                for k, _, t in self._encoder_items()})
        return c[1]

    def engine_param_names(self):
        return [k for k, _, t in self._encoder_items() if isinstance(t, nn.Parameter)]

    def encoder_parameters(self):
        return tuple(t for _, _, t in self._encoder_items() if isinstance(t, nn.Parameter))

    def encode(self, x):
        """[B, 1, n_mels, T] -> [B, T/8, d] (activation dtype)."""
        if not x.is_cuda:
            raise RuntimeError("AudioNTT2020 (HIP) needs a GPU tensor - there is no CPU fallback")
        x = x.float().contiguous()
        keep = self.dropout_masks.next(x.shape[0] * (x.shape[-1] // 8), self.d, x.device) if self.training else None
        params = self.encoder_parameters()
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _EncoderFn.apply(self, x, keep, *params)[3]
        return E.encoder_forward(self.param_dict(), x, self.precision, keep=keep, p_drop=self.fc[2].p, train=self.training)[3]

    def project(self, z):
        ph = self.projection_head
        act = z.dtype
        a = LinearFn.apply(z, ph[0].weight, ph[0].bias, False)
        h = BatchNorm1dFn.apply(a, ph[1].weight, ph[1].bias, ph[1].running_mean, ph[1].running_var, self.precision, True,
                                self.training)
        return LinearFn.apply(h.to(act), ph[3].weight, ph[3].bias, False)

    def forward(self, batch):
        z = MaxMeanFn.apply(self.encode(batch[0]))          # first augmentation
        z_new = MaxMeanFn.apply(self.encode(batch[1]))      # second augmentation
        x = self.project(z)
        x_new = self.project(z_new)
        protos = self.prototypes(x_new.to(z.dtype)) if isinstance(self.prototypes, MultiPrototypes) else \
            LinearFn.apply(x_new.to(z.dtype), self.prototypes.weight, None, False)
        return x, protos
