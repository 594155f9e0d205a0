"""Minimal stand-in for the slice of `pytorch_lightning.LightningModule` the reference's experts use
(`save_hyperparameters`, `hparams`, `log_dict`, `trainer`, `load_from_checkpoint`, `configure_optimizers`).
pytorch-lightning 1.9 is not in the image; the training loop lives in `train_upstream.py` (`HipTrainer`)."""
import inspect

import torch
import torch.nn as nn


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class _NoTrainer:
    use_ddp = False
    use_ddp2 = False
    datamodule = type("dm", (), {"name": "none"})

    @property
    def world_size(self):
        return torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1


class UpstreamModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.hparams = AttrDict()
        self.trainer = _NoTrainer()
        self.logged = {}

    def save_hyperparameters(self):
        frame = inspect.currentframe().f_back                       # the __init__ that called us
        code = frame.f_code
        names = code.co_varnames[:code.co_argcount + code.co_kwonlyargcount]
        loc = frame.f_locals
        simple = (int, float, str, bool, type(None), dict, list, tuple)
        for k in names:
            if k in loc and k not in ("self", "args", "kwargs") and isinstance(loc[k], simple):
                self.hparams[k] = loc[k]

    def log_dict(self, d, *a, **k):
        self.logged.update({k_: (v.detach() if isinstance(v, torch.Tensor) else v) for k_, v in d.items()})

    def checkpoint(self, epoch=0, global_step=0):
        """Lightning-shaped checkpoint dict (`state_dict`, `hyper_parameters`, ...)."""
        return {"epoch": epoch, "global_step": global_step, "state_dict": self.state_dict(),
                "hyper_parameters": dict(self.hparams)}

    @classmethod
    def load_from_checkpoint(cls, path, strict=True, map_location="cpu", **overrides):
        ck = torch.load(path, map_location=map_location, weights_only=True)
        hp = dict(ck.get("hyper_parameters", {}))
        hp.update(overrides)
        base = hp.pop("base_encoder", None)
        if base is None or isinstance(base, str):
            import importlib
            name = base or hp["config"]["pretrain"]["base_encoder"]["type"]
            base = getattr(importlib.import_module("src.encoder"), name)
        model = cls(hp.pop("config"), base_encoder=base, **hp)
        model.load_state_dict(ck["state_dict"], strict=strict)
        return model
