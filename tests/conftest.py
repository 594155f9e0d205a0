import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "audio-ssl_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


CFG_S = {"run": {"batch_size": 16, "world_size": 1, "num_dataloader_workers": 0, "save_path": "/tmp/audiossl_s/"},
         "pretrain": {"base_encoder": {"type": "AudioNTT2020Task6", "output_dim": 2048, "return_all_layers": False},
                      "projection_dim": 2048, "normalization": "mean_var", "lambda_barlow": 5e-5,
                      "input": {"type": "raw_wav", "sampling_rate": 16000, "length_wave": 1.0, "n_mels": 64},
                      "augmentations": {"MixupBYOLA": {"ratio": 0.4, "log_mixup_exp": True},
                                        "RandomResizeCrop": {"virtual_crop_scale": [1.0, 1.5],
                                                             "freq_crop_scale": [0.6, 1.5],
                                                             "time_crop_scale": [0.6, 1.5]}}}}
CFG_M = {"run": dict(CFG_S["run"], save_path="/tmp/audiossl_m/"),
         "pretrain": dict(CFG_S["pretrain"], contrastive_dim=128, lambda_barlow=[5e-5, 5e-5, 5e-5], loss_scale="1/32",
                          base_encoder={"type": "AudioNTT2020Task6", "output_dim": 2048, "return_all_layers": True})}


@pytest.fixture(scope="session")
def cfg_s():
    import copy
    return copy.deepcopy(CFG_S)


@pytest.fixture(scope="session")
def cfg_m():
    import copy
    return copy.deepcopy(CFG_M)
