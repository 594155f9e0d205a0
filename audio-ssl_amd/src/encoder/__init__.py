"""Backbones, looked up by name as the reference does: getattr(import_module('src.encoder'), cfg.base_encoder.type)
(`train_upstream.py:40-41`).  Only AudioNTT2020Task6 is on the HIP hot path this round."""
from src.encoder.audiontt import AudioNTT2020Task6  # noqa: F401
