"""Backbones, looked up by name as the reference does: getattr(import_module('src.encoder'), cfg.base_encoder.type)
(`train_upstream.py:40-41`): the BYOL-A conv encoder (delores / slicer / decar paths) and the AST-base transformer (MAST)."""
from src.encoder.audiontt import AudioNTT2020Task6  # noqa: F401
from src.encoder.mast import MAST, ASTModel  # noqa: F401
