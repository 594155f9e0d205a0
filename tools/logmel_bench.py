"""Diagnostic (GPU box): time the log-mel front end at the bench shape (B = 512, 1 s @ 16 kHz, 64 mel) under the current
environment (AUDIOSSL_LOGMEL_V1=1 selects the round-1 kernel) and check it against the other kernel form."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd"), os.path.join(ROOT, "tools")]
import numpy as np, torch
from src.utils import MelSpectrogramLibrosa
from gemm_shapes import timeit
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = np.random.RandomState(0)
t = np.arange(16000) / 16000.0
w = torch.from_numpy((g.uniform(-0.1, 0.1, (B, 16000)) + 0.3 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)).cuda()
mel = MelSpectrogramLibrosa()
out = mel.logmel(w)
us = timeit(lambda: mel.logmel(w))
byt = B * (4.0 * 16000 + 4.0 * 64 * 101)
print(f"logmel B={B}: {us:.1f} us  {byt / us / 1e3:.1f} GB/s algorithmic ({byt / us / 1e3 / 8000:.3f} of 8 TB/s)  "
      f"checksum {float(out.double().sum()):.6f}  V1={os.environ.get('AUDIOSSL_LOGMEL_V1')}")
