"""Diagnostic (GPU box): the front-end launches (log-mel, clip moments, RunningNorm scan, normalise, two views) at B = 512 - run under
tools/kstats.py for per-kernel times."""
import copy, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]
import numpy as np
import torch
import bench
from src.augmentations import AugmentationModule
from src.dataset import UpstreamFrontEnd
cfg = copy.deepcopy(bench.CFG)
B = 512
np.random.seed(1); random.seed(1)
tfms = AugmentationModule(cfg, 100000, max_batch=B)
fe = UpstreamFrontEnd(cfg, tfms)
waves = torch.randn(B, 16000, device="cuda") * 0.1
for _ in range(12):
    a, b = fe(waves)
torch.cuda.synchronize()
print("views", tuple(a.shape), float(a.abs().mean()))
