"""Diagnostic (GPU box): cProfile of the host side of `UpstreamFrontEnd.submit` (the eager front end of the next batch) in steady state."""
import copy, cProfile, os, pstats, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]
import numpy as np, torch
import bench
from src.augmentations import AugmentationModule
from src.dataset import UpstreamFrontEnd
cfg = copy.deepcopy(bench.CFG); cfg["run"]["precision"] = "bf16"; cfg["run"]["batch_size"] = 512
B = 512; dev = torch.device("cuda", 0)
np.random.seed(31); random.seed(31); torch.manual_seed(0)
front = UpstreamFrontEnd(cfg, AugmentationModule(cfg, 100000, max_batch=B))
waves = torch.from_numpy(bench.synth_waves(B, 16000, 1234)).to(dev)
for i in range(10):
    t = front.submit(waves); front.collect(t)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(100):
    t = front.submit(waves)
    if i % 4 == 3:
        torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout).sort_stats("cumulative")
st.print_stats(35)
