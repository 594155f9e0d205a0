"""Diagnostic (GPU box): cProfile of the steady-state step loop (which call inside front.submit blocks the host?)."""
import cProfile, copy, os, pstats, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]
import numpy as np, torch
import bench
from src.augmentations import AugmentationModule
from src.dataset import UpstreamFrontEnd
from src.encoder import AudioNTT2020Task6
from src.upstream.delores_m.upstream_expert import Upstream_Expert
cfg = copy.deepcopy(bench.CFG); cfg["run"]["precision"] = "bf16"; cfg["run"]["batch_size"] = 512
B = 512; dev = torch.device("cuda", 0)
np.random.seed(31); random.seed(31); torch.manual_seed(0)
model = Upstream_Expert(cfg, base_encoder=AudioNTT2020Task6, num_negatives=65536).to(dev).train()
front = UpstreamFrontEnd(cfg, AugmentationModule(cfg, 100000, max_batch=B))
opt = model.configure_optimizers()
waves = torch.from_numpy(bench.synth_waves(B, 16000, 1234)).to(dev)
gstep = model.graphed_step(opt)
t = front.submit(waves)
for i in range(gstep.eager_steps + 6):
    a, b = front.collect(t); t = front.submit(waves); gstep(a, b)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(40):
    a, b = front.collect(t); t = front.submit(waves); gstep(a, b)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
