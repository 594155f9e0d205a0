"""Diagnostic (GPU box): cProfile of the launch side of the bench step (where does the host time go)."""
import cProfile, copy, os, pstats, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]
import numpy as np, torch
import bench
from src import _native as N
from src.augmentations import AugmentationModule
from src.dataset import UpstreamFrontEnd
from src.encoder import AudioNTT2020Task6
from src.upstream.delores_m.upstream_expert import Upstream_Expert
cfg = copy.deepcopy(bench.CFG); cfg["run"]["precision"] = "bf16"; cfg["run"]["batch_size"] = 512
B = 512; dev = torch.device("cuda", 0)
np.random.seed(31); random.seed(31); torch.manual_seed(0)
model = Upstream_Expert(cfg, base_encoder=AudioNTT2020Task6, num_negatives=65536).to(dev).train()
tfms = AugmentationModule(cfg, 100000, max_batch=B); front = UpstreamFrontEnd(cfg, tfms)
opt = model.configure_optimizers()
waves = torch.from_numpy(bench.synth_waves(B, 16000, 1234)).to(dev)
def step(i):
    a, b = front(waves); opt.zero_grad(); loss = model.training_step((a, b), i); loss.backward(); model.all_reduce_grads(); opt.step()
for i in range(5): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(20): step(5 + i)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
