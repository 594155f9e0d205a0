"""Diagnostic (GPU box): where does the HOST spend its time per step in steady state (no synchronisation added)?"""
import copy, os, random, sys, time
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
parts = {"collect": 0.0, "submit": 0.0, "copy": 0.0, "replay": 0.0}
n = 60
t00 = time.perf_counter()
for i in range(n):
    t0 = time.perf_counter(); a, b = front.collect(t)
    t1 = time.perf_counter(); t = front.submit(waves)
    t2 = time.perf_counter(); gstep.in_1.copy_(a); gstep.in_2.copy_(b)
    t3 = time.perf_counter(); gstep.graphs[gstep.replays % len(gstep.graphs)].replay(); gstep.replays += 1
    t4 = time.perf_counter()
    parts["collect"] += t1 - t0; parts["submit"] += t2 - t1; parts["copy"] += t3 - t2; parts["replay"] += t4 - t3
th = time.perf_counter() - t00
torch.cuda.synchronize(); td = time.perf_counter() - t00
print({k: round(v / n * 1e3, 3) for k, v in parts.items()}, "host total %.3f ms/step, device %.3f ms/step" % (th / n * 1e3, td / n * 1e3))
if os.environ.get("HOST_PARTS_PROFILE", "1") == "1":
    # where inside submit does the host wait?  (cProfile over the same loop: a call that blocks on the device shows its wait as tottime)
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n):
        a, b = front.collect(t); t = front.submit(waves); gstep(a, b)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(14)

