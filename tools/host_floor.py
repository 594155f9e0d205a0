"""Diagnostic (GPU box): is the graphed bench step bound by the host launch side or by the device?
Times, per step and with the device idle at the start of each measurement: the host-side duration of the front-end
submit/collect, the host-side duration of the graph replay call, and the device duration of the replay alone."""
import copy, os, random, sys, time
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
gstep = model.graphed_step(opt)
t = front.submit(waves)
for i in range(gstep.eager_steps + 4):
    a, b = front.collect(t); t = front.submit(waves); gstep(a, b)
torch.cuda.synchronize()
hf, hg, dg = [], [], []
for i in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); a, b = front.collect(t); t = front.submit(waves); t1 = time.perf_counter()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t2 = time.perf_counter(); e0.record(); gstep(a, b); e1.record(); t3 = time.perf_counter()
    torch.cuda.synchronize()
    hf.append(t1 - t0); hg.append(t3 - t2); dg.append(e0.elapsed_time(e1))
med = lambda x: sorted(x)[len(x) // 2]
print(f"host: front-end submit+collect {med(hf)*1e3:.3f} ms, graph replay call {med(hg)*1e3:.3f} ms;  device: replay alone {med(dg):.3f} ms")
# back-to-back steady state for comparison
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(40):
    a, b = front.collect(t); t = front.submit(waves); gstep(a, b)
th = time.perf_counter() - t0; torch.cuda.synchronize(); td = time.perf_counter() - t0
print(f"steady state: {td/40*1e3:.3f} ms/step (host side {th/40*1e3:.3f})")
