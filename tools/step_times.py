"""Diagnostic (GPU box): per-step completion times of bench.py's timed region (5 warm-up steps, synchronise, 20 steps), from one
event recorded on the main stream after every graph launch: where do the first replays after the synchronisation lose their time?"""
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
if os.environ.get("PREWARM"):                        # does a pre-grown allocator pool of the front-end stream remove the early bubbles?
    with torch.cuda.stream(front._stream):
        xs = [torch.empty(B, 1, 64, 96, device=dev) for _ in range(int(os.environ["PREWARM"]))]
        ys = [torch.empty(1 << 14, device=dev) for _ in range(64)]
        del xs, ys
W = int(os.environ.get("W", "5"))
for i in range(gstep.eager_steps + 1 + W):
    a, b = front.collect(t); t = front.submit(waves); gstep(a, b)
torch.cuda.synchronize()
if os.environ.get("SPIN_MS"):                        # keep the device busy right up to the timed region
    x = torch.randn(8192, 8192, device=dev)
    t_end = time.perf_counter() + float(os.environ["SPIN_MS"]) * 1e-3
    while time.perf_counter() < t_end:
        x @ x
    torch.cuda.synchronize()
import cProfile, pstats
for rep in range(int(os.environ.get("REPS", "1"))):
    pr = cProfile.Profile() if os.environ.get("PROFILE") else None
    if pr: pr.enable()
    ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
    evs, host = [], []
    t0 = time.perf_counter()
    for i in range(20):
        a, b = front.collect(t); t = front.submit(waves); gstep(a, b)
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e); host.append(time.perf_counter() - t0)
    if pr:
        pr.disable()
        pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(8)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ts = [ev0.elapsed_time(e) for e in evs]
    print("region %d: wall %.3f ms/step" % (rep, dt / 20 * 1e3))
    print("step length (ms):", " ".join("%.2f" % (b - a) for a, b in zip([0.0] + ts[:-1], ts)))
    print("host issue (ms): ", " ".join("%.2f" % (h * 1e3) for h in host))
