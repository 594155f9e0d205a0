"""Diagnostic (GPU box): throughput of the SS-MAST step (AST-base 12 x 768, 128 mel x 101 frames -> 108 patches) on one GPU."""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]
import torch
from src.upstream.ssmast.upstream_expert import Upstream_Expert
CFG = {"run": {"batch_size": 8, "precision": "bf16"},
       "pretrain": {"base_encoder": {"type": "MAST", "output_dim": 768, "depth": 12, "num_heads": 12, "fstride": 10, "tstride": 10,
                                     "return_all_layers": False}, "normalization": "mean_var",
                    "input": {"type": "raw_wav", "sampling_rate": 16000, "length_wave": 1.0, "n_mels": 128}}}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
graph = (sys.argv[3] != "eager") if len(sys.argv) > 3 else True
T = int(sys.argv[4]) if len(sys.argv) > 4 else 101          # frames: 101 = 1 s (108 patches), 1001 = 10 s (1,188 patches: block-walking attention)
CFG["pretrain"]["input"]["length_wave"] = (T - 1) / 100.0
if len(sys.argv) > 5 and sys.argv[5] == "mvit":            # the MViTv2-B encoder SS-MAST instantiates (pooling attention, 24 blocks)
    CFG["pretrain"]["base_encoder"]["model_size"] = "mvit"
torch.manual_seed(0)
ex = Upstream_Expert(copy.deepcopy(CFG), num_negatives=65536).cuda().train()
opt = ex.configure_optimizers()
a, b = torch.randn(B, 1, 128, T, device="cuda"), torch.randn(B, 1, 128, T, device="cuda")
if graph:
    step = ex.graphed_step(opt, eager_steps=1)
else:
    def step(a, b):
        opt.zero_grad(); l = ex.training_step((a, b), 0); l.backward(); opt.step(); return l
for _ in range(3): loss = step(a, b)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps): loss = step(a, b)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
flops = 2 * B * 3 * 2 * (108 * (256 * 768 + 12 * (4 * 768 * 768 + 2 * 768 * 3072 + 2 * 108 * 768)))   # 2 q passes fwd+bwd(x3) ... approx
print(f"B={B} T={T} graph={graph}: {dt * 1e3:.2f} ms/step, {B / dt:.0f} clips/s, loss {float(loss):.4f}, mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
