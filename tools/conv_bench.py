"""Diagnostic (GPU box): isolated timings of the 3x3 conv kernels (forward / dgrad / wgrad) at the step's shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd"), os.path.join(ROOT, "tools")]
import torch
from src import _native as N
from gemm_shapes import timeit
ws = torch.empty(2 * 256 * 64 * 576, device="cuda")
for Nimg, Ti, Fi in ((512, 50, 32), (512, 25, 16)):
    x = torch.randn(Nimg, Ti, Fi, 64, device="cuda").bfloat16(); dy = torch.randn_like(x)
    w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05; b = torch.randn(64, device="cuda")
    Wf = torch.empty(64, 576, device="cuda", dtype=torch.bfloat16); Wd = torch.empty_like(Wf)
    N.call("pack_conv_w", 1, w, Wf, Wd)
    Y = torch.empty_like(x); dx = torch.empty(Nimg, Ti, Fi, 64, device="cuda")
    sq = torch.zeros(2, 64, dtype=torch.float64, device="cuda"); dWp = torch.zeros(64, 576, device="cuda")
    gf = 2.0 * Nimg * Ti * Fi * 64 * 576 / 1e6
    t = timeit(lambda: N.call("conv3x3_fwd", x, Wf, b, Y, 0, sq[0], sq[1], 1, Nimg, Ti, Fi)); print(f"{(Nimg,Ti,Fi)} fwd+stats {t:7.1f} us {gf/t:6.1f} TF/s")
    t = timeit(lambda: N.call("conv3x3_fwd", dy, Wd, None, dx, 1, None, None, 1, Nimg, Ti, Fi)); print(f"{(Nimg,Ti,Fi)} dgrad f32 {t:7.1f} us {gf/t:6.1f} TF/s")
    t = timeit(lambda: N.call("conv3x3_wgrad", dy, x, dWp, ws, ws.numel(), Nimg, Ti, Fi)); print(f"{(Nimg,Ti,Fi)} wgrad 2-stage {t:7.1f} us {gf/t:6.1f} TF/s")
    t = timeit(lambda: N.call("conv3x3_wgrad", dy, x, dWp, None, 0, Nimg, Ti, Fi)); print(f"{(Nimg,Ti,Fi)} wgrad atomics {t:7.1f} us {gf/t:6.1f} TF/s")
# stem: batch statistics from the 54 image moments, forward, backward
Nimg, F, T = 512, 64, 101
img = torch.randn(Nimg, F, T, device="cuda"); w = torch.randn(64, 1, 3, 3, device="cuda") * 0.3; b = torch.randn(64, device="cuda")
gm = torch.rand(64, device="cuda") + 0.5; bt = torch.randn(64, device="cuda"); rm = torch.zeros(64, device="cuda"); rv = torch.ones(64, device="cuda")
mom = torch.zeros(16 * 54, dtype=torch.float64, device="cuda"); st = torch.empty(4, 64, device="cuda")
t = timeit(lambda: N.call("conv1_stats", img, Nimg, F, T, w, b, gm, bt, rm, rv, 0.1, 1e-5, mom, st[0], st[1], st[2], st[3])); print(f"conv1_stats {t:7.1f} us")
P = torch.empty(Nimg, T // 2, F // 2, 64, device="cuda", dtype=torch.bfloat16)
t = timeit(lambda: N.call("conv1_fwd", 1, img, Nimg, F, T, w, b, st[0], st[1], P, None)); print(f"conv1_fwd (MFMA) {t:7.1f} us")
xl = torch.empty(4, Nimg, (F // 2) * 64, device="cuda")
t = timeit(lambda: N.call("conv1_fwd", 1, img, Nimg, F, T, w, b, st[0], st[1], P, xl)); print(f"conv1_fwd (MFMA) + layer mean {t:7.1f} us")
