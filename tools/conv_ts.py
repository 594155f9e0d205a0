"""Diagnostic (GPU box): per-tile cycle stamps of workgroup 0 / wave 0 of the weights-stationary conv kernel
(AUDIOSSL_CONV_DBG=16): tile start, end of k-loop, halo DMA landed, epilogue issued."""
import os, sys
os.environ["AUDIOSSL_CONV_DBG"] = str(16 | int(os.environ.get("AUDIOSSL_CONV_DBG", "0")))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]
import torch
from src import _native as N
Nimg, Ti, Fi = 512, 50, 32
x = torch.randn(Nimg, Ti, Fi, 64, device="cuda").bfloat16()
w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
Wf = torch.empty(64, 576, device="cuda", dtype=torch.bfloat16); Wd = torch.empty_like(Wf)
N.call("pack_conv_w", 1, w, Wf, Wd)
Y = torch.empty_like(x)
ts = torch.zeros(128, dtype=torch.float64, device="cuda")
for _ in range(3):
    N.call("conv3x3_fwd", x, Wf, None, Y, 0, ts[:64], ts[64:], 1, Nimg, Ti, Fi)
torch.cuda.synchronize()
t = ts.view(torch.int64).cpu().numpy()[:52].reshape(13, 4)
print("tile  kloop  dma_wait  epilogue  barrier+next  (cycles)")
for i in range(13):
    nxt = t[i + 1, 0] if i < 12 else t[i, 3]
    print(f"{i:3d} {t[i,1]-t[i,0]:7d} {t[i,2]-t[i,1]:8d} {t[i,3]-t[i,2]:8d} {nxt-t[i,3]:8d}")
print("total cycles", t[12, 3] - t[0, 0])
r = ts.view(torch.int64).cpu().numpy()
for name, o in (("first workgroup", 64), ("last workgroup", 72)):
    e, l0, l1, w = r[o], r[o + 1], r[o + 2], r[o + 3]
    print(f"{name}: entry at {(e - r[64]) * 10} ns; weights in registers +{(w - e) * 10} ns; loop starts +{(l0 - e) * 10} ns; loop {(l1 - l0) * 10} ns")
print("s_memtime runs at", (t[12, 3] - t[0, 0]) / ((r[66] - r[65]) * 10.0), "GHz")
