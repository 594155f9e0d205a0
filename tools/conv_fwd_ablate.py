"""Diagnostic (GPU box): forward 3x3 conv at the step's shapes; run under AUDIOSSL_CONV_DBG=<bits> (1 no stores, 2 no halo loads,
4 no MFMA loop) to see which part of the tile loop the time goes to."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd"), os.path.join(ROOT, "tools")]
import torch
from src import _native as N
from gemm_shapes import timeit
for Nimg, Ti, Fi in ((512, 50, 32), (512, 25, 16)):
    x = torch.randn(Nimg, Ti, Fi, 64, device="cuda").bfloat16()
    w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05; b = torch.randn(64, device="cuda")
    Wf = torch.empty(64, 576, device="cuda", dtype=torch.bfloat16); Wd = torch.empty_like(Wf)
    N.call("pack_conv_w", 1, w, Wf, Wd)
    Y = torch.empty_like(x); dx = torch.empty(Nimg, Ti, Fi, 64, device="cuda")
    sq = torch.zeros(2, 64, dtype=torch.float64, device="cuda")
    gf = 2.0 * Nimg * Ti * Fi * 64 * 576 / 1e6
    t = timeit(lambda: N.call("conv3x3_fwd", x, Wf, b, Y, 0, sq[0], sq[1], 1, Nimg, Ti, Fi))
    t2 = timeit(lambda: N.call("conv3x3_fwd", x, Wd, None, dx, 1, None, None, 1, Nimg, Ti, Fi))
    t3 = timeit(lambda: N.call("conv3x3_fwd", x, Wf, b, Y, 0, None, None, 1, Nimg, Ti, Fi))
    print(f"dbg={os.environ.get('AUDIOSSL_CONV_DBG', '0')} ws={os.environ.get('AUDIOSSL_CONV_WS', '-')} {(Nimg, Ti, Fi)} fwd+stats {t:7.1f} us {gf / t:6.1f} TF/s | "
          f"dgrad f32 {t2:7.1f} us | fwd {t3:7.1f} us", flush=True)
