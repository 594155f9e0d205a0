"""Diagnostic (GPU box): the forward 3x3 conv (bf16 out + statistics) at B = 512, block 2, a few dozen launches - for rocprofv3 --pmc."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]
import torch
from src import _native as N
Nimg, Ti, Fi = 512, 50, 32
x = torch.randn(Nimg, Ti, Fi, 64, device="cuda").bfloat16()
w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05; b = torch.randn(64, device="cuda")
Wf = torch.empty(64, 576, device="cuda", dtype=torch.bfloat16); Wd = torch.empty_like(Wf)
N.call("pack_conv_w", 1, w, Wf, Wd)
Y = torch.empty_like(x)
sq = torch.zeros(2, 64, dtype=torch.float64, device="cuda")
for _ in range(30):
    N.call("conv3x3_fwd", x, Wf, b, Y, 0, sq[0], sq[1], 1, Nimg, Ti, Fi)
torch.cuda.synchronize()
