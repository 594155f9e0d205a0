"""Diagnostic (GPU box): dump conv3x3 forward / dgrad outputs for the shapes of the golden-step tests so that two kernel
variants (AUDIOSSL_CONV_WAVES=4 / 8, separate processes) can be compared bit for bit; also checks run-to-run equality."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]
import torch
from src import _native as N
out = {}
for Nimg, Ti, Fi in ((16, 50, 32), (16, 25, 16), (1024, 50, 32)):
    g = torch.Generator().manual_seed(Ti)
    x = torch.randn(Nimg, Ti, Fi, 64, generator=g).cuda().bfloat16()
    w = (torch.rand(64, 64, 3, 3, generator=g) * 0.1 - 0.05).cuda()
    b = torch.randn(64, generator=g).cuda()
    Wf = torch.empty(64, 576, device="cuda", dtype=torch.bfloat16); Wd = torch.empty_like(Wf)
    N.call("pack_conv_w", 1, w, Wf, Wd)
    res = []
    for rep in range(3):
        Y = torch.empty(Nimg, Ti, Fi, 64, device="cuda", dtype=torch.bfloat16)
        sq = torch.zeros(2, 64, dtype=torch.float64, device="cuda")
        N.call("conv3x3_fwd", x, Wf, b, Y, 0, sq[0], sq[1], 1, Nimg, Ti, Fi)
        dx = torch.empty(Nimg, Ti, Fi, 64, device="cuda", dtype=torch.float32)
        N.call("conv3x3_fwd", x, Wd, None, dx, 1, None, None, 1, Nimg, Ti, Fi)
        torch.cuda.synchronize()
        res.append((Y.clone(), dx.clone(), sq.clone()))
    same = all(torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1]) for r in res[1:])
    print(f"{(Nimg, Ti, Fi)} run-to-run identical: {same}; sq rel spread {float(((res[0][2]-res[1][2]).abs()/res[0][2].abs()).max()):.1e}")
    if Nimg <= 16:
        out[f"Y{Ti}"] = res[0][0].cpu(); out[f"dx{Ti}"] = res[0][1].cpu(); out[f"sq{Ti}"] = res[0][2].cpu()
tag = os.environ.get("AUDIOSSL_CONV_WAVES", "8")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
torch.save(out, f"/tmp/convw_{tag}.pt")
other = f"/tmp/convw_{'4' if tag == '8' else '8'}.pt"
if os.path.exists(other):
    o = torch.load(other)
    for k in out:
        print(k, "bit-identical across variants:", torch.equal(out[k], o[k]), "max abs diff", float((out[k].double() - o[k].double()).abs().max()))
