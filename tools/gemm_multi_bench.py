"""Diagnostic (GPU box): three independent GEMMs of the Barlow heads as three launches vs one multi-problem launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]
import torch
from src import _native as N
from src import engine as E
from gemm_shapes import timeit
for mode, M, Nn, K, f32, atomic in (("NT", 1024, 2048, 2048, 0, 0), ("NN", 1024, 2048, 2048, 1, 0), ("TN", 2048, 2048, 1024, 1, 1),
                                    ("TN", 2048, 2048, 512, 1, 0), ("NT", 512, 2048, 2048, 1, 0), ("NN", 512, 2048, 2048, 1, 0)):
    ta, tb = {"NT": (0, 0), "NN": (0, 1), "TN": (1, 1)}[mode]
    As = [torch.randn((K, M) if ta else (M, K), device="cuda").bfloat16() for _ in range(3)]
    Bs = [torch.randn((K, Nn) if tb else (Nn, K), device="cuda").bfloat16() for _ in range(3)]
    Cs = [torch.zeros(M, Nn, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16) for _ in range(3)]
    def single():
        for h in range(3):
            E.gemm(1, ta, tb, M, Nn, K, As[h], As[h].shape[1], Bs[h], Bs[h].shape[1], Cs[h], Nn, out_f32=f32, atomic=2 if atomic else 0)
    def multi():
        E.gemm_multi(ta, tb, M, Nn, [K] * 3, As, [a.shape[1] for a in As], Bs, [b.shape[1] for b in Bs], Cs, Nn, out_f32=f32, atomic=2 if atomic else 0)
    print(f"{mode} {M}x{Nn}x{K}: 3 launches {timeit(single):.1f} us, one multi launch {timeit(multi):.1f} us", flush=True)
