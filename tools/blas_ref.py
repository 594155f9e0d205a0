"""Diagnostic (GPU box): what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) does on the step's plain GEMM shapes,
timed like tools/gemm_shapes.py (graph of back-to-back launches).  Reference point for the hand-written kernels only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd"), os.path.join(ROOT, "tools")]
import torch
from gemm_shapes import timeit
for mode, M, Nn, K in (("NT", 1024, 2048, 2048), ("NN", 1024, 2048, 2048), ("TN", 2048, 2048, 1024), ("NT", 6144, 2048, 2048),
                       ("NN", 6144, 2048, 2048), ("TN", 2048, 2048, 6144), ("NT", 6144, 2048, 512), ("NN", 6144, 512, 2048),
                       ("NT", 512, 2048, 2048), ("TN", 2048, 2048, 512), ("NT", 55296, 768, 768), ("NT", 55296, 3072, 768)):
    ta, tb = {"NT": (0, 0), "NN": (0, 1), "TN": (1, 1)}[mode]
    A = torch.randn((K, M) if ta else (M, K), device="cuda").bfloat16()
    B = torch.randn((K, Nn) if tb else (Nn, K), device="cuda").bfloat16()
    a = A.t() if ta else A
    b = B if tb else B.t()
    out = torch.empty(M, Nn, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: torch.matmul(a, b, out=out))
    print(f"{mode} {M:5d}x{Nn:5d}x{K:5d}  torch.matmul {us:7.1f} us  {2.0*M*Nn*K/us/1e6:7.1f} TF/s", flush=True)
