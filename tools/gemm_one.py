"""Diagnostic (GPU box): time one bf16 GEMM shape under the current environment switches.  usage: gemm_one.py NT 6144 2048 2048"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd"), os.path.join(ROOT, "tools")]
import torch
from src import engine as E
from gemm_shapes import timeit
mode, M, Nn, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ta, tb = {"NT": (0, 0), "NN": (0, 1), "TN": (1, 1)}[mode]
torch.manual_seed(0)
A = torch.randn((K, M) if ta else (M, K), device="cuda").bfloat16()
B = torch.randn((K, Nn) if tb else (Nn, K), device="cuda").bfloat16()
C = torch.zeros(M, Nn, device="cuda", dtype=torch.bfloat16)
run = lambda: E.gemm(1, ta, tb, M, Nn, K, A, A.shape[1], B, B.shape[1], C, Nn)
run(); torch.cuda.synchronize()
ref = (A.float().t() if ta else A.float()) @ (B.float() if tb else B.float().t())
err = ((C.float() - ref).abs().max() / ref.abs().max()).item()
us = timeit(run)
print(f"{mode} {M}x{Nn}x{K}: {us:.1f} us  {2.0 * M * Nn * K / us / 1e6:.1f} TF/s  relerr {err:.1e}  env " +
      " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("AUDIOSSL_GEMM")))
