"""Diagnostic (GPU box): launch ONE GEMM shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py NT|NN|TN M N K [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]
import torch
from src import engine as E
mode, M, Nn, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
ta, tb = {"NT": (0, 0), "NN": (0, 1), "TN": (1, 1)}[mode]
A = torch.randn((K, M) if ta else (M, K), device="cuda").bfloat16()
B = torch.randn((K, Nn) if tb else (Nn, K), device="cuda").bfloat16()
C = torch.zeros(M, Nn, device="cuda", dtype=torch.bfloat16)
for _ in range(reps):
    E.gemm(1, ta, tb, M, Nn, K, A, A.shape[1], B, B.shape[1], C, Nn)
torch.cuda.synchronize()
