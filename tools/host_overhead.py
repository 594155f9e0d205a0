"""Diagnostic (GPU box): host cost of one C-ABI launch vs GPU time of small kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]
import torch
from src import _native as N
from src import engine as E
x = torch.zeros(64, device="cuda"); y = torch.zeros(64, device="cuda", dtype=torch.bfloat16)
def loop(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    th = time.perf_counter() - t
    torch.cuda.synchronize()
    tw = time.perf_counter() - t
    return th / n * 1e6, tw / n * 1e6
print("cast(64 elems)       host %.1f us  wall %.1f us" % loop(lambda: N.call("cast", 1, x, y, 64)))
print("torch.empty          host %.1f us  wall %.1f us" % loop(lambda: torch.empty(1024, 2048, device="cuda", dtype=torch.bfloat16)))
for M, Nn, K in ((512, 128, 2048), (1024, 2048, 2048), (512, 2048, 2048)):
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(Nn, K, device="cuda").bfloat16(); C = torch.empty(M, Nn, device="cuda", dtype=torch.bfloat16)
    print(f"gemm NT {M}x{Nn}x{K}   host %.1f us  wall %.1f us" % loop(lambda: E.gemm(1, 0, 0, M, Nn, K, A, K, B, K, C, Nn), 500))
