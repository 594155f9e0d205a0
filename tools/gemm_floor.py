"""Diagnostic (GPU box): where does the ~35 us floor of small GEMM launches come from?  Times single launches by HIP events
inside a captured graph of 50 back-to-back launches (no host in the loop)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]
import torch
from src import _native as N
from src import engine as E
def timeit(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
x = torch.zeros(64, device="cuda"); y = torch.zeros(64, device="cuda", dtype=torch.bfloat16)
print("cast(64)                 %.1f us" % timeit(lambda: N.call("cast", 1, x, y, 64)))
for M, Nn, K in ((128, 128, 64), (128, 128, 2048), (512, 128, 2048), (1024, 2048, 64), (1024, 2048, 512), (1024, 2048, 2048), (6144, 2048, 64), (6144, 2048, 2048)):
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(Nn, K, device="cuda").bfloat16()
    C = torch.empty(M, Nn, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: E.gemm(1, 0, 0, M, Nn, K, A, K, B, K, C, Nn))
    print(f"gemm NT {M}x{Nn}x{K}   {us:.1f} us   {2.0*M*Nn*K/us/1e6:.1f} TF/s", flush=True)
