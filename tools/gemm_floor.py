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
print("---- NN (dX = dY W) and TN (dW += dY^T X, split-K atomics) shapes of the step")
for mode, M, Nn, K in (("NN", 1024, 2048, 2048), ("NN", 512, 2048, 2048), ("NN", 6144, 2048, 2048), ("TN", 2048, 2048, 1024), ("TN", 2048, 2048, 512),
                       ("TN", 2048, 2048, 6144), ("TN", 2048, 2048, 64)):
    ta, tb = {"NN": (0, 1), "TN": (1, 1)}[mode]
    A = torch.randn((K, M) if ta else (M, K), device="cuda").bfloat16()
    B = torch.randn((K, Nn) if tb else (Nn, K), device="cuda").bfloat16()
    atomic = mode == "TN"
    ks = E._ksplit(M, Nn, K) if atomic else 1
    C = torch.zeros(M, Nn, device="cuda", dtype=torch.float32 if atomic else torch.bfloat16)
    us = timeit(lambda: E.gemm(1, ta, tb, M, Nn, K, A, A.shape[1], B, B.shape[1], C, Nn, out_f32=int(atomic), atomic=int(atomic), ksplit=ks))
    print(f"gemm {mode} {M}x{Nn}x{K} ksplit={ks}   {us:.1f} us   {2.0*M*Nn*K/us/1e6:.1f} TF/s", flush=True)
print("---- TN 2048x2048x1024: split-K / atomic variants")
M, Nn, K = 2048, 2048, 1024
A = torch.randn(K, M, device="cuda").bfloat16(); B = torch.randn(K, Nn, device="cuda").bfloat16()
C = torch.zeros(M, Nn, device="cuda", dtype=torch.float32)
for ks, atomic in ((1, 0), (1, 1), (2, 1), (4, 1)):
    us = timeit(lambda: E.gemm(1, 1, 1, M, Nn, K, A, M, B, Nn, C, Nn, out_f32=1, atomic=atomic, ksplit=ks))
    print(f"ksplit={ks} atomic={atomic}   {us:.1f} us", flush=True)
print("---- TN split-K variants for the long-K weight gradients of the encoder fc layers")
for M, Nn, K in ((2048, 2048, 6144), (2048, 512, 6144), (2048, 2048, 1024), (2048, 1024, 1024), (2048, 512, 1024)):
    A = torch.randn(K, M, device="cuda").bfloat16(); B = torch.randn(K, Nn, device="cuda").bfloat16()
    C = torch.zeros(M, Nn, device="cuda", dtype=torch.float32)
    row = []
    for ks in (1, 2, 4, 8):
        us = timeit(lambda: E.gemm(1, 1, 1, M, Nn, K, A, M, B, Nn, C, Nn, out_f32=1, atomic=1, ksplit=ks), 20)
        row.append(f"ks={ks}: {us:6.1f}")
    print(f"TN {M}x{Nn}x{K}  " + "  ".join(row) + f"   (_ksplit -> {E._ksplit(M, Nn, K)})", flush=True)
