"""Diagnostic (GPU box): isolated timing of the GEMM shapes of the delores_m step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]
import torch
from src import _native as N
from src import engine as E
shapes = [("NT", 6144, 2048, 2048), ("NT", 6144, 2048, 512), ("NT", 1024, 2048, 2048), ("NT", 512, 2048, 2048), ("NT", 512, 128, 2048),
          ("NN", 6144, 2048, 2048), ("NN", 1024, 2048, 2048), ("NN", 512, 2048, 2048), ("NN", 512, 65536, 128), ("NN", 6144, 512, 2048),
          ("TN", 2048, 2048, 6144), ("TN", 2048, 2048, 1024), ("TN", 2048, 512, 6144), ("TN", 2048, 2048, 512), ("NT", 512, 128, 65536)]
dt = 1
for mode, M, Nn, K in shapes:
    ta, tb = {"NT": (0, 0), "NN": (0, 1), "TN": (1, 1)}[mode]
    A = torch.randn((K, M) if ta else (M, K), device="cuda").bfloat16()
    B = torch.randn((K, Nn) if tb else (Nn, K), device="cuda").bfloat16()
    atomic = mode == "TN" or (mode == "NT" and K == 65536)
    ks = E._ksplit(M, Nn, K) if atomic else 1
    C = torch.zeros(M, Nn, device="cuda", dtype=torch.float32 if atomic else torch.bfloat16)
    def run():
        E.gemm(dt, ta, tb, M, Nn, K, A, A.shape[1], B, B.shape[1], C, Nn, out_f32=int(atomic), atomic=int(atomic), ksplit=ks)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{mode} M={M:6d} N={Nn:6d} K={K:6d} ksplit={ks:3d}  {us:8.1f} us  {2.0*M*Nn*K/us/1e6:7.1f} TF/s", flush=True)
