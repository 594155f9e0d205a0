"""Diagnostic (GPU box): time and check every GEMM shape of the DeLoRes-M step under one kernel variant
(AUDIOSSL_GEMM_RING = 10 * MI + NST, 0 = register-staged kernels only, unset = production dispatch).
Launches are timed inside a captured graph of back-to-back launches; the result is compared with torch.matmul on the same
bf16 operands."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]
import torch
from src import _native as N
from src import engine as E

def timeit(fn, reps=20):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

SHAPES = [  # mode, M, N, K, count per step, extras
    ("NT", 1024, 2048, 2048, 7, ""), ("NT", 6144, 2048, 512, 2, "bias relu keep"), ("NT", 6144, 2048, 2048, 2, "bias relu keep"),
    ("NT", 512, 2048, 2048, 3, ""), ("NT", 512, 128, 2048, 2, ""), ("NT", 512, 128, 65536, 1, "atomic"),
    ("NN", 1024, 2048, 2048, 6, "f32"), ("NN", 6144, 2048, 2048, 1, "gate"), ("NN", 512, 2048, 2048, 4, "f32"),
    ("NN", 6144, 512, 2048, 1, "f32"), ("NN", 512, 65536, 128, 1, "f32"),
    ("TN", 2048, 2048, 1024, 7, "atomic"), ("TN", 2048, 2048, 6144, 1, "atomic"), ("TN", 2048, 2048, 512, 3, "atomic"),
    ("TN", 2048, 512, 6144, 1, "atomic"), ("TN", 2048, 2048, 512, 3, "f32"),
]
def main():
    torch.manual_seed(0)
    tot = 0.0
    for mode, M, Nn, K, cnt, extra in SHAPES:
        ta, tb = {"NT": (0, 0), "NN": (0, 1), "TN": (1, 1)}[mode]
        A = torch.randn((K, M) if ta else (M, K), device="cuda").bfloat16()
        B = torch.randn((K, Nn) if tb else (Nn, K), device="cuda").bfloat16()
        atomic = "atomic" in extra
        f32 = atomic or "f32" in extra
        ks = E._ksplit(M, Nn, K, 256 if (mode == "NT" and atomic) else 512) if atomic else 1
        C = torch.zeros(M, Nn, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
        kw = {}
        if "bias" in extra:
            kw.update(bias=torch.randn(Nn, device="cuda"), relu=1, keep=(torch.rand(M, Nn, device="cuda") > 0.3).to(torch.uint8), ldk=Nn, keep_scale=1 / 0.7)
        if "gate" in extra:
            kw.update(gate=torch.randn(M, Nn, device="cuda").bfloat16(), ldg=Nn)
        run = lambda: E.gemm(1, ta, tb, M, Nn, K, A, A.shape[1], B, B.shape[1], C, Nn, out_f32=int(f32), atomic=(1 if ks > 1 else 2) if atomic else 0, ksplit=ks, **kw)
        C.zero_(); torch.cuda.synchronize(); run(); torch.cuda.synchronize()
        ref = (A.float().t() if ta else A.float()) @ (B.float() if tb else B.float().t())
        if "bias" in extra:
            ref = torch.relu(ref + kw["bias"]) * kw["keep"].float() * kw["keep_scale"]
        if "gate" in extra:
            ref = ref * (kw["gate"].float() > 0)
        err = ((C.float() - ref).abs().max() / ref.abs().max()).item()
        us = timeit(run)
        tot += us * cnt
        print(f"{mode} {M:5d}x{Nn:5d}x{K:5d} ks={ks} {extra:15s} {us:7.1f} us x{cnt}  {2.0*M*Nn*K/us/1e6:7.1f} TF/s  relerr {err:.1e}" + ("  <-- WRONG" if err > 2e-2 else ""), flush=True)
    print(f"weighted total {tot:.0f} us/step   (AUDIOSSL_GEMM_RING={os.environ.get('AUDIOSSL_GEMM_RING')})")


if __name__ == "__main__":
    main()
