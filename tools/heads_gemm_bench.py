"""Diagnostic (GPU box): the twelve multi-problem GEMM launches of the grouped Barlow heads (B = 512: M = 1024 stacked views),
timed back to back inside a captured graph and checked against torch.matmul on the same bf16 operands.
Kernel choice follows the production dispatch; AUDIOSSL_GEMM_P6=0 / AUDIOSSL_GEMM_P8=0 in the environment switch the hand-scheduled
kernels off (run the script twice to compare)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd"), os.path.join(ROOT, "tools")]
import torch
from src import _native as N
from src import engine as E
from gemm_shapes import timeit

B, D = 512, 2048
KIN = [2048, 1024, 512]
# name, ta, tb, M, Ns, Ks, out_f32, atomic
LAUNCHES = [
    ("fwd L1   NT", 0, 0, 2 * B, [D] * 3, KIN, 0, 0),
    ("fwd L2/3 NT", 0, 0, 2 * B, [D] * 3, [D] * 3, 0, 0),
    ("dzn      NT", 0, 0, B, [D] * 3, [D] * 3, 1, 0),
    ("dzn      NN", 0, 1, B, [D] * 3, [D] * 3, 1, 0),
    ("dgrad    NN", 0, 1, 2 * B, [D] * 3, [D] * 3, 1, 0),
    ("dgrad L1 NN", 0, 1, B, KIN, [D] * 3, 1, 0),
    ("wgrad    TN", 1, 1, D, [D] * 3, [2 * B] * 3, 1, 0),
    ("wgrad L1 TN", 1, 1, D, KIN, [2 * B] * 3, 1, 0),
    ("corr     TT", 1, 1, D, [D] * 3, [B] * 3, 1, 0),
]
COUNT = {"fwd L1   NT": 1, "fwd L2/3 NT": 2, "dzn      NT": 1, "dzn      NN": 1, "dgrad    NN": 2, "dgrad L1 NN": 1,
         "wgrad    TN": 2, "wgrad L1 TN": 1, "corr     TT": 1}


def main():
    torch.manual_seed(0)
    tot_us, tot_fl = 0.0, 0.0
    for name, ta, tb, M, Ns, Ks, f32, atomic in LAUNCHES:
        As = [torch.randn((k, M) if ta else (M, k), device="cuda").bfloat16() for k in Ks]
        Bs = [torch.randn((k, n) if tb else (n, k), device="cuda").bfloat16() for n, k in zip(Ns, Ks)]
        Cs = [torch.zeros(M, n, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16) for n in Ns]
        run = lambda: E.gemm_multi(ta, tb, M, Ns, Ks, As, [a.shape[1] for a in As], Bs, [b.shape[1] for b in Bs], Cs, Ns,
                                   out_f32=f32, atomic=atomic)
        run(); torch.cuda.synchronize()
        err = 0.0
        for a, b, c in zip(As, Bs, Cs):
            ref = (a.float().t() if ta else a.float()) @ (b.float() if tb else b.float().t())
            err = max(err, ((c.float() - ref).abs().max() / ref.abs().max()).item())
        us = timeit(run)
        fl = sum(2.0 * M * n * k for n, k in zip(Ns, Ks))
        tot_us += us * COUNT[name]; tot_fl += fl * COUNT[name]
        print(f"{name}  M={M:5d} N={Ns} K={Ks}: {us:7.1f} us  {fl / us / 1e6:7.1f} TF/s  relerr {err:.1e}"
              + ("  <-- WRONG" if err > 2e-2 else ""), flush=True)
    print(f"heads GEMM total {tot_us:.0f} us/step, {tot_fl / tot_us / 1e6:.0f} TF/s "
          f"(P6={os.environ.get('AUDIOSSL_GEMM_P6')}, P8={os.environ.get('AUDIOSSL_GEMM_P8')})")


if __name__ == "__main__":
    main()
