"""Diagnostic (GPU box): the heads' fused weight-gradient + SGD launches (three problems 2048 x {2048, 2048, 2048} / {2048, 1024, 512},
K = 1,024 rows) under the production dispatch; AUDIOSSL_GEMM_P6=1 / 5 force the 256 x 128 / 128 x 128 tiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd"), os.path.join(ROOT, "tools")]
import torch
from src import _native as N
from src import engine as E
from gemm_shapes import timeit
torch.manual_seed(0)
D, M = 2048, 1024
for ks in ([2048] * 3, [2048, 1024, 512]):
    dys = [torch.randn(M, D, device="cuda").bfloat16() for _ in ks]
    xs = [torch.randn(M, k, device="cuda").bfloat16() for k in ks]
    P = [torch.randn(D, k, device="cuda") for k in ks]
    Mo = [torch.zeros(D, k, device="cuda") for k in ks]
    S = [torch.empty(D, k, device="cuda", dtype=torch.bfloat16) for k in ks]
    G = [torch.empty(D, k, device="cuda") for k in ks]
    t_f = timeit(lambda: E.gemm_multi_sgd(1, 1, D, ks, [M] * 3, dys, [D] * 3, xs, ks, P, Mo, S, ks, (1e-3, 0.9, 1e-4, 1.0, None)))
    t_u = timeit(lambda: E.gemm_multi(1, 1, D, ks, [M] * 3, dys, [D] * 3, xs, ks, G, ks, out_f32=1, atomic=0))
    print(f"K-widths {ks}: fused {t_f:6.1f} us ({N.last_kernel()[:40]}), plain store {t_u:6.1f} us  "
          f"(P6={os.environ.get('AUDIOSSL_GEMM_P6')})", flush=True)
