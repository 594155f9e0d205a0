"""Shared test helpers (inputs identical to tests/golden/make_goldens.py)."""
import numpy as np
import torch

from oracle import fill


def drop_mask(shape, salt, p=0.3):
    return torch.from_numpy((fill.uniform01(shape, salt) >= p).astype(np.float32))


def views(B, T, salt):
    return torch.from_numpy(fill.normalish((B, 1, 64, T), salt))


def closed_queue(emb_dim, K):
    q = torch.from_numpy(fill.normalish((emb_dim, K), 777))
    return torch.nn.functional.normalize(q, dim=0)


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).flatten()
    b = torch.as_tensor(b, dtype=torch.float64).flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def grad_digest(module):
    names, norms, heads = [], [], []
    for n, p in module.named_parameters():
        if p.grad is None:
            continue
        names.append(n)
        norms.append(float(p.grad.norm()))
        h = p.grad.flatten()[:8].detach().cpu().numpy()
        heads.append(np.pad(h, (0, 8 - h.size)))
    return names, np.array(norms), np.stack(heads)
