"""Oracle: DeepCluster-v2 spherical k-means (test infrastructure).

CPU restatement of `extras/decar-v2/utils.py:276-346` (cluster_memory) for one head / one crop: E step
`mm(mem, centroids.t()).max(1)`, M step per-cluster sum (the reference goes through scipy csr_matrix on the CPU),
optional all-reduce hooks, `centroids[mask] = sums[mask] / counts[mask]` (empty clusters keep their centroid),
`F.normalize`.  PINNED (round 3): tests/golden/kmeans_ref.npz holds the outputs of the reference function itself, run on the CPU
by tests/golden/make_goldens.py g16 (no-op `.cuda`, 1-process gloo group) - seeds, centroids after every iteration, assignments,
an empty-cluster case; tests/test_oracle_golden.py checks this file against it.  The collective logic is exercised with gloo in
tests/test_distributed_cpu.py."""
import torch
import torch.nn.functional as F


def cluster_memory(mem, centroids, n_iters=10, all_reduce=None):
    """mem [N, D] fp32, centroids [K, D] initial -> (centroids, assignments int64 [N])."""
    centroids = centroids.clone()
    K, D = centroids.shape
    for it in range(n_iters + 1):
        dots = torch.mm(mem, centroids.t())
        _, assign = dots.max(dim=1)
        if it == n_iters:
            break
        counts = torch.zeros(K, dtype=torch.int32)
        sums = torch.zeros(K, D)
        for k in range(K):
            idx = torch.nonzero(assign == k)[:, 0]
            if len(idx) > 0:
                sums[k] = mem[idx].sum(dim=0)
                counts[k] = len(idx)
        if all_reduce is not None:
            counts, sums = all_reduce(counts), all_reduce(sums)
        mask = counts > 0
        centroids[mask] = sums[mask] / counts[mask].unsqueeze(1)
        centroids = F.normalize(centroids, dim=1, p=2)
    return centroids, assign


def pca_whiten_l2(x, pca=128):
    """faiss.PCAMatrix(d, pca, eigen_power=-0.5) + row L2 normalisation as `extras/decar-v2/clustering.py:19-42` uses them
    (faiss absent: restated from its documented behaviour - centre, covariance eigenvectors, eigenvalue^-0.5 scaling)."""
    import numpy as np
    x = np.asarray(x, np.float64)
    xc = x - x.mean(0, keepdims=True)
    lam, vec = np.linalg.eigh(xc.T @ xc / len(x))
    order = np.argsort(lam)[::-1][:pca]
    y = xc @ (vec[:, order] * np.power(np.maximum(lam[order], 1e-20), -0.5))
    return y / np.linalg.norm(y, axis=1, keepdims=True)


def lloyd(x, init, niter=20):
    """Euclidean k-means from given initial centroids: nearest centroid, plain means, empty clusters keep theirs."""
    import numpy as np
    x = np.asarray(x, np.float64)
    c = np.array(init, np.float64)
    for it in range(niter + 1):
        d2 = (x * x).sum(1)[:, None] - 2.0 * x @ c.T + (c * c).sum(1)[None]
        a = d2.argmin(1)
        if it == niter:
            break
        for k in range(len(c)):
            m = a == k
            if m.any():
                c[k] = x[m].mean(0)
    return a, float(np.maximum(d2[np.arange(len(x)), a], 0).sum()), c
