"""Oracle: DeepCluster-v2 spherical k-means (test infrastructure).

CPU restatement of `extras/decar-v2/utils.py:276-346` (cluster_memory) for one head / one crop: E step
`mm(mem, centroids.t()).max(1)`, M step per-cluster sum (the reference goes through scipy csr_matrix on the CPU),
optional all-reduce hooks, `centroids[mask] = sums[mask] / counts[mask]` (empty clusters keep their centroid),
`F.normalize`.  The reference function itself calls `.cuda()` and cannot run here: parity of this file is by reading,
its collective logic is exercised with gloo in tests/test_distributed_cpu.py."""
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
