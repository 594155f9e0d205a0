"""Offline pseudo-labelling for DeepCluster / UnFuSeD-style label files on MI355X - `extras/decar-v2/clustering.py:9-115` and the
tail of `extras/decar-v2/store_clusters.py:64-160` of the reference: embeddings -> PCA-whitening to 128 dims -> L2 rows ->
k-means (20 Lloyd iterations) -> `images_lists` -> a `path,label` CSV.

The reference does all of it with faiss (`PCAMatrix(512, 128, eigen_power=-0.5)`, `Clustering` on a `GpuIndexFlatL2`); faiss is
a third-party dependency that is neither in the reference tree nor in this image, so this follows its documented algorithm
(parity unpinned): centre, eigendecompose the covariance, keep the `pca` leading directions scaled by eigenvalue^-0.5; Lloyd's
algorithm from `k` distinct random points (seed = np.random.randint(1234) as the reference sets it), nearest centroid in L2,
plain means, empty clusters keep their centroid (faiss re-splits large clusters instead).
GPU work: covariance, projection and the E step are MFMA GEMMs (fp32 path), argmin = `row_argmax` of x.c - |c|^2 / 2, M step =
`kmeans_accumulate` + `kmeans_update_mean`.  The 512 x 512 eigendecomposition runs once on the host (numpy, float64)."""
import time

import numpy as np
import torch

from src import _native as N
from src import engine as E


def rearrange_clusters(image_list):
    """cluster -> member ids lists -> pseudo-label per id, in id order (`clustering.py:9-17`)."""
    ids, labels = [], []
    for cluster, images in enumerate(image_list):
        ids.extend(images)
        labels.extend([cluster] * len(images))
    return np.asarray(labels)[np.argsort(ids)]


@torch.no_grad()
def preprocess_features(npdata, pca=128):
    """[N, ndim] features -> [N, pca] PCA-reduced, whitened, L2-normalised (float32, on the device)."""
    x = torch.as_tensor(npdata, dtype=torch.float32).cuda().contiguous()
    n, d = x.shape
    xc = (x - x.mean(0, keepdim=True)).contiguous()
    cov = torch.empty(d, d, dtype=torch.float32, device=x.device)
    E.gemm(N.F32, 1, 1, d, d, n, xc, d, xc, d, cov, d, alpha=1.0 / n, out_f32=1)                 # xc^T xc / n
    lam, vec = np.linalg.eigh(cov.double().cpu().numpy())
    order = np.argsort(lam)[::-1][:pca]
    proj = (vec[:, order] * np.power(np.maximum(lam[order], 1e-20), -0.5)).astype(np.float32)    # [d, pca], eigen_power -0.5
    pt = torch.from_numpy(np.ascontiguousarray(proj.T)).cuda()                                    # [pca, d]: Linear weight layout
    y = torch.empty(n, pca, dtype=torch.float32, device=x.device)
    E.gemm(N.F32, 0, 0, n, pca, d, xc, d, pt, d, y, pca, out_f32=1)
    out = torch.empty_like(y)
    inv = torch.empty(n, dtype=torch.float32, device=x.device)
    N.call("l2norm_fwd", N.F32, y, n, pca, out, out, inv)
    return out


@torch.no_grad()
def run_kmeans(x, nmb_clusters, verbose=False, niter=20, seed=None):
    """x [N, d] (device or numpy) -> (cluster id per row as a python list, final loss = sum of squared distances)."""
    x = torch.as_tensor(x, dtype=torch.float32).cuda().contiguous()
    n, d = x.shape
    seed = np.random.randint(1234) if seed is None else seed      # the reference changes faiss's seed at every call this way
    pick = np.random.RandomState(seed).permutation(n)[:nmb_clusters]
    assert len(pick) == nmb_clusters, "fewer points than clusters"
    cent = x[torch.from_numpy(pick).cuda()].clone()
    dot = torch.empty(n, nmb_clusters, dtype=torch.float32, device=x.device)
    assign = torch.empty(n, dtype=torch.int64, device=x.device)
    sums = torch.empty(nmb_clusters, d, dtype=torch.float32, device=x.device)
    counts = torch.empty(nmb_clusters, dtype=torch.int32, device=x.device)
    bias = torch.empty(nmb_clusters, dtype=torch.float32, device=x.device)
    losses = []
    for it in range(niter + 1):
        N.call("row_sqnorm", cent, nmb_clusters, d, -0.5, bias)
        E.gemm(N.F32, 0, 0, n, nmb_clusters, d, x, d, cent, d, dot, nmb_clusters, bias=bias, out_f32=1)   # x.c - |c|^2 / 2
        N.call("row_argmax", dot, n, nmb_clusters, assign)
        best = dot.gather(1, assign[:, None])[:, 0]
        losses.append(float(((x * x).sum(1) - 2.0 * best).clamp_min(0).sum()))          # sum_i |x_i - c_a(i)|^2
        if it == niter:
            break
        N.call("kmeans_accumulate", x, assign, n, nmb_clusters, d, sums, counts)
        N.call("kmeans_update_mean", sums, counts, nmb_clusters, d, cent)
    if verbose:
        print("k-means loss evolution: {0}".format(np.array(losses)))
    return [int(v) for v in assign.cpu().tolist()], losses[-1]


class Kmeans(object):
    def __init__(self, k):
        self.k = k

    def cluster(self, data, verbose=False, pca=128):
        end = time.time()
        xb = preprocess_features(data, pca=pca)
        I, loss = run_kmeans(xb, self.k, verbose)
        self.images_lists = [[] for _ in range(self.k)]
        for i in range(len(data)):
            self.images_lists[I[i]].append(i)
        if verbose:
            print("k-means time: {0:.0f} s".format(time.time() - end))
        return loss


def write_pseudolabel_csv(paths, kmeans, out_csv):
    """`store_clusters.py:142-159`: one `path,pseudolabel` line per file, in file order."""
    labels = rearrange_clusters(kmeans.images_lists)
    with open(out_csv, "w") as f:
        for p, l in zip(paths, labels):
            f.write(f"{p},{int(l)}\n")
    return labels
