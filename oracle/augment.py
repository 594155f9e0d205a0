"""Oracle: running normalisation + two augmented views (test infrastructure).

Sequential, per-clip restatement of (reference file:line)
  * `src/augmentations/augmentations.py:215-286`  RunningMean/Variance/Norm
  * `src/augmentations/augmentations.py:8-12, 82-116`  log_mixup_exp, MixupBYOLA
  * `src/augmentations/augmentations.py:14-61`    RandomResizeCrop
  * `src/augmentations/__init__.py:5-35`          AugmentationModule
  * `extras/delores-s/specaugment.py:68-122`      freq_mask / time_mask
Pinned by tests/golden/aug_*.npz (outputs of the reference's own classes).

RNG: the global `numpy.random` (legacy MT19937) and python `random` streams
are consumed in exactly the reference's order (SURVEY a8'):
  per view: np.random.random() -> [np.random.randint(len(bank))] ->
            np.random.uniform (h) -> np.random.uniform (w) ->
            [random.randint i] -> [random.randint j]
"""
import random

import numpy as np
import torch

F32_EPS = float(np.finfo(np.float32).eps)
F32_MAX = float(np.finfo(np.float32).max)


# ------------------------------------------------------------------ RunningNorm
class RunningNorm:
    """augmentations.py:215-282 with axis=[1,2] on a [1,F,T] input.

    Quirk kept on purpose: the mean recurrence divides by the sample count
    *before* the increment (`mu += (m - mu) / n`), so the second sample
    overwrites the first."""

    def __init__(self, epoch_samples, max_update_epochs=10):
        self.max_update = epoch_samples * max_update_epochs
        self.n = 0
        self.mu = None          # float32 scalars held as 0-d tensors
        self.s2 = None
        self.mean = None
        self.std = None

    def stats_update(self, x):
        m = x.mean()
        if self.n == 0:
            self.mu = m.clone()
        else:
            self.mu = self.mu + (m - self.mu) / self.n
        v = ((x - self.mu) ** 2).mean()
        if self.n == 0:
            self.s2 = v.clone()
        else:
            self.s2 = self.s2 + (v - self.s2) / self.n
        self.n += 1
        self.mean = self.mu
        self.std = torch.clamp(torch.sqrt(self.s2), F32_EPS, F32_MAX)

    def __call__(self, x):
        if self.n < self.max_update:
            self.stats_update(x)
        return (x - self.mean) / self.std


# -------------------------------------------------------------------- Mixup
def log_mixup_exp(xa, xb, alpha):
    x = alpha * xa.exp() + (1.0 - alpha) * xb.exp()
    return torch.log(x + F32_EPS)


class MixupBYOLA:
    def __init__(self, ratio=0.4, n_memory=2048, log_mixup_exp=True):
        self.ratio, self.n, self.lme = ratio, n_memory, log_mixup_exp
        self.memory_bank = []
        self.trace = []          # (alpha, bank_index or -1) per call

    def __call__(self, x):
        alpha = self.ratio * np.random.random()
        if self.memory_bank:
            k = np.random.randint(len(self.memory_bank))
            z = self.memory_bank[k]
            mixed = log_mixup_exp(x, z, 1.0 - alpha) if self.lme \
                else alpha * z + (1.0 - alpha) * x
        else:
            k = -1
            mixed = x
        self.trace.append((alpha, k))
        self.memory_bank = (self.memory_bank + [x])[-self.n:]
        return mixed.to(torch.float)


# ------------------------------------------------------------ RandomResizeCrop
CUBIC_A = -0.75   # PyTorch's bicubic coefficient


class Kmix:
    """`src/augmentations/augmentations.py:119-189`: mixup with a partner from the farthest non-empty centroid cluster.
    Per call: alpha = ratio * np.random.random(); with a non-empty bank the partner index is np.random.randint(len(bank))
    below 128 entries, else `get_index`: clusters = nearest (unit-row) centroid of the time-averaged spectrum of every bank
    entry / of x, centroids visited from the farthest to the nearest to x's own, first cluster with members wins, its first
    128 members (bank order) are the candidates, one drawn with np.random.randint; then the log-domain mix and the FIFO
    append of x (the INPUT, not the mix)."""

    def __init__(self, ratio=0.4, n_memory=2048, log_mixup_exp=True, top_k=None, centroids=None):
        self.ratio, self.n, self.lme, self.top_k = ratio, n_memory, log_mixup_exp, top_k
        self.centroids = torch.as_tensor(centroids, dtype=torch.float32)
        self.memory_bank = []
        self.draws = []

    def get_index(self, x):
        if len(self.memory_bank) < self.top_k:
            return None
        c = self.centroids / self.centroids.norm(dim=-1, keepdim=True)
        avg = torch.stack([z.squeeze(0).T.mean(dim=0) for z in self.memory_bank])
        avg = avg / avg.norm(dim=-1, keepdim=True)
        far_first = torch.topk(torch.cdist(c, c, p=2), k=len(c), dim=1).indices
        bank_cluster = torch.argmin(torch.cdist(avg, c, p=2), dim=1)
        own = int(torch.argmin(torch.cdist(x.squeeze(0).T.mean(dim=0).unsqueeze(0), c, p=2), dim=1))
        members = []
        for cid in far_first[own].tolist():
            members = (bank_cluster == cid).nonzero().flatten().tolist()
            if members:
                break
        members = members[:128]
        k = np.random.randint(len(members))
        self.draws.append((len(members), int(k)))
        return members[k]

    def __call__(self, x):
        alpha = self.ratio * np.random.random()
        if self.memory_bank:
            if len(self.memory_bank) >= 128:
                j = self.get_index(x)
            else:
                j = np.random.randint(len(self.memory_bank))
                self.draws.append((len(self.memory_bank), int(j)))
            z = self.memory_bank[j]
            mixed = log_mixup_exp(x, z, 1.0 - alpha) if self.lme else alpha * z + (1.0 - alpha) * x
        else:
            mixed = x
        self.memory_bank = (self.memory_bank + [x])[-self.n:]
        return mixed.to(torch.float)


def cubic_coeffs(t):
    """PyTorch get_cubic_upsample_coefficients (A=-0.75) for fraction t."""
    A = CUBIC_A
    x1 = t
    c0 = ((A * (x1 + 1) - 5 * A) * (x1 + 1) + 8 * A) * (x1 + 1) - 4 * A
    c1 = ((A + 2) * x1 - (A + 3)) * x1 * x1 + 1
    x2 = 1 - t
    c2 = ((A + 2) * x2 - (A + 3)) * x2 * x2 + 1
    c3 = ((A * (x2 + 1) - 5 * A) * (x2 + 1) + 8 * A) * (x2 + 1) - 4 * A
    return c0, c1, c2, c3


def bicubic_resize_align_corners(src, out_h, out_w):
    """F.interpolate(src[None,None], (out_h,out_w), 'bicubic', align_corners=True).

    src: [h, w] float32.  Separable 4x4 taps, border indices clamped."""
    h, w = src.shape
    sy = torch.tensor((h - 1) / (out_h - 1) if out_h > 1 else 0.0, dtype=torch.float32)
    sx = torch.tensor((w - 1) / (out_w - 1) if out_w > 1 else 0.0, dtype=torch.float32)
    ry = sy * torch.arange(out_h, dtype=torch.float32)
    rx = sx * torch.arange(out_w, dtype=torch.float32)
    iy = torch.floor(ry)
    ix = torch.floor(rx)
    ty = ry - iy
    tx = rx - ix
    iy = iy.long()
    ix = ix.long()
    wy = cubic_coeffs(ty)
    wx = cubic_coeffs(tx)
    out = torch.zeros(out_h, out_w, dtype=torch.float32)
    for a in range(4):
        yy = (iy - 1 + a).clamp(0, h - 1)
        row = torch.zeros(out_h, out_w, dtype=torch.float32)
        for b in range(4):
            xx = (ix - 1 + b).clamp(0, w - 1)
            row = row + src[yy][:, xx] * wx[b][None, :]
        out = out + row * wy[a][:, None]
    return out


class RandomResizeCrop:
    def __init__(self, virtual_crop_scale=(1.0, 1.5), freq_scale=(0.6, 1.5), time_scale=(0.6, 1.5)):
        assert time_scale[1] >= 1.0 and freq_scale[1] >= 1.0
        self.vcs, self.fs, self.ts = virtual_crop_scale, freq_scale, time_scale
        self.trace = []          # (i, j, h, w)

    @staticmethod
    def get_params(canvas, in_size, time_scale, freq_scale):
        ch, cw = canvas
        sh, sw = in_size
        h = int(np.clip(int(np.random.uniform(*freq_scale) * sh), 1, ch))
        w = int(np.clip(int(np.random.uniform(*time_scale) * sw), 1, cw))
        i = random.randint(0, ch - h) if ch > h else 0
        j = random.randint(0, cw - w) if cw > w else 0
        return i, j, h, w

    def __call__(self, lms):
        _, H, W = lms.shape
        ch, cw = int(H * self.vcs[0]), int(W * self.vcs[1])
        canvas = torch.zeros(ch, cw, dtype=torch.float32)
        x0, y0 = (cw - W) // 2, (ch - H) // 2
        canvas[y0:y0 + H, x0:x0 + W] = lms[0]
        i, j, h, w = self.get_params((ch, cw), (H, W), self.ts, self.fs)
        self.trace.append((i, j, h, w))
        crop = canvas[i:i + h, j:j + w]
        return bicubic_resize_align_corners(crop, H, W)[None]


# ---------------------------------------------------------- AugmentationModule
class AugmentationModule:
    """`src/augmentations/__init__.py:5-35` for the MixupBYOLA+RandomResizeCrop
    chain (Kmix / PatchDrop keys are dropped)."""

    def __init__(self, config, len_of_files):
        aug = config["pretrain"]["augmentations"]
        self.mix = None
        self.rrc = None
        if "MixupBYOLA" in aug:
            self.mix = MixupBYOLA(ratio=aug["MixupBYOLA"]["ratio"],
                                  log_mixup_exp=aug["MixupBYOLA"]["log_mixup_exp"])
        if "RandomResizeCrop" in aug:
            r = aug["RandomResizeCrop"]
            self.rrc = RandomResizeCrop(r["virtual_crop_scale"], r["freq_crop_scale"], r["time_crop_scale"])
        self.pre_norm = None
        if config["pretrain"]["normalization"] == "mean_var":
            self.pre_norm = RunningNorm(epoch_samples=2 * len_of_files)

    def _chain(self, x):
        if self.mix is not None:
            x = self.mix(x)
        if self.rrc is not None:
            x = self.rrc(x)
        return x

    def __call__(self, x):
        if self.pre_norm:
            x = self.pre_norm(x)
        return self._chain(x), self._chain(x)


# ------------------------------------------------------------------ SpecAugment
def _band_mask(spec, width_max, num_masks, replace_with_zero, axis, trace=None):
    """Shared body of freq_mask (axis=1) / time_mask (axis=0) on a (T, dim) input,
    `extras/delores-s/specaugment.py:68-122`: early return when the drawn width
    is 0; the fill value is the *current* tensor mean (already-masked values
    included)."""
    out = spec.clone()
    n = out.shape[axis]
    for _ in range(num_masks):
        f = random.randrange(0, width_max)
        f0 = random.randrange(0, n - f)
        if f == 0:
            if trace is not None:
                trace.append((f, f0, -1))
            return out
        end = random.randrange(f0, f0 + f)
        if trace is not None:
            trace.append((f, f0, end))
        fill = 0.0 if replace_with_zero else out.mean()
        if axis == 1:
            out[:, f0:end] = fill
        else:
            out[f0:end, :] = fill
    return out


def freq_mask(spec, F=30, num_masks=1, replace_with_zero=False, trace=None):
    return _band_mask(spec, F, num_masks, replace_with_zero, 1, trace)


def time_mask(spec, T=40, num_masks=1, replace_with_zero=False, trace=None):
    return _band_mask(spec, T, num_masks, replace_with_zero, 0, trace)
