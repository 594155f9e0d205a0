"""Two-view augmentation front end: host planner + HIP kernels.

Mirror of the reference's `src/augmentations/__init__.py:5-35` (AugmentationModule) and
`augmentations.py` (RunningNorm :215-286, MixupBYOLA :82-116, RandomResizeCrop :14-61), re-designed for
the GPU: the per-clip Python chain of the reference becomes, per *batch*,

  host   : draw every random parameter with the SAME generators in the SAME order as the reference
           (global `numpy.random` and python `random`; SURVEY a8') -> two small tables
  device : clip_moments -> runnorm_scan (sequential recurrence, one thread) -> aug_normalize (writes the
           normalised clips into a device ring that doubles as MixupBYOLA's 2048-entry FIFO) ->
           aug_views (mix + bicubic random-resize-crop, one block per (clip, view))
           [-> mask_fill for the optional SpecAugment band masks]

The reference's FIFO receives every clip twice (once per view).  Entry g of that virtual sequence is
clip g//2, so `bank[k]` resolves to a clip index and the device ring only keeps the last 1025 + B
distinct normalised clips.  Semantics are those of one sequential stream (DataLoader worker count 0).
"""
import os
import random

import numpy as np
import torch
from torch import nn

from src import _native as N

__all__ = ["AugmentationModule", "RunningNorm", "MixupBYOLA", "RandomResizeCrop", "SpecAugment", "log_mixup_exp",
           "PrecomputedNorm", "NormalizeBatch"]

_F32 = np.float32


def log_mixup_exp(xa, xb, alpha):
    """Reference formula (augmentations.py:8-12), torch ops; kept for API parity / small host-side use."""
    x = alpha * xa.exp() + (1.0 - alpha) * xb.exp()
    return torch.log(x + torch.finfo(x.dtype).eps)


class RunningNorm(nn.Module):
    """Configuration + device state of the running normaliser (statistics live on the GPU)."""

    def __init__(self, epoch_samples, max_update_epochs=10, axis=(1, 2)):
        super().__init__()
        self.max_update = int(epoch_samples * max_update_epochs)
        self.axis = list(axis)
        self.state_i = None      # int64 [2] = {n_seen, max_update}
        self.state_f = None      # float32 [2] = {mu, s2}

    def device_state(self, device):
        if self.state_i is None:
            self.state_i = torch.tensor([0, self.max_update], dtype=torch.int64, device=device)
            self.state_f = torch.zeros(2, dtype=torch.float32, device=device)
        return self.state_i, self.state_f

    def __len__(self):
        return 0 if self.state_i is None else int(self.state_i[0])

    @property
    def mean(self):
        return None if self.state_f is None else self.state_f[0]

    @property
    def std(self):
        if self.state_f is None:
            return None
        return torch.clamp(torch.sqrt(self.state_f[1]), torch.finfo().eps, torch.finfo().max)

    def __repr__(self):
        return self.__class__.__name__ + f"(max_update={self.max_update},axis={self.axis})"


class MixupBYOLA(nn.Module):
    def __init__(self, ratio=0.4, n_memory=2048, log_mixup_exp=True):
        super().__init__()
        self.ratio, self.n, self.log_mixup_exp = ratio, n_memory, log_mixup_exp

    def __repr__(self):
        return self.__class__.__name__ + f"(ratio={self.ratio},n={self.n},log_mixup_exp={self.log_mixup_exp})"


class RandomResizeCrop(nn.Module):
    def __init__(self, virtual_crop_scale=(1.0, 1.5), freq_scale=(0.6, 1.5), time_scale=(0.6, 1.5)):
        super().__init__()
        assert time_scale[1] >= 1.0 and freq_scale[1] >= 1.0
        self.virtual_crop_scale, self.freq_scale, self.time_scale = virtual_crop_scale, freq_scale, time_scale
        self.interpolation = "bicubic"

    @staticmethod
    def get_params(virtual_crop_size, in_size, time_scale, freq_scale):
        canvas_h, canvas_w = virtual_crop_size
        src_h, src_w = in_size
        h = int(np.clip(int(np.random.uniform(*freq_scale) * src_h), 1, canvas_h))
        w = int(np.clip(int(np.random.uniform(*time_scale) * src_w), 1, canvas_w))
        i = random.randint(0, canvas_h - h) if canvas_h > h else 0
        j = random.randint(0, canvas_w - w) if canvas_w > w else 0
        return i, j, h, w

    def __repr__(self):
        s = self.__class__.__name__ + f"(virtual_crop_size={self.virtual_crop_scale}"
        s += ", time_scale={0}".format(tuple(round(v, 4) for v in self.time_scale))
        return s + ", freq_scale={0})".format(tuple(round(v, 4) for v in self.freq_scale))


class Kmix(nn.Module):
    """K-mix (`src/augmentations/augmentations.py:119-189` of the reference): mixup whose partner comes from the memory-bank
    entries of the centroid cluster FARTHEST from the view's own cluster (first non-empty cluster in descending centroid
    distance, first 128 entries of it, one drawn with np.random.randint).  Same constructor; `forward(x[1, F, T])` is the
    reference's per-call form, `mix_batch(views[n, F, T])` the same n calls with ONE cluster launch, one device->host copy of
    the n cluster ids and one mixing launch.
    Device side: the bank is a ring of views (HBM), cluster ids come from `kmix_cluster`, the mixing from `kmix_apply`.
    Host side (numpy-stream exact): alpha = ratio * np.random.random(), then np.random.randint(len(l)) - or randint(len(bank))
    while the bank holds fewer than 128 entries - per call, the FIFO of cluster ids, the far-to-near centroid order
    (torch.topk(torch.cdist(c, c), k=K).indices, computed once with the same torch ops the reference runs per call)."""

    def __init__(self, ratio=0.4, n_memory=2048, log_mixup_exp=True, top_k=None, centroid_path=None, centroids=None):
        super().__init__()
        self.ratio, self.n, self.log_mixup_exp, self.top_k = ratio, n_memory, log_mixup_exp, top_k
        c = centroids if centroids is not None else torch.load(centroid_path, map_location="cpu", weights_only=True)
        self.centroids = torch.as_tensor(c, dtype=torch.float32).cpu()
        cn = self.centroids / self.centroids.norm(dim=-1, keepdim=True)
        self.far_order = torch.topk(torch.cdist(cn, cn, p=2), k=len(cn), dim=1).indices.numpy()
        self._cn = cn.contiguous()
        self.bank_ids = []            # cluster id of every entry of the reference's FIFO (oldest first)
        self.entries = 0              # entries appended so far; FIFO position j <-> ring slot (entries - len + j) % R
        self.ring = self.R = None
        self.last_choice = None       # FIFO position chosen by the last call (None: no mixing)
        self.draws = []               # (randint argument, value) of every index draw - what the parity test compares

    def __repr__(self):
        return self.__class__.__name__ + f"(ratio={self.ratio},n={self.n},log_mixup_exp={self.log_mixup_exp})"

    def _ensure(self, n_new, numel, dev):
        need = self.n + max(n_new, 2)
        if self.ring is None or self.ring.device != dev or self.ring.shape[1] != numel or need > self.R:
            old, old_R = self.ring, self.R
            self.R = max(need, self.R or 0)
            self.ring = torch.zeros(self.R, numel, dtype=torch.float32, device=dev)
            if old is not None and old.shape[1] == numel:
                for g in range(max(0, self.entries - len(self.bank_ids)), self.entries):
                    self.ring[g % self.R].copy_(old[g % old_R])
            self._cn_dev = self._cn.to(dev)

    def _select(self, cx):
        """FIFO position of the partner for a view of cluster cx (`get_index` + the small-bank branch of `forward`)."""
        nb = len(self.bank_ids)
        if nb >= 128:
            if nb < self.top_k:
                raise TypeError("list indices must be integers or slices, not NoneType")     # get_index returned None
            ids = np.asarray(self.bank_ids)
            for cid in self.far_order[cx]:
                l = np.nonzero(ids == cid)[0]
                if len(l):
                    break
            l = l[:128]
            k = int(np.random.randint(len(l)))
            self.draws.append((len(l), k))
            return int(l[k])
        k = int(np.random.randint(nb))
        self.draws.append((nb, k))
        return k

    def _plan_call(self, cx, slot_out, coef_out, i):
        alpha = self.ratio * np.random.random()
        nb = len(self.bank_ids)
        if nb:
            j = self._select(cx)
            slot_out[i] = (self.entries - nb + j) % self.R
            a1 = 1.0 - alpha
            coef_out[i] = (_F32(a1), _F32(1.0 - a1))
            self.last_choice = j
        else:
            slot_out[i] = -1
            self.last_choice = None
        self.bank_ids.append(int(cx))
        if len(self.bank_ids) > self.n:
            self.bank_ids.pop(0)
        self.entries += 1

    @torch.no_grad()
    def mix_batch(self, views):
        """views [n, F, T] (device, in call order) -> the n results of consecutive `forward` calls, [n, F, T] float32."""
        if not views.is_cuda:
            raise RuntimeError("Kmix runs on the GPU only (no CPU fallback)")
        views = views.contiguous().float()
        n, F, T = views.shape
        dev = views.device
        self._ensure(n, F * T, dev)
        ids_d = torch.empty(n, dtype=torch.int32, device=dev)
        N.call("kmix_cluster", views, self._cn_dev, n, F, T, self._cn.shape[0], ids_d)
        g0 = self.entries
        flat = views.view(n, F * T)
        first = g0 % self.R                                          # this batch's views enter the ring before anybody reads them
        k = min(n, self.R - first)
        self.ring[first:first + k].copy_(flat[:k])
        if k < n:
            self.ring[:n - k].copy_(flat[k:])
        ids = ids_d.cpu().numpy()                                    # the one device -> host round trip
        slot = np.full(n, -1, np.int32)
        coef = np.zeros((n, 2), _F32)
        for i in range(n):
            self._plan_call(ids[i], slot, coef, i)
        out = torch.empty_like(views)
        N.call("kmix_apply", views, self.ring, torch.from_numpy(slot).to(dev), torch.from_numpy(coef).to(dev), n, F * T,
               int(bool(self.log_mixup_exp)), out)
        return out

    def forward(self, x):
        """x [1, F, T] -> mixed [1, F, T] (float32), as the reference's per-call form."""
        src = x.device
        out = self.mix_batch((x if x.is_cuda else x.cuda()).reshape(1, x.shape[-2], x.shape[-1]))
        return out if src.type == "cuda" else out.to(src)


class SpecAugment(nn.Module):
    """Frequency / time band masks of `extras/delores-s/specaugment.py:68-122` (time_warp excluded: it calls the
    removed torch.solve).  Draw order per view: all frequency masks, then all time masks."""

    def __init__(self, F=30, T=40, num_freq_masks=2, num_time_masks=2, replace_with_zero=False):
        super().__init__()
        self.F, self.T, self.nf, self.nt, self.zero = F, T, num_freq_masks, num_time_masks, replace_with_zero

    def plan(self, n_mel, n_time):
        """-> list of (axis, start, end); axis 1 = frequency rows, 0 = time columns."""
        out = []
        for axis, width, size, count in ((1, self.F, n_mel, self.nf), (0, self.T, n_time, self.nt)):
            for _ in range(count):
                f = random.randrange(0, width)
                f0 = random.randrange(0, size - f)
                if f == 0:
                    break                                   # the reference returns out of this mask family
                end = random.randrange(f0, f0 + f)
                out.append((axis, f0, end))
        return out

    @property
    def max_masks(self):
        return self.nf + self.nt


class PrecomputedNorm(nn.Module):
    def __init__(self, stats, axis=(1, 2)):
        super().__init__()
        self.axis = list(axis)
        self.mean, self.std = stats

    def forward(self, X):
        return (X - self.mean) / self.std


class NormalizeBatch(nn.Module):
    def __init__(self, axis=(0, 2, 3)):
        super().__init__()
        self.axis = list(axis)

    def forward(self, X):
        m = X.mean(dim=self.axis, keepdims=True)
        s = torch.clamp(X.std(dim=self.axis, keepdims=True), torch.finfo().eps, torch.finfo().max)
        return (X - m) / s


class AugmentationModule:
    """The Augmentation Module (same constructor and per-sample call as the reference; batched on the GPU)."""

    def __init__(self, config, len_of_files, max_batch=1024):
        aug = config["pretrain"]["augmentations"]
        self.mix = self.rrc = self.spec = None
        if "MixupBYOLA" in aug:
            self.mix = MixupBYOLA(ratio=aug["MixupBYOLA"]["ratio"], log_mixup_exp=aug["MixupBYOLA"]["log_mixup_exp"])
        if "RandomResizeCrop" in aug:
            r = aug["RandomResizeCrop"]
            self.rrc = RandomResizeCrop(virtual_crop_scale=r["virtual_crop_scale"], freq_scale=r["freq_crop_scale"],
                                        time_scale=r["time_crop_scale"])
        if "SpecAugment" in aug:
            self.spec = SpecAugment(**aug["SpecAugment"])
        self.kmix = None
        if "Kmix" in aug:
            k = aug["Kmix"]
            self.kmix = Kmix(ratio=k["ratio"], log_mixup_exp=k["log_mixup_exp"], top_k=k["top_k"], centroid_path=k.get("centroid_path"),
                             centroids=k.get("centroids"))
        if "PatchDrop" in aug:
            raise NotImplementedError("PatchDrop is not part of the HIP two-view path (SURVEY 2.4); drop the key")
        self.train_transform = nn.Sequential(*[m for m in (self.mix, self.rrc, self.kmix, self.spec) if m is not None])
        self.pre_norm = None
        if config["pretrain"]["normalization"] == "mean_var":
            self.pre_norm = RunningNorm(epoch_samples=2 * len_of_files)
        self.n_memory = self.mix.n if self.mix is not None else 0
        self.max_batch = max_batch
        self.n_entries = 0        # entries appended to the reference's virtual FIFO so far
        self.clips_seen = 0
        self.bank = None          # [R][F*T] ring of normalised clips
        self.R = 0
        self.last_plan = None

    # ------------------------------------------------------------------ host planner (bit-exact RNG order)
    def plan(self, B, F, T, lens=None, unit=0):
        """Native planner (csrc/planner.hip): continues numpy's and python's MT19937 states in C.
        lens (optional int array [B]) + unit: also draw the random window start of every clip, interleaved per clip
        as the reference's dataset does; the starts are left in `self.last_starts`."""
        import ctypes
        ip = np.zeros((B, 2, 8), np.int32)
        fp = np.zeros((B, 2, 2), _F32)
        K = self.spec.max_masks if self.spec is not None else 1
        mk = np.full((B, 2, K, 4), -1, np.int32)
        ch = cw = 0
        if self.rrc is not None:
            ch, cw = int(F * self.rrc.virtual_crop_scale[0]), int(T * self.rrc.virtual_crop_scale[1])
        lens_a = None if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
        starts = np.zeros(B, np.int32)
        st = np.random.get_state()
        np_key = np.ascontiguousarray(st[1], dtype=np.uint32)
        np_pos = ctypes.c_int(int(st[2]))
        ps = random.getstate()
        py_key = np.array(ps[1][:624], dtype=np.uint32)
        py_pos = ctypes.c_int(int(ps[1][624]))
        ne = ctypes.c_longlong(self.n_entries)
        fs = self.rrc.freq_scale if self.rrc is not None else (1.0, 1.0)
        ts = self.rrc.time_scale if self.rrc is not None else (1.0, 1.0)
        sp = self.spec
        N.call_host("aug_plan_host", np_key.ctypes.data, ctypes.addressof(np_pos), py_key.ctypes.data,
                    ctypes.addressof(py_pos), B, F, T, self.clips_seen, ctypes.addressof(ne), self.R, self.n_memory,
                    int(self.mix is not None), float(self.mix.ratio) if self.mix is not None else 0.0,
                    int(self.rrc is not None), float(fs[0]), float(fs[1]), float(ts[0]), float(ts[1]), ch, cw,
                    int(sp is not None), sp.F if sp else 1, sp.T if sp else 1, sp.nf if sp else 0, sp.nt if sp else 0,
                    lens_a.ctypes.data if lens_a is not None else None, int(unit), starts.ctypes.data, ip.ctypes.data,
                    fp.ctypes.data, mk.ctypes.data)
        self.last_starts = starts if lens_a is not None else None
        np.random.set_state((st[0], np_key, np_pos.value, st[3], st[4]))
        random.setstate((ps[0], tuple(int(v) for v in py_key) + (py_pos.value,), ps[2]))
        self.n_entries = ne.value
        self.clips_seen += B
        masks = []
        if sp is not None:
            for b in range(B):
                for v in range(2):
                    masks.append([tuple(int(x) for x in m[:3]) for m in mk[b, v] if m[0] >= 0])
        return ip, fp, (ch, cw), masks

    def plan_py(self, B, F, T, lens=None, unit=0):
        """Pure-Python planner: the same draws made with numpy / `random` themselves (cross-check of `plan`)."""
        starts = np.zeros(B, np.int32)
        ip = np.zeros((B, 2, 8), np.int32)
        fp = np.zeros((B, 2, 2), _F32)
        masks = []
        ch = cw = 0
        if self.rrc is not None:
            ch, cw = int(F * self.rrc.virtual_crop_scale[0]), int(T * self.rrc.virtual_crop_scale[1])
        for b in range(B):
            c = self.clips_seen + b
            if lens is not None:
                over = int(lens[b]) - unit
                starts[b] = random.randint(0, over) if over > 0 else 0
            for v in range(2):
                e = ip[b, v]
                e[0] = c % self.R
                e[1] = -1
                if self.mix is not None:
                    alpha = self.mix.ratio * np.random.random()
                    n_bank = min(self.n_entries, self.n_memory)
                    if n_bank > 0:
                        k = np.random.randint(n_bank)
                        g = self.n_entries - n_bank + k
                        e[1] = (g // 2) % self.R
                        a1 = 1.0 - alpha
                        fp[b, v, 0] = _F32(a1)
                        fp[b, v, 1] = _F32(1.0 - a1)
                    self.n_entries += 1
                if self.rrc is not None:
                    i, j, h, w = RandomResizeCrop.get_params((ch, cw), (F, T), self.rrc.time_scale, self.rrc.freq_scale)
                    e[2:7] = (i, j, h, w, 1)
                if self.spec is not None:
                    masks.append(self.spec.plan(F, T))
        self.clips_seen += B
        self.last_starts = starts if lens is not None else None
        return ip, fp, (ch, cw), masks

    def _ensure_ring(self, B, n=None, device=None):
        """Size (or grow) the ring of normalised clips; slots are clip_index % R, so R is fixed before planning."""
        need = 1025 + max(B, self.max_batch)
        if self.R == 0:
            self.R = need
        elif need > self.R:
            if self.bank is not None:
                old, old_R = self.bank, self.R
                self.bank = torch.zeros(need, old.shape[1], dtype=torch.float32, device=old.device)
                for c in range(max(0, self.clips_seen - 1025), self.clips_seen):
                    self.bank[c % need].copy_(old[c % old_R])
            self.R = need
        self.max_batch = max(self.max_batch, B)
        if self.bank is None and n is not None:
            self.bank = torch.zeros(self.R, n, dtype=torch.float32, device=device)

    def _upload(self, key, arr, dev, depth=int(os.environ.get("AUDIOSSL_UPLOAD_DEPTH", "2"))):
        """Host table -> device through a small ring of PINNED staging buffers.  A pageable source would make the copy
        synchronous, i.e. one host/device rendezvous per step that keeps the launch side from running ahead.
        The ring is also what bounds HOW FAR the launch side runs ahead (a slot is reused once its copy of `depth` batches ago has
        completed).  Two batches is enough to keep the device fed (the host needs ~0.8 ms per 2 ms step) and it matters early in a
        process: with four batches in flight the first ~20 graph replays after start-up contained two 1-2 ms stalls of the DEVICE -
        the host blocked for 6-12 ms inside an asynchronous copy while the runtime grew its pools, the device ran dry
        (tools/step_times.py: 2.14 ms per step over bench.py's 20-step window with depth 4, 2.06 with 3, 2.02 with 2; later windows of
        the same process read 1.98 either way)."""
        ring = self.__dict__.setdefault("_staging", {}).setdefault((key, arr.shape, arr.dtype.str), {"i": 0, "slots": []})
        if len(ring["slots"]) < depth:
            ring["slots"].append([torch.from_numpy(np.empty_like(arr)).pin_memory(), None])
        slot = ring["slots"][ring["i"] % len(ring["slots"])]
        ring["i"] += 1
        if slot[1] is not None:
            slot[1].synchronize()                       # the copy that last used this slot (depth steps ago)
        slot[0].numpy()[...] = arr
        out = slot[0].to(dev, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        return out

    def _ensure_bank(self, B, n, device):
        self._ensure_ring(B, n, device)

    # ------------------------------------------------------------------ batched device path
    @torch.no_grad()
    def augment_batch(self, lms, plan=None):
        """lms [B, F, T] or [B, 1, F, T] log-mel (device) -> (view1, view2), each [B, 1, F, T] float32.
        plan: result of `self.plan(...)` made earlier for exactly this batch (e.g. at collate time, together with the
        window crops); by default the draws happen here."""
        if lms.dim() == 4:
            lms = lms[:, 0]
        lms = lms.contiguous().float()
        if not lms.is_cuda:
            raise RuntimeError("AugmentationModule runs on the GPU only (no CPU fallback)")
        B, F, T = lms.shape
        n = F * T
        dev = lms.device
        self._ensure_bank(B, n, dev)
        mu = torch.zeros(B, dtype=torch.float32, device=dev)
        sd = torch.ones(B, dtype=torch.float32, device=dev)
        if self.pre_norm is not None:
            mom = torch.empty(B, 2, dtype=torch.float64, device=dev)
            si, sf = self.pre_norm.device_state(dev)
            N.call("clip_moments", lms, mom, B, n)
            N.call("runnorm_scan", mom, B, n, si, sf, mu, sd)
        if plan is None:
            slot0 = self.clips_seen % self.R
            plan = self.plan(B, F, T)
        else:
            slot0 = int(plan[0][0, 0, 0])
        N.call("aug_normalize", lms, mu, sd, self.bank, slot0, self.R, B, n)
        ip, fp, (ch, cw), masks = plan
        self.last_plan = (ip, fp, masks)
        ip_d = self._upload("ip", ip, dev)
        fp_d = self._upload("fp", fp, dev)
        v1 = torch.empty(B, 1, F, T, dtype=torch.float32, device=dev)
        v2 = torch.empty(B, 1, F, T, dtype=torch.float32, device=dev)
        log_mix = 1 if (self.mix is None or self.mix.log_mixup_exp) else 0
        N.call("aug_views", self.bank, self.R, ip_d, fp_d, v1, v2, B, F, T, max(ch, F), max(cw, T), log_mix)
        if self.kmix is not None:
            # Kmix closes the reference's Sequential (MixupBYOLA, RandomResizeCrop, Kmix): its calls come in the order clip 0
            # view 1, clip 0 view 2, clip 1 view 1, ...  Batched: the 2B views are clustered in one launch and mixed in one;
            # the numpy draws of these calls are made AFTER the batch's MixupBYOLA / RandomResizeCrop draws (the reference
            # interleaves them view by view; its len(l)-dependent randint makes the interleaved stream depend on device results,
            # i.e. one host <-> device round trip per view - `Kmix.forward` used view by view keeps that order exactly)
            both = torch.stack([v1[:, 0], v2[:, 0]], dim=1).reshape(2 * B, F, T)
            mixed = self.kmix.mix_batch(both).view(B, 2, F, T)
            v1, v2 = mixed[:, 0:1].contiguous(), mixed[:, 1:2].contiguous()
        if self.spec is not None:
            K = self.spec.max_masks
            tab = np.full((B, 2, K, 4), -1, np.int32)
            for idx, ms in enumerate(masks):
                for k, (axis, st, en) in enumerate(ms):
                    tab[idx // 2, idx % 2, k] = (axis, st, en, 0)
            for v, view in enumerate((v1, v2)):
                t = self._upload(f"mask{v}", np.ascontiguousarray(tab[:, v]), dev)
                N.call("mask_fill", view, t, B, K, F, T, int(self.spec.zero))
        return v1, v2

    def __call__(self, x):
        """Per-sample form of the reference: x [1, F, T] -> (view1 [1, F, T], view2 [1, F, T])."""
        src = x.device
        v1, v2 = self.augment_batch(x.cuda() if not x.is_cuda else x)
        v1, v2 = v1[0], v2[0]
        return (v1, v2) if src.type == "cuda" else (v1.to(src), v2.to(src))
