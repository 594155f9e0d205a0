"""CPU: the native (C) augmentation planner must continue numpy's and python's Mersenne Twisters bit-exactly, i.e.
produce the same tables and leave both generators in the same state as the draws made by numpy / `random` themselves
(which is what the reference's Python does)."""
import copy
import random

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from conftest import CFG_S


def _module(cfg, max_batch=64):
    from src.augmentations import AugmentationModule
    m = AugmentationModule(cfg, 1000, max_batch=max_batch)
    m.R = 1025 + max_batch
    return m


def _cfg(spec=False, mix=True, rrc=True):
    c = copy.deepcopy(CFG_S)
    aug = {}
    if mix:
        aug["MixupBYOLA"] = c["pretrain"]["augmentations"]["MixupBYOLA"]
    if rrc:
        aug["RandomResizeCrop"] = c["pretrain"]["augmentations"]["RandomResizeCrop"]
    if spec:
        aug["SpecAugment"] = dict(F=30, T=40, num_freq_masks=2, num_time_masks=2, replace_with_zero=False)
    c["pretrain"]["augmentations"] = aug
    return c


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), sizes=st.lists(st.integers(1, 40), min_size=1, max_size=4),
       spec=st.booleans(), mix=st.booleans(), rrc=st.booleans(), T=st.sampled_from([96, 101, 12]))
def test_native_planner_bit_exact(seed, sizes, spec, mix, rrc, T):
    cfg = _cfg(spec, mix, rrc)
    F = 64 if T > 12 else 8
    if spec and T == 12:
        spec = False
        cfg = _cfg(False, mix, rrc)
    outs = []
    for native in (True, False):
        np.random.seed(seed % (2**32))
        random.seed(seed)
        np.random.random(seed % 700)                   # move both generators off their seed position
        [random.random() for _ in range(seed % 650)]
        m = _module(cfg)
        rec = []
        for B in sizes:
            lens = (np.arange(B) * 3571 + seed) % 40000 + 100       # some shorter, some longer than the window
            ip, fp, canvas, masks = (m.plan if native else m.plan_py)(B, F, T, lens=lens, unit=16000)
            rec.append((ip.copy(), fp.copy(), canvas, masks + [tuple(int(v) for v in m.last_starts)]))
        outs.append((rec, np.random.random(), random.random(), m.n_entries, m.clips_seen))
    (ra, na, pa, ea, ca), (rb, nb, pb, eb, cb) = outs
    assert na == nb and pa == pb and ea == eb and ca == cb
    for (ipa, fpa, cva, mka), (ipb, fpb, cvb, mkb) in zip(ra, rb):
        assert np.array_equal(ipa, ipb)
        assert np.array_equal(fpa, fpb)
        assert cva == cvb and mka == mkb


def test_native_planner_long_stream_crosses_fifo_and_twister_refills():
    """> 2048 FIFO entries and > 624 words of both generators."""
    cfg = _cfg(True)
    outs = []
    for native in (True, False):
        np.random.seed(3)
        random.seed(3)
        m = _module(cfg, 256)
        acc = []
        for _ in range(6):
            ip, fp, _, masks = (m.plan if native else m.plan_py)(256, 64, 101)
            acc.append((ip.copy(), fp.copy(), masks))
        outs.append((acc, np.random.random(), random.random()))
    assert outs[0][1:] == outs[1][1:]
    for a, b in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
