"""GPU: every HIP kernel through the C ABI against the CPU oracle (torch fp32 / numpy) on seeded inputs.

Tolerances: fp32 path (exact-f32 MFMA) 1e-4 relative; bf16 path 2e-2 rel-L2 (bf16 has 8 significand bits);
index / mask outputs bit-exact."""
import random

import numpy as np
import pytest
import torch

from oracle import augment as OA
from oracle import fill
from oracle import frontend as FE
from helpers import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from src import _native
    _native.lib()
    assert torch.cuda.is_available()
    return _native


def dev(x):
    return torch.as_tensor(x).cuda().contiguous()


TOL = {0: 1e-4, 1: 2e-2}


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("mode", ["NT", "NN", "TN"])
@pytest.mark.parametrize("shape", [(8, 2048, 512), (200, 136, 72), (512, 128, 2048), (1000, 264, 40), (64, 576, 1600)])
def test_gemm_modes(N, dtype, mode, shape):
    M, Nn, K = shape
    ta, tb = {"NT": (0, 0), "NN": (0, 1), "TN": (1, 1)}[mode]
    if (ta and M % 8) or (tb and Nn % 8) or ((not ta or not tb) and K % 8):
        pytest.skip("vector dimension must be a multiple of 8")
    td = N.torch_dtype(dtype)
    A = torch.from_numpy(fill.uniform((K, M) if ta else (M, K), 11 + M))
    B = torch.from_numpy(fill.uniform((K, Nn) if tb else (Nn, K), 12 + Nn))
    Ad, Bd = dev(A).to(td), dev(B).to(td)
    Aq, Bq = Ad.float().cpu(), Bd.float().cpu()               # operands as the kernel sees them
    ref = (Aq.T if ta else Aq).double() @ (Bq if tb else Bq.T).double()
    C = torch.full((M, Nn), float("nan"), device="cuda", dtype=torch.float32)
    N.call("gemm", dtype, ta, tb, M, Nn, K, 1.0, Ad, Ad.shape[1], Bd, Bd.shape[1], C, Nn, None, 0, None, 0, 1.0,
           None, 0, 1, 0, 1, None, 0)
    torch.cuda.synchronize()
    assert torch.isfinite(C).all()
    assert rel_l2(C.cpu(), ref) < 1e-5                         # fp32 accumulate of exactly-represented operands
    # split-K + atomic accumulation on top of a non-zero C
    C2 = torch.ones(M, Nn, device="cuda", dtype=torch.float32)
    N.call("gemm", dtype, ta, tb, M, Nn, K, 0.5, Ad, Ad.shape[1], Bd, Bd.shape[1], C2, Nn, None, 0, None, 0, 1.0,
           None, 0, 1, 1, 3, None, 0)
    torch.cuda.synchronize()
    assert rel_l2(C2.cpu(), 1.0 + 0.5 * ref) < 1e-5


@pytest.mark.parametrize("dtype", [0, 1])
def test_gemm_epilogue(N, dtype):
    M, Nn, K = 72, 200, 64
    td = N.torch_dtype(dtype)
    A = dev(fill.uniform((M, K), 21)).to(td)
    W = dev(fill.uniform((Nn, K), 22)).to(td)
    bias = dev(fill.uniform((Nn,), 23))
    keep = dev((fill.uniform01((M, Nn), 24) >= 0.3).astype(np.uint8))
    gate = dev(fill.uniform((M, Nn), 25)).to(td)
    out = torch.empty(M, Nn, device="cuda", dtype=td)
    N.call("gemm", dtype, 0, 0, M, Nn, K, 1.0, A, K, W, K, out, Nn, bias, 1, keep, Nn, 1.0 / 0.7, gate, Nn, 0, 0, 1, None, 0)
    torch.cuda.synchronize()
    ref = torch.relu(A.float() @ W.float().T + bias) * keep.float() / 0.7 * (gate.float() > 0)
    assert rel_l2(out.float().cpu(), ref.cpu()) < (1e-5 if dtype == 0 else 4e-3)


def test_linear_fwd_split_k_adds_the_bias_once(N):
    """`engine.linear_fwd` on a tiny output with a long contraction (the 2048 -> 128 embedding layers: M = 512) splits K over
    eight workgroups per tile with fp32 atomics; the bias comes from the first split only.  Also through the C ABI: relu with
    ksplit > 1 is rejected."""
    from src import engine as E
    g = torch.Generator().manual_seed(29)
    M, Nn, K = 512, 128, 2048
    X = (torch.randn(M, K, generator=g) * 0.5).cuda().bfloat16()
    W = (torch.randn(Nn, K, generator=g) * 0.05).cuda().bfloat16()
    bias = torch.randn(Nn, generator=g).cuda()
    N.PROFILE, N.PROFILE_ALL = {}, True
    try:
        out = E.linear_fwd(N.BF16, X, W, M, Nn, K, bias=bias, out_f32=1)
        torch.cuda.synchronize()
        args = N.PROFILE["audiossl_gemm"][0][2]
    finally:
        N.PROFILE, N.PROFILE_ALL = None, False
    assert args[-2] == 8 and args[-3] == 1                       # (..., atomic, ksplit, ldr): ksplit 8, atomic accumulation
    ref = X.double() @ W.double().T + bias.double()
    assert out.dtype == torch.float32 and rel_l2(out.double().cpu(), ref.cpu()) < 1e-5
    lib = N.lib()
    assert lib.audiossl_gemm(1, 0, 0, M, Nn, K, 1.0, X.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), Nn, bias.data_ptr(), 1, None, 0, 1.0,
                             None, 0, 1, 1, 8, None, 0, None) == -1


@pytest.mark.parametrize("shape", [(72, 200, 64, 200), (256, 128, 192, 128), (64, 104, 128, 105)])
def test_gemm_exclusive_accumulate(N, shape):
    """atomic = 2: C += A^T B by plain load-add-store (the weight-gradient GEMMs without split-K); ldc = 105 breaks the
    16-byte row alignment and so exercises the one-column-per-lane epilogue."""
    M, Nn, K, ldc = shape
    A = dev(fill.uniform((K, M), 31)).bfloat16()
    B = dev(fill.uniform((K, Nn), 32)).bfloat16()
    C0 = dev(fill.uniform((M, ldc), 33))
    C = C0.clone()
    N.call("gemm", 1, 1, 1, M, Nn, K, 0.5, A, M, B, Nn, C, ldc, None, 0, None, 0, 1.0, None, 0, 1, 2, 1, None, 0)
    torch.cuda.synchronize()
    ref = C0.clone()
    ref[:, :Nn] += 0.5 * (A.float().T @ B.float())
    assert rel_l2(C.cpu(), ref.cpu()) < 1e-5
    with pytest.raises(RuntimeError, match="EINVAL"):                              # split-K needs the atomic form
        N.call("gemm", 1, 1, 1, M, Nn, K, 0.5, A, M, B, Nn, C, ldc, None, 0, None, 0, 1.0, None, 0, 1, 2, 2, None, 0)


def test_gemm_rejects_bad_arguments(N):
    a = torch.zeros(16, 16, device="cuda")
    with pytest.raises(RuntimeError, match="EINVAL"):
        N.call("gemm", 0, 0, 0, 16, 16, 12, 1.0, a, 16, a, 16, a, 16, None, 0, None, 0, 1.0, None, 0, 1, 0, 1, None, 0)   # K % 8
    with pytest.raises(RuntimeError, match="EALIGN"):
        N.call("gemm", 0, 0, 0, 16, 16, 8, 1.0, a, 12, a, 16, a, 16, None, 0, None, 0, 1.0, None, 0, 1, 0, 1, None, 0)    # lda % 8
    with pytest.raises(RuntimeError, match="device tensors"):
        N.call("gemm", 0, 0, 0, 16, 16, 16, 1.0, a.cpu(), 16, a, 16, a, 16, None, 0, None, 0, 1.0, None, 0, 1, 0, 1, None, 0)


# ------------------------------------------------------------------------------------------------ log-mel
def _waves(B, L, salt):
    w = fill.uniform((B, L), salt, -0.1, 0.1)
    t = np.arange(L) / 16000.0
    w += (0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t)).astype(np.float32)
    return w


@pytest.mark.parametrize("L", [16000, 15200])
def test_logmel_matches_oracle(N, L):
    from src.utils import MelSpectrogramLibrosa, extract_log_mel_spectrogram
    w = _waves(6, L, 31)
    w[4] = 0.0                 # silence
    w[5] = 1.0                 # full-scale DC
    mel = MelSpectrogramLibrosa()
    got = extract_log_mel_spectrogram(torch.from_numpy(w).cuda(), mel)
    ref = FE.log_mel_batch(torch.from_numpy(w))
    assert got.shape == ref.shape == (6, 64, 1 + L // 160)
    e_got, e_ref = torch.exp(got.cpu().double()), torch.exp(ref.double())
    # SURVEY 8d tolerance: |d exp(logmel)| <= 1e-5*max + 1e-7, per clip
    for b in range(6):
        tol = 1e-5 * float(e_ref[b].max()) + 1e-7
        assert float((e_got[b] - e_ref[b]).abs().max()) <= tol, b
    # silence floor is exact: log(eps)
    assert torch.allclose(got[4].cpu(), ref[4], atol=1e-5)
    # per-clip API keeps the reference's contract (mel power, [n_mels, T])
    p = mel(torch.from_numpy(w[0]))
    assert p.shape == (64, 1 + L // 160) and not p.is_cuda
    assert rel_l2(p, FE.MelSpectrogram()(w[0])) < 1e-5


@pytest.mark.parametrize("n_mels,L,B", [(128, 16000, 5), (64, 16000, 600), (64, 2240, 3), (40, 15200, 2)])
def test_logmel_persistent_kernel_shapes(N, n_mels, L, B):
    """The persistent log-mel kernel on the shapes the main test does not reach: 128 mel rows (two per lane: the SS-MAST
    front end), more frames than waves can take one each (B = 600: 60,600 frames over 2,048 waves), clips so short that every
    frame touches the reflected border, a row count that is not a multiple of the wave.  Against the oracle."""
    from src.utils import MelSpectrogramLibrosa
    mel = MelSpectrogramLibrosa(n_mels=n_mels)
    ref_mel = FE.MelSpectrogram(n_mels=n_mels)
    w = fill.uniform((B, L), 95 + n_mels, -0.4, 0.4)
    w[:, ::7] *= 0.1
    got = mel.logmel(torch.from_numpy(w).cuda()).cpu()
    pick = list(range(B)) if B <= 8 else [0, 1, B // 2, B - 2, B - 1]
    ref = FE.log_mel_batch(torch.from_numpy(w[pick]), ref_mel)
    assert got.shape == (B, n_mels, 1 + L // 160)
    for i, b in enumerate(pick):
        e_got, e_ref = got[b].exp(), ref[i].exp()
        assert float((e_got - e_ref).abs().max()) <= 1e-5 * float(e_ref.max()) + 1e-7, b


def test_l2_waveform_normalisation_front_end(cfg_s):
    """`normalization: l2` (src/dataset/upstream_dataset.py:61-62: F.normalize(waveform, dim=-1, p=2) before the log-mel, no
    RunningNorm): the batched HIP front end against torch's F.normalize + the oracle log-mel, incl. an all-zero clip (the
    1e-12 clamp) and a full-scale clip."""
    import copy
    from src.augmentations import AugmentationModule
    from src.dataset import UpstreamFrontEnd
    cfg = copy.deepcopy(cfg_s)
    cfg["pretrain"]["normalization"] = "l2"
    tf = AugmentationModule(cfg, 1000, max_batch=6)
    assert tf.pre_norm is None                                   # l2 replaces the running mean/var normalisation
    front = UpstreamFrontEnd(cfg, tf)
    w = fill.uniform((6, 16000), 91, -0.4, 0.4)
    w[4] = 0.0
    w[5] = 1.0
    got = front.log_mel(torch.from_numpy(w).cuda()).cpu()
    wn = torch.nn.functional.normalize(torch.from_numpy(w), dim=-1, p=2)
    ref = FE.log_mel_batch(wn)
    assert got.shape == (6, 64, 101)
    assert float((got.exp() - ref.exp()).abs().max()) <= 1e-5 * float(ref.exp().max()) + 1e-7
    v1, v2 = front(torch.from_numpy(w).cuda())
    assert v1.shape == v2.shape == (6, 1, 64, 101) and bool(torch.isfinite(v1).all()) and bool(torch.isfinite(v2).all())


def test_logmel_tables_match_oracle():
    from src.utils.utils import slaney_mel_filterbank, pack_filterbank
    fb = slaney_mel_filterbank(16000, 1024, 64, 60, 7800)
    assert np.array_equal(fb, FE.mel_filterbank())
    st, pk = pack_filterbank(fb)
    assert pk.shape[1] == 45 and int((pk != 0).sum()) == 966


# ------------------------------------------------------------------------------------------------ augmentation
@pytest.mark.parametrize("T", [101, 96])
def test_aug_two_views_match_reference_goldens(N, golden, cfg_s, T):
    """Batched GPU path vs (a) the reference's own outputs (golden) and (b) the sequential oracle."""
    from src.augmentations import AugmentationModule
    g = golden(f"aug_T{T}")
    np.random.seed(31)
    random.seed(31)
    tf = AugmentationModule(cfg_s, 100, max_batch=8)
    xs = torch.stack([torch.from_numpy(fill.normalish((1, 64, T), 1000 + c) * 3.0 - 8.0) for c in range(20)])
    v1s, v2s, ijhw = [], [], []
    for lo, hi in ((0, 1), (1, 8), (8, 13), (13, 20)):          # ragged batch sizes, state carried across
        a, b = tf.augment_batch(xs[lo:hi].cuda())
        v1s.append(a.cpu())
        v2s.append(b.cpu())
        ijhw.append(tf.last_plan[0][:, :, 2:6].reshape(-1, 4))
    v1, v2 = torch.cat(v1s)[:, 0].numpy(), torch.cat(v2s)[:, 0].numpy()
    assert np.array_equal(np.concatenate(ijhw), g["ijhw"])                      # bit-exact crop indices
    assert np.random.random() == float(g["np_state_after"]) and random.random() == float(g["py_state_after"])
    for k, c in enumerate(g["keep"]):
        np.testing.assert_allclose(v1[c], g["v1"][k], rtol=0, atol=2e-5)
        np.testing.assert_allclose(v2[c], g["v2"][k], rtol=0, atol=2e-5)
    dig = np.array([[v.sum(dtype=np.float64), np.abs(v).sum(dtype=np.float64)] for v in list(v1) + list(v2)])
    np.testing.assert_allclose(dig, g["digest"], rtol=2e-5, atol=2e-2)


@pytest.mark.parametrize("B,n0,max_update", [(512, 0, 10 ** 9), (512, 1, 10 ** 9), (512, 5000, 5300), (3840, 123456, 10 ** 9),
                                             (70, 7, 7), (64, 0, 1)])
def test_runnorm_scan_is_the_fp32_recurrence_bit_for_bit(N, B, n0, max_update):
    """RunningNorm state after every clip of a batch (`augmentations.py:215-282`: mu += (m - mu) / n in fp32, the variance
    sample uses the updated mean, updates stop at max_update) - the pipelined kernel against a numpy float32 loop: EXACT."""
    n_elem = 64 * 101
    g = np.random.RandomState(B + n0)
    ex = g.randn(B) * 0.7 - 4.0
    ex2 = ex ** 2 + np.abs(g.randn(B)) * 3.0 + 0.1
    mom = np.stack([ex * n_elem, ex2 * n_elem], 1).astype(np.float64)
    mu0, s20 = np.float32(-3.9), np.float32(2.5)
    st_i = torch.tensor([n0, max_update], dtype=torch.int64).cuda()
    st_f = torch.tensor([mu0, s20], dtype=torch.float32).cuda()
    mu_out, sd_out = torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    N.call("runnorm_scan", dev(mom), B, n_elem, st_i, st_f, mu_out, sd_out)
    torch.cuda.synchronize()
    inv = 1.0 / n_elem
    n, mu, s2 = n0, mu0, s20
    want_mu, want_sd = np.empty(B, np.float32), np.empty(B, np.float32)
    for c in range(B):
        if n < max_update:
            e, e2 = mom[c, 0] * inv, mom[c, 1] * inv
            m = np.float32(e)
            mu = m if n == 0 else np.float32(mu + np.float32(np.float32(m - mu) / np.float32(n)))
            v = np.float32(e2 - 2.0 * float(mu) * e + float(mu) * float(mu))
            s2 = v if n == 0 else np.float32(s2 + np.float32(np.float32(v - s2) / np.float32(n)))
            n += 1
        want_mu[c] = mu
        want_sd[c] = min(max(np.sqrt(np.float32(s2)), np.float32(1.1920928955078125e-07)), np.finfo(np.float32).max)
    assert np.array_equal(mu_out.cpu().numpy(), want_mu)
    assert np.array_equal(sd_out.cpu().numpy(), want_sd)
    assert st_i.cpu().tolist() == [n, max_update]
    assert np.array_equal(st_f.cpu().numpy(), np.array([mu, s2], np.float32))


def test_kmix_vs_reference_golden(N, golden):
    """Kmix (`src/augmentations/augmentations.py:119-189`) against the reference class run on the same 170 closed-form views:
    every index draw - randint(len(bank)) below 128 entries, randint(len(l)) of the farthest non-empty cluster above - with its
    argument AND value (bit-exact: the arguments depend on the cluster ids computed on the GPU), outputs <= 2e-5, the numpy
    stream position afterwards; once call by call (`forward`), once as two batches (`mix_batch`: one cluster launch, one
    device->host copy, one mixing launch per batch)."""
    from src.augmentations.augmentations import Kmix
    g = golden("kmix")
    F, T, calls = 64, 24, 170
    xs = [torch.from_numpy(fill.normalish((1, F, T), 5000 + c)) * (0.5 + 0.1 * (c % 7)) + 0.2 * (c % 5) for c in range(calls)]
    want = [tuple(p) for p in g["picks"].tolist() if p[1] >= 0]
    for mode in ("calls", "batches"):
        km = Kmix(ratio=0.4, n_memory=140, log_mixup_exp=True, top_k=16, centroids=torch.from_numpy(g["centroids"]))
        np.random.seed(77)
        if mode == "calls":
            ys = [km(x.cuda()).cpu() for x in xs]
        else:
            a = km.mix_batch(torch.cat(xs[:50]).cuda()).cpu()
            b = km.mix_batch(torch.cat(xs[50:]).cuda()).cpu()
            ys = [y[None] for y in torch.cat([a, b])]
        assert km.draws == want, (mode, [i for i, (p, q) in enumerate(zip(km.draws, want)) if p != q][:5])
        assert np.random.random() == float(g["tail"])
        for c in (0, 1, 5, 127, 128, 129, 141, 169):
            assert float((ys[c] - torch.from_numpy(g[f"y{c}"])).abs().max()) <= 2e-5, (mode, c)
        assert len(km.bank_ids) == 140 and km.entries == calls


def test_kmix_inside_the_augmentation_module(N, cfg_s):
    """`Kmix` as the third stage of the two-view pipeline (`src/augmentations/__init__.py:18-30`): views stay finite, the bank
    fills with 2 entries per clip in call order, PatchDrop is still refused."""
    import copy
    from src.augmentations import AugmentationModule
    cfg = copy.deepcopy(cfg_s)
    cent = torch.from_numpy(fill.normalish((16, 64), 4243)) + 0.2
    cfg["pretrain"]["augmentations"]["Kmix"] = {"ratio": 0.4, "log_mixup_exp": True, "top_k": 16, "centroids": cent}
    tf = AugmentationModule(cfg, 1000, max_batch=96)
    np.random.seed(5); random.seed(5)
    lms = torch.from_numpy(fill.normalish((96, 64, 101), 31) * 2.0 - 4.0).cuda()
    v1, v2 = tf.augment_batch(lms)
    assert v1.shape == v2.shape == (96, 1, 64, 101) and bool(torch.isfinite(v1).all()) and bool(torch.isfinite(v2).all())
    assert tf.kmix.entries == 192 and len(tf.kmix.draws) == 191            # every call but the very first mixes
    assert tf.kmix.draws[126][0] == 127 and tf.kmix.draws[127][0] < 128    # call 128 is the first cluster-guided one
    cfg["pretrain"]["augmentations"]["PatchDrop"] = {"ratio": 0.1}
    with pytest.raises(NotImplementedError):
        AugmentationModule(cfg, 1000)


def test_aug_fifo_wraparound_vs_oracle(N, cfg_s):
    """More clips than the 2048-entry FIFO holds: partner resolution across the ring wrap, small images."""
    from src.augmentations import AugmentationModule
    F_, T_ = 8, 12
    n_clips = 1100
    xs = torch.from_numpy(fill.normalish((n_clips, 1, F_, T_), 77) * 2.0 - 3.0)
    np.random.seed(7); random.seed(7)
    ref = OA.AugmentationModule(cfg_s, 1000)
    r1, r2 = zip(*[ref(xs[c]) for c in range(n_clips)])
    np.random.seed(7); random.seed(7)
    tf = AugmentationModule(cfg_s, 1000, max_batch=64)
    o1, o2 = [], []
    for lo in range(0, n_clips, 50):
        a, b = tf.augment_batch(xs[lo:lo + 50].cuda())
        o1.append(a.cpu()); o2.append(b.cpu())
    o1, o2 = torch.cat(o1), torch.cat(o2)
    np.testing.assert_allclose(o1[:, 0].numpy(), torch.cat(r1).numpy(), rtol=0, atol=5e-5)
    np.testing.assert_allclose(o2[:, 0].numpy(), torch.cat(r2).numpy(), rtol=0, atol=5e-5)


def test_specaugment_masks_bit_exact(N, golden, cfg_s):
    import copy
    from src.augmentations import AugmentationModule
    g = golden("specaug")
    for zero in (False, True):
        cfg = copy.deepcopy(cfg_s)
        cfg["pretrain"]["augmentations"] = {"SpecAugment": dict(F=30, T=40, num_freq_masks=2, num_time_masks=2,
                                                                replace_with_zero=zero)}
        cfg["pretrain"]["normalization"] = "l2"
        for seed in range(8):
            x = torch.from_numpy(fill.normalish((101, 64), 3000 + seed))          # (T, dim) as the reference takes it
            random.seed(1234 + seed)
            tf = AugmentationModule(cfg, 10, max_batch=4)
            v1, _ = tf.augment_batch(x.T.contiguous()[None].cuda())              # module layout is [F, T]
            want = g["outs"][2 * seed + (1 if zero else 0)]
            got = v1[0, 0].cpu().numpy().T
            if zero:
                assert np.array_equal(got, want)
            else:
                np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)


# ------------------------------------------------------------------------------------------------ stem (conv1)
@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("T", [101, 96])
def test_conv1_block_fwd_bwd(N, dtype, T):
    import torch.nn as nn
    Nimg, F_ = 3, 64
    td = N.torch_dtype(dtype)
    blk = nn.Sequential(nn.Conv2d(1, 64, 3, padding=1), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(2, 2))
    fill.fill_state_dict_(blk, seed=5)
    blk.train()
    x = torch.from_numpy(fill.normalish((Nimg, 1, F_, T), 41))
    ref = blk(x)                                                     # [N, 64, F/2, T/2]
    gP = torch.from_numpy(fill.uniform(tuple(ref.shape), 42))
    gX1 = torch.from_numpy(fill.uniform((Nimg, 32 * 64), 43))         # grad of the temporal mean x_1
    x1 = ref.permute(0, 3, 2, 1).reshape(Nimg, T // 2, -1).mean(1)
    ((ref * gP).sum() + (x1 * gX1).sum()).backward()

    conv, bn = blk[0], blk[1]
    w = dev(conv.weight.detach().reshape(64, 9)); b = dev(conv.bias.detach())
    gamma, beta = dev(bn.weight.detach()), dev(bn.bias.detach())
    rm, rv = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
    mom = torch.empty(16 * 54, dtype=torch.float64, device="cuda")
    scale, shift, mean, rstd = (torch.empty(64, device="cuda") for _ in range(4))
    img = dev(x[:, 0])
    N.call("conv1_stats", img, Nimg, F_, T, w, b, gamma, beta, rm, rv, 0.1, 1e-5, mom, scale, shift, mean, rstd)
    P = torch.empty(Nimg, T // 2, F_ // 2, 64, device="cuda", dtype=td)
    xl = torch.full((4, Nimg, (F_ // 2) * 64), float("nan"), device="cuda") if dtype == 1 else None
    N.call("conv1_fwd", dtype, img, Nimg, F_, T, w, b, scale, shift, P, xl)
    torch.cuda.synchronize()
    if xl is not None:              # fused layer output x_1 = mean over time of the pooled map as stored, left as four partial sums
        want = P.float().mean(1).reshape(Nimg, -1)
        np.testing.assert_allclose(xl.sum(0).cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-6)
        x1 = torch.full_like(want, float("nan"))
        P2 = torch.randn(Nimg, 6, 16, 64, device="cuda").bfloat16(); P3 = torch.randn(Nimg, 3, 8, 64, device="cuda").bfloat16()
        x2, x3 = torch.empty(Nimg, 16 * 64, device="cuda"), torch.empty(Nimg, 8 * 64, device="cuda")
        N.call("tmean3_fwd", 1, 1, None, x1, xl, T // 2, F_ // 2, P2, x2, 6, 16, P3, x3, 3, 8, Nimg)
        torch.cuda.synchronize()
        assert torch.equal(x1, (xl[0] + xl[1]) + xl[2] + xl[3])     # the parts are added in this fixed order
        np.testing.assert_allclose(x2.cpu().numpy(), P2.float().mean(1).reshape(Nimg, -1).cpu().numpy(), rtol=1e-5, atol=1e-6)
    ref_cl = ref.detach().permute(0, 3, 2, 1)                       # [N, T/2, F/2, 64]
    assert rel_l2(P.float().cpu(), ref_cl) < (1e-5 if dtype == 0 else 1e-2)         # dtype 1: bf16 MFMA operands + bf16 output
    np.testing.assert_allclose(rm.cpu().numpy(), bn.running_mean.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), bn.running_var.numpy(), rtol=1e-4, atol=1e-6)

    dP = dev(gP.permute(0, 3, 2, 1))                  # gradients entering the stem backward are fp32 in both modes
    dxl = dev(gX1)
    acc = torch.empty(32 * 64 * 11, device="cuda")
    dW, db, dg, dbt = (torch.zeros(s, device="cuda") for s in ((64, 9), (64,), (64,), (64,)))
    N.call("conv1_bwd", 0, dtype, img, Nimg, F_, T, w, b, gamma, scale, shift, mean, rstd, mom, dP, dxl, acc, dW, db, dg, dbt)
    torch.cuda.synchronize()
    # dtype 1: the conv is recomputed as a bf16 MFMA, so a few pooling windows with near-tied candidates route their gradient
    # to a different position than the fp32 reference does (measured 5.6 % on dW here; the tap sums themselves are hi+lo exact)
    tol = 2e-4 if dtype == 0 else 8e-2
    assert rel_l2(dW.cpu(), conv.weight.grad.reshape(64, 9)) < tol
    assert rel_l2(dg.cpu(), bn.weight.grad) < tol
    assert rel_l2(dbt.cpu(), bn.bias.grad) < tol
    assert float(conv.bias.grad.abs().max()) < 1e-3 and float(db.abs().max()) == 0.0
    if dtype == 1:
        # the MFMA flavour also takes the pooled gradient in bf16 (dxl stays fp32): same result up to the rounding of dP
        dW2, db2, dg2, dbt2 = (torch.zeros(s, device="cuda") for s in ((64, 9), (64,), (64,), (64,)))
        N.call("conv1_bwd", 1, 1, img, Nimg, F_, T, w, b, gamma, scale, shift, mean, rstd, mom, dP.bfloat16().contiguous(), dxl, acc,
               dW2, db2, dg2, dbt2)
        torch.cuda.synchronize()
        assert rel_l2(dW2.cpu(), dW.cpu()) < 5e-3 and rel_l2(dg2.cpu(), dg.cpu()) < 5e-3 and rel_l2(dbt2.cpu(), dbt.cpu()) < 5e-3


@pytest.mark.parametrize("shape", [(5, 40, 300), (2, 64, 129), (3, 8, 2), (7, 64, 101), (2, 120, 128)])
def test_conv1_tap_moments(N, shape):
    """The 9 first + 45 second moments of the shifted copies of the input the stem's batch statistics are built from
    (csrc/conv1.hip): time chunks of 128 columns with a halo, chunk boundaries inside and at the end of the image."""
    import torch.nn.functional as Fnn
    Nimg, F_, T = shape
    x = torch.from_numpy(fill.normalish((Nimg, 1, F_, T), 7 + T)).double()
    taps = Fnn.unfold(x, 3, padding=1)                               # [N, 9, F*T], tap order kh*3+kw
    s1 = taps.sum((0, 2))
    s2 = torch.einsum("nap,nbp->ab", taps, taps)
    want = torch.cat([s1, torch.stack([s2[a, b] for a in range(9) for b in range(a, 9)])])
    mom = torch.empty(16 * 54, dtype=torch.float64, device="cuda")
    N.call("conv1_moments", x[:, 0].float().cuda().contiguous(), Nimg, F_, T, mom)
    torch.cuda.synchronize()
    np.testing.assert_allclose(mom[:54].cpu().numpy(), want.numpy(), rtol=2e-5, atol=1e-3)


# ------------------------------------------------------------------------------------------------ implicit-GEMM conv
@pytest.mark.parametrize("shape", [(3, 50, 32), (2, 48, 32), (3, 25, 16), (5, 24, 16), (1, 3, 32),
                                   (80, 50, 32), (600, 25, 16), (261, 24, 16)])
def test_conv3x3_implicit_gemm_fwd_dgrad_wgrad(N, shape):
    """bf16 implicit-GEMM conv vs torch conv2d on the same (bf16-exact) operands: forward + fused BN statistics,
    data gradient, weight gradient.  Tiles end inside images and images end inside tiles for these shapes.
    The last three shapes have 500 / 938 / 392 tiles of 256 pixels: more than the 256 persistent workgroups, so every
    workgroup walks several tiles (`tile += gridDim.x` with the next tile's halo prefetched) and the weight gradient folds
    256 per-workgroup partial results - the path the B = 512 step runs."""
    import torch.nn.functional as Fnn
    Nimg, Ti, Fi = shape
    x = torch.from_numpy(fill.normalish((Nimg, Ti, Fi, 64), 61 + Ti)).cuda().bfloat16()          # [N][T][F][C]
    w = torch.from_numpy(fill.uniform((64, 64, 3, 3), 62, -0.05, 0.05)).cuda()
    b = torch.from_numpy(fill.uniform((64,), 63)).cuda()
    Wf = torch.empty(64, 576, device="cuda", dtype=torch.bfloat16)
    Wd = torch.empty_like(Wf)
    N.call("pack_conv_w", 1, w, Wf, Wd)
    w2 = torch.from_numpy(fill.uniform((64, 64, 3, 3), 66, -0.05, 0.05)).cuda()
    pk = [torch.empty_like(Wf) for _ in range(4)]
    N.call("pack_conv_w2", 1, w, pk[0], pk[1], w2, pk[2], pk[3])            # both layers in one launch = two single packs
    Wf2, Wd2 = torch.empty_like(Wf), torch.empty_like(Wf)
    N.call("pack_conv_w", 1, w2, Wf2, Wd2)
    torch.cuda.synchronize()
    assert torch.equal(pk[0], Wf) and torch.equal(pk[1], Wd) and torch.equal(pk[2], Wf2) and torch.equal(pk[3], Wd2)
    wq = Wf.float().view(64, 9, 64).permute(0, 2, 1).reshape(64, 64, 3, 3)                        # bf16-rounded [co][ci][kh][kw]
    x_nchw = x.float().permute(0, 3, 2, 1).contiguous()                                           # [N][C][F][T]
    ref = Fnn.conv2d(x_nchw.double(), wq.double(), b.double(), padding=1)                         # [N][co][F][T]
    Y = torch.full((Nimg, Ti, Fi, 64), float("nan"), device="cuda", dtype=torch.bfloat16)
    sq = torch.empty(2, 64, dtype=torch.float64, device="cuda")
    N.call("conv3x3_fwd", x, Wf, b, Y, 0, sq[0], sq[1], 1, Nimg, Ti, Fi)
    torch.cuda.synchronize()
    ref_cl = ref.permute(0, 3, 2, 1)
    assert rel_l2(Y.float().cpu(), ref_cl.cpu()) < 4e-3                                           # bf16 output rounding only
    # the fused batch statistics are those of the tensor AS STORED (bf16): exact sums of the kernel's own output, and within
    # rounding noise (2^-9 relative per element, averaging out over the pixels) of the unrounded convolution's statistics
    Yd = Y.double()
    np.testing.assert_allclose(sq[0].cpu().numpy(), Yd.sum((0, 1, 2)).cpu().numpy(), rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(sq[1].cpu().numpy(), (Yd ** 2).sum((0, 1, 2)).cpu().numpy(), rtol=1e-5)
    npix = Nimg * Ti * Fi
    mean_ref, mean = ref_cl.mean((0, 1, 2)).cpu(), sq[0].cpu() / npix
    var_ref, var = ref_cl.var((0, 1, 2), unbiased=False).cpu(), sq[1].cpu() / npix - mean ** 2
    assert float(((mean - mean_ref).abs() / var_ref.sqrt()).max()) < 2e-3 / np.sqrt(npix / 4800.0) + 1e-4
    assert float(((var - var_ref).abs() / var_ref).max()) < 2e-3
    Y32 = torch.full((Nimg, Ti, Fi, 64), float("nan"), device="cuda", dtype=torch.float32)        # fp32-output variant (bf16_hp)
    N.call("conv3x3_fwd", x, Wf, b, Y32, 1, None, None, 1, Nimg, Ti, Fi)
    assert rel_l2(Y32.cpu(), ref_cl.cpu()) < 1e-5
    # data gradient = the same kernel on dY with the flipped / transposed weights
    dy = torch.from_numpy(fill.normalish((Nimg, Ti, Fi, 64), 64 + Ti)).cuda().bfloat16()
    dx = torch.empty(Nimg, Ti, Fi, 64, device="cuda", dtype=torch.float32)       # fp32 output variant
    N.call("conv3x3_fwd", dy, Wd, None, dx, 1, None, None, 1, Nimg, Ti, Fi)
    dy_nchw = dy.float().permute(0, 3, 2, 1).contiguous().double()
    ref_dx = torch.nn.grad.conv2d_input(x_nchw.shape, wq.double(), dy_nchw, padding=1).permute(0, 3, 2, 1)
    assert rel_l2(dx.cpu(), ref_dx.cpu()) < 1e-5
    # weight gradient
    ref_dw = torch.nn.grad.conv2d_weight(x_nchw.double(), wq.shape, dy_nchw, padding=1)
    ws = torch.empty(2 * 256 * 64 * 576, device="cuda")
    for workspace in (None, ws):                                   # fp32 atomics / per-workgroup results + fold
        dWp = torch.zeros(64, 576, device="cuda")
        N.call("conv3x3_wgrad", dy, x, dWp, workspace, 0 if workspace is None else workspace.numel(), Nimg, Ti, Fi)
        dW = torch.zeros(64, 64, 3, 3, device="cuda")
        N.call("unpack_conv_dw", dWp, dW)
        torch.cuda.synchronize()
        assert rel_l2(dW.cpu(), ref_dw.cpu()) < 1e-5


# ----------------------------------------------------------------- BatchNorm2d(train) + ReLU + MaxPool2d(2) backward
@pytest.mark.parametrize("shape", [(5, 24, 16), (3, 25, 32), (40, 50, 32)])
def test_bn_relu_pool_bwd_sums_from_the_pooled_output(N, shape):
    """`bn_relu_pool_bwd_p` (dbeta / dgamma from the pooled forward output) against `bn_relu_pool_bwd` (a first sweep over the
    conv output) and against torch autograd through BatchNorm2d(train) -> ReLU -> MaxPool2d(2) in fp64 on the same bf16 conv
    output.  Odd Ti: the last time row is outside the pooled area."""
    Nimg, Ti, Fi = shape
    To, Fo = Ti // 2, Fi // 2
    g = torch.Generator().manual_seed(11 + Ti)
    Y = (torch.randn(Nimg, Ti, Fi, 64, generator=g) * 1.3 + 0.2).cuda().bfloat16()
    gamma = (torch.rand(64, generator=g) + 0.5).cuda()
    gamma[3] = -0.7                                                  # a negative scale: the window's arg-max is the conv output's arg-min
    beta = (torch.randn(64, generator=g) * 0.3).cuda()
    dP = torch.randn(Nimg, To, Fo, 64, generator=g).cuda()
    dxl = torch.randn(Nimg, Fo, 64, generator=g).cuda()
    # reference (fp64 autograd)
    y64 = Y.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mean, var = y64.mean((0, 1, 2)), y64.var((0, 1, 2), unbiased=False)
    a = (y64 - mean) / torch.sqrt(var + 1e-5) * g64 + b64
    # torch's layout is [N][C][F][T] (audiontt.py): ties inside a window (a few % of the bf16 windows) go to the first maximum in (f, t) order
    P64 = torch.nn.functional.max_pool2d(torch.relu(a).permute(0, 3, 2, 1), 2).permute(0, 3, 2, 1)     # -> [N][To][Fo][64]
    (P64 * dP.double()).sum().add((P64.mean(1) * dxl.double()).sum()).backward()
    rstd = 1.0 / torch.sqrt(var.detach() + 1e-5)
    scale = (gamma.double() * rstd).float().contiguous()
    shift = (beta.double() - mean.detach() * gamma.double() * rstd).float().contiguous()
    mean32, rstd32 = mean.detach().float().contiguous(), rstd.float().contiguous()
    Pb = P64.detach().bfloat16().contiguous()                       # what the forward kernel stores
    outs = []
    for entry in ("bn_relu_pool_bwd", "bn_relu_pool_bwd_p"):
        stat = torch.empty(33 * 128, device="cuda")
        dY = torch.empty(Nimg, Ti, Fi, 64, device="cuda", dtype=torch.bfloat16)
        dg, db = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
        args = (N.BF16, N.BF16, N.F32, Y) + ((Pb,) if entry.endswith("_p") else ()) + (dP, dxl, scale, shift, mean32, rstd32, stat, dY, dg, db,
                                                                                        Nimg, Ti, Fi)
        N.call(entry, *args)
        torch.cuda.synchronize()
        outs.append((dg.cpu(), db.cpu(), dY.float().cpu()))
        assert rel_l2(dg.cpu(), g64.grad.float().cpu()) < 4e-3, entry
        assert rel_l2(db.cpu(), b64.grad.float().cpu()) < 4e-3, entry
        assert rel_l2(dY.float().cpu(), y64.grad.float().cpu()) < 8e-3, entry
    (dg0, db0, dY0), (dg1, db1, dY1) = outs
    assert rel_l2(dg1, dg0) < 2e-3 and rel_l2(db1, db0) < 2e-3 and rel_l2(dY1, dY0) < 4e-3


# ------------------------------------------------------------------------- fused BatchNorm1d(train) of the projector
@pytest.mark.parametrize("adtype,gdtype", [(1, 0), (0, 0), (1, 1)])
@pytest.mark.parametrize("M,C,affine,relu", [(512, 256, True, 1), (200, 64, False, 0), (1000, 96, True, 1)])
def test_colbn_train_fused_fwd_bwd_vs_torch(N, adtype, gdtype, M, C, affine, relu):
    """colbn_train_fwd / the single-launch colbn_bwd against torch's train-mode batch_norm (fp32, autograd) on two groups
    that share the layer: statistics per group, running statistics updated group after group."""
    G = 2
    g = torch.Generator().manual_seed(5)
    a32 = (torch.randn(G, M, C, generator=g) * 1.7 + 0.4).cuda()
    a = a32.to(N.torch_dtype(adtype)).contiguous()
    aref = a.float().clone().requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).cuda() if affine else None
    beta = (torch.randn(C, generator=g) * 0.1).cuda() if affine else None
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    rm_ref, rv_ref = rm.clone(), rv.clone()
    gref = gamma.clone().requires_grad_(True) if affine else None
    bref = beta.clone().requires_grad_(True) if affine else None
    outs = []
    for k in range(G):
        y = torch.nn.functional.batch_norm(aref[k], rm_ref, rv_ref, gref, bref, True, 0.1, 1e-5)
        outs.append(torch.relu(y) if relu else y)
    href = torch.stack(outs)
    h = torch.empty(G, M, C, device="cuda", dtype=torch.bfloat16)
    st = torch.empty(4, G * C, device="cuda")
    N.call("colbn_train_fwd", 1, adtype, a, gamma, beta, rm, rv, 0.1, 1e-5, relu, G, M, C, h, st[0], st[1], st[2], st[3])
    torch.cuda.synchronize()
    assert rel_l2(h.float().cpu(), href.detach().cpu()) < 4e-3
    assert rel_l2(rm.cpu(), rm_ref.cpu()) < 1e-5 and rel_l2(rv.cpu(), rv_ref.cpu()) < 1e-5
    # the separate kernels produce the same statistics
    sq = torch.zeros(2, G * C, device="cuda", dtype=torch.float64)
    st2 = torch.empty(4, G * C, device="cuda")
    if C % 64 == 0:
        N.call("colstats", adtype, a, G, M, C, C, 1, sq[0], sq[1])
        N.call("bn_finalize", sq[0], sq[1], G, float(M), C, gamma, beta, None, None, 0.1, 1e-5, st2[0], st2[1], st2[2], st2[3])
        torch.cuda.synchronize()
        assert rel_l2(st.cpu(), st2.cpu()) < 1e-5
    # backward
    dh32 = torch.randn(G, M, C, generator=g).cuda()
    dh = dh32.to(N.torch_dtype(gdtype)).contiguous()
    href.backward(dh.float())
    da = torch.empty(G, M, C, device="cuda", dtype=torch.bfloat16)
    dg, db = (torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")) if affine else (None, None)
    tmp = torch.zeros(2 * G * C, device="cuda", dtype=torch.float64)
    if C % 64 == 0:
        N.call("colbn_bwd", 1, adtype, gdtype, a, dh, st[0], st[1], st[2], st[3], relu, G, M, C, tmp, da, dg, db)
        torch.cuda.synchronize()
        assert rel_l2(da.float().cpu(), aref.grad.cpu()) < 6e-3
        if affine:
            assert rel_l2(dg.cpu(), gref.grad.cpu()) < 1e-4 and rel_l2(db.cpu(), bref.grad.cpu()) < 1e-4


# ------------------------------------------------------------------------------------ GEMM: multi-problem and tile variants
def test_center_cast_and_running_mean_shift(N):
    """Centred bf16 cast of the pooled features + the running-mean repair, and the identity they rest on: a bias-free Linear
    followed by train-mode BatchNorm gives the same output, the same weight gradient and (after the repair) the same
    running mean on centred inputs as on the raw ones (`delores_s/upstream_expert.py:15-22`)."""
    G, D, mom = 2, 96, 0.1
    for M, K in [(512, 64), (300, 2048), (1000, 512)]:
        g = torch.Generator().manual_seed(M + K)
        y = (torch.rand(G, M, K, generator=g) * 0.3 + 5.0 * torch.rand(K, generator=g)).cuda()      # |column mean| >> std
        yc = torch.empty(G, M, K, device="cuda", dtype=torch.bfloat16)
        cm = torch.empty(G, K, device="cuda")
        N.call("center_cast", y, yc, cm, G, M, K)
        torch.cuda.synchronize()
        mu = y.double().mean(1)
        np.testing.assert_allclose(cm.cpu().numpy(), mu.cpu().numpy(), rtol=1e-6)
        want = y - mu[:, None, :].float()
        assert float((yc.float() - want).abs().max()) <= 2 ** -8 * float(want.abs().max())          # one bf16 rounding of the CENTRED value
        W = (torch.randn(D, K, generator=g) * 0.05).cuda().bfloat16()
        rm_ref = torch.zeros(D, device="cuda", dtype=torch.float64)
        rm_c = torch.zeros(D, device="cuda")
        for k in range(G):                                   # running mean group after group, raw vs centred inputs
            rm_ref = (1 - mom) * rm_ref + mom * (y[k].double() @ W.double().T).mean(0)
            rm_c = (1 - mom) * rm_c + mom * (want[k] @ W.float().T).mean(0)
        N.call("shift_running_mean", W, cm, rm_c, D, K, G, mom)
        torch.cuda.synchronize()
        np.testing.assert_allclose(rm_c.cpu().numpy(), rm_ref.cpu().numpy(), rtol=2e-4, atol=1e-5)
        # BatchNorm output and weight gradient are unchanged by the shift
        a_raw, a_c = y[0].double() @ W.double().T, want[0].double() @ W.double().T
        bn = lambda a: (a - a.mean(0)) / a.std(0, unbiased=False)
        assert rel_l2(bn(a_c).cpu(), bn(a_raw).cpu()) < 1e-5
        da = torch.randn(M, D, generator=g).cuda().double()
        da = da - da.mean(0)                                 # a BatchNorm input gradient sums to zero over the batch
        assert rel_l2((da.T @ want[0].double()).cpu(), (da.T @ y[0].double()).cpu()) < 1e-5


def test_multi_head_launches_equal_single_ones(N):
    """center_cast_multi / shift_running_mean_multi / tmean3_fwd are the per-head (per-layer) launches folded into one grid:
    bit-identical to the single-problem entry points at the widths DeLoRes-M uses (2048 / 1024 / 512; F = 32 / 16 / 8)."""
    import ctypes
    from src import engine as E
    g = torch.Generator().manual_seed(5)
    G, M, D, mom = 2, 384, 96, 0.1
    Ks = [2048, 1024, 512]
    ys = [(torch.rand(G * M, K, generator=g) + 3.0).cuda() for K in Ks]
    one = [(torch.empty(G * M, K, device="cuda", dtype=torch.bfloat16), torch.empty(G, K, device="cuda")) for K in Ks]
    for y, (yc, cm), K in zip(ys, one, Ks):
        N.call("center_cast", y, yc, cm, G, M, K)
    ycs = [torch.empty_like(o[0]) for o in one]
    cms = [torch.empty_like(o[1]) for o in one]
    vp, adr = ctypes.c_void_p, ctypes.addressof
    a = (E._harr(vp, ys), E._harr(vp, ycs), E._harr(vp, cms), E._harr(ctypes.c_int, Ks))
    N.call("center_cast_multi", 3, adr(a[0]), adr(a[1]), adr(a[2]), adr(a[3]), G, M)
    torch.cuda.synchronize()
    for (yc, cm), yc2, cm2 in zip(one, ycs, cms):
        assert torch.equal(yc, yc2) and torch.equal(cm, cm2)
    Ws = [(torch.randn(D, K, generator=g) * 0.05).cuda().bfloat16() for K in Ks]
    rm1 = [torch.full((D,), 0.25, device="cuda") for _ in Ks]
    rm2 = [t.clone() for t in rm1]
    for W, cm, rm, K in zip(Ws, cms, rm1, Ks):
        N.call("shift_running_mean", W, cm, rm, D, K, G, mom)
    a = (E._harr(vp, Ws), E._harr(vp, cms), E._harr(vp, rm2), E._harr(ctypes.c_int, Ks))
    N.call("shift_running_mean_multi", 3, adr(a[0]), adr(a[1]), adr(a[2]), D, adr(a[3]), G, mom)
    torch.cuda.synchronize()
    for r1, r2 in zip(rm1, rm2):
        assert torch.equal(r1, r2) and float((r1 - 0.25).abs().max()) > 0
    Nimg = 6
    for dt, td, o32 in ((N.F32, torch.float32, 1), (N.BF16, torch.bfloat16, 1), (N.BF16, torch.bfloat16, 0)):
        dims = [(48, 32), (24, 16), (12, 8)]
        Ps = [torch.randn(Nimg, T, F, 64, generator=g).cuda().to(td) for T, F in dims]
        to = torch.float32 if o32 else td
        x1 = [torch.empty(Nimg, F * 64, device="cuda", dtype=to) for _, F in dims]
        x3 = [torch.full_like(t, -1.0) for t in x1]
        for P, x, (T, F) in zip(Ps, x1, dims):
            N.call("tmean_fwd", dt, o32, P, x, Nimg, T, F)
        N.call("tmean3_fwd", dt, o32, Ps[0], x3[0], None, *dims[0], Ps[1], x3[1], *dims[1], Ps[2], x3[2], *dims[2], Nimg)
        torch.cuda.synchronize()
        for u, v in zip(x1, x3):
            assert torch.equal(u, v)


def test_gemm_multi_barlow_epilogue(N):
    """Cross-correlation GEMM of three heads with the Barlow loss and its gradient folded into the epilogue
    (`delores_s/upstream_expert.py:118-131`): dc = dscale (c - I) in bf16, loss replicas summing to coef * sum (c - I)^2,
    c = A^T B / rows never stored."""
    import ctypes
    from src import engine as E
    g = torch.Generator().manual_seed(23)
    D, nh = 256, 3
    Ks = [512, 512, 512]
    A = [(torch.randn(k, D, generator=g) * 0.7).cuda().bfloat16() for k in Ks]
    Bm = [(0.6 * a.float().cpu() + 0.8 * torch.randn(a.shape, generator=g) * 0.7).cuda().bfloat16() for a in A]
    dc = [torch.full((D, D), float("nan"), device="cuda", dtype=torch.bfloat16) for _ in range(nh)]
    rep = torch.zeros(nh, 32, device="cuda")
    coef, dsc = [0.5, 1.0, 2.0], [0.01, 0.02, 0.03]
    vp, fl, adr = ctypes.c_void_p, ctypes.c_float, ctypes.addressof
    arrs = (E._harr(ctypes.c_int, Ks), E._harr(vp, A), E._harr(ctypes.c_long, [D] * nh), E._harr(vp, Bm), E._harr(ctypes.c_long, [D] * nh),
            E._harr(vp, dc), E._harr(fl, coef), E._harr(fl, dsc), E._harr(vp, [rep[h] for h in range(nh)]))
    N.call("gemm_multi_barlow", nh, D, adr(arrs[0]), 1.0 / 512, adr(arrs[1]), adr(arrs[2]), adr(arrs[3]), adr(arrs[4]), adr(arrs[5]),
           adr(arrs[6]), adr(arrs[7]), adr(arrs[8]))
    torch.cuda.synchronize()
    for h in range(nh):
        c = (A[h].double().T @ Bm[h].double()) / 512
        d = c - torch.eye(D, dtype=torch.float64, device="cuda")
        assert abs(float(rep[h].sum()) - coef[h] * float((d * d).sum())) <= 1e-4 * coef[h] * float((d * d).sum())
        assert rel_l2(dc[h].float().cpu(), (dsc[h] * d).float().cpu()) < 5e-3                     # bf16 output rounding


def test_gemm_multi_per_problem_shapes(N):
    """audiossl_gemm_multi with per-problem N / K / leading dimensions (the first projector layer of the three Barlow heads:
    widths 2048 / 1024 / 512) in every transpose mode, plain and exclusive-accumulate outputs."""
    from src import engine as E
    g = torch.Generator().manual_seed(11)
    M = 200
    for ta, tb in ((0, 0), (0, 1), (1, 1)):
        Ns, Ks = [256, 128, 72], [192, 64, 128]
        if ta:
            Ks = [200, 200, 200]                                   # the weight-gradient form: the contraction runs over the batch rows
        As = [torch.randn((k, M) if ta else (M, k), generator=g).cuda().bfloat16() for k in Ks]
        Bs = [torch.randn((k, n) if tb else (n, k), generator=g).cuda().bfloat16() for n, k in zip(Ns, Ks)]
        refs = [(a.float().T if ta else a.float()) @ (b.float() if tb else b.float().T) for a, b in zip(As, Bs)]
        Cs = [torch.empty(M, n, device="cuda", dtype=torch.bfloat16) for n in Ns]
        E.gemm_multi(ta, tb, M, Ns, Ks, As, [a.shape[1] for a in As], Bs, [b.shape[1] for b in Bs], Cs, Ns)
        C0 = [torch.randn(M, n, generator=g).cuda() for n in Ns]
        Cacc = [c.clone() for c in C0]
        E.gemm_multi(ta, tb, M, Ns, Ks, As, [a.shape[1] for a in As], Bs, [b.shape[1] for b in Bs], Cacc, Ns, alpha=0.5, out_f32=1, atomic=2)
        torch.cuda.synchronize()
        for c, c0, ca, r in zip(Cs, C0, Cacc, refs):
            assert rel_l2(c.float().cpu(), r.cpu()) < 4e-3
            assert rel_l2(ca.cpu(), (c0 + 0.5 * r).cpu()) < 1e-5


def test_gemm_multi_hand_scheduled_kernels(N):
    """The multi-problem launches of the grouped Barlow heads at B = 512 (M = 1,024 stacked views) take the hand-scheduled
    256 x 128 / 256 x 256 kernels behind audiossl_gemm_multi's one-dimensional, XCD-contiguous tile order; per-problem widths
    2048 / 1024 / 512 as in the first projector layer (forward, data gradient, weight gradient), bf16 and fp32 outputs."""
    from src import engine as E
    g = torch.Generator().manual_seed(19)
    D, M = 2048, 1024
    cases = [(0, 0, M, [D] * 3, [2048, 1024, 512], 0, 0),        # first layer, forward
             (0, 0, M, [D] * 3, [D] * 3, 0, 0),                  # layers 2, 3
             (0, 1, M, [D] * 3, [D] * 3, 1, 0),                  # data gradients
             (0, 1, M, [2048, 1024, 512], [D] * 3, 1, 0),        # (full-batch form of the last data gradient)
             (1, 1, D, [D] * 3, [M] * 3, 1, 2),                  # weight gradients, exclusive accumulation
             (1, 1, D, [2048, 1024, 512], [M] * 3, 1, 0),        # first layer's weight gradient, store-only
             (1, 1, D, [D] * 3, [512] * 3, 1, 0),                # cross-correlation
             (0, 0, 512, [D] * 3, [D] * 3, 1, 0),                # dzn of view 1 (128 x 128 kernel)
             (0, 1, 512, [D] * 3, [D] * 3, 1, 0),                # dzn of view 2
             (0, 1, 512, [2048, 1024, 512], [D] * 3, 1, 0)]      # last data gradient (view 1 rows only)
    for ta, tb, m, Ns, Ks, f32, atomic in cases:
        As = [(torch.randn((k, m) if ta else (m, k), generator=g) * 0.5).cuda().bfloat16() for k in Ks]
        Bs = [(torch.randn((k, n) if tb else (n, k), generator=g) * 0.5).cuda().bfloat16() for n, k in zip(Ns, Ks)]
        C0 = [torch.randn(m, n, generator=g).cuda().to(torch.float32 if f32 else torch.bfloat16) for n in Ns]
        Cs = [c.clone() for c in C0]
        E.gemm_multi(ta, tb, m, Ns, Ks, As, [a.shape[1] for a in As], Bs, [b.shape[1] for b in Bs], Cs, Ns, out_f32=f32, atomic=atomic)
        torch.cuda.synchronize()
        for a, b, c, c0 in zip(As, Bs, Cs, C0):
            ref = (a.double().T if ta else a.double()) @ (b.double() if tb else b.double().T)
            if atomic:
                ref = ref + c0.double()
            assert rel_l2(c.double().cpu(), ref.cpu()) < (1e-5 if f32 else 4e-3), (ta, tb, m, Ns, Ks)


def test_gemm_dropout_draws_the_mask_the_mask_kernel_writes(N):
    """`gemm_dropout` (Linear -> ReLU -> Dropout with the keep mask drawn in the epilogue) against `gemm` fed with the mask that
    `dropout_mask` writes for the same (seed, p, counter): BIT FOR BIT, on the encoder's fc.0 shapes at two batch sizes (the K-step-32
    kernel with half-height epilogue tiles and the 128-row kernels) and a ragged one (one-column-per-lane epilogue), with and
    without the device counter."""
    from src import engine as E
    g = torch.Generator().manual_seed(29)
    for M, Nn, K, use_counter in ((6144, 2048, 512, True), (768, 2048, 512, True), (200, 264, 72, False)):
        A = (torch.randn(M, K, generator=g) * 0.5).cuda().bfloat16()
        W = (torch.randn(Nn, K, generator=g) * 0.1).cuda().bfloat16()
        bias = torch.randn(Nn, generator=g).cuda()
        counter = torch.tensor([41], dtype=torch.int64, device="cuda") if use_counter else None
        seed, p = 0x1234ABCD5E, 0.3
        keep = torch.empty(M, Nn, dtype=torch.uint8, device="cuda")
        N.call("dropout_mask", keep, M * Nn, seed, p, counter)
        want = torch.empty(M, Nn, device="cuda", dtype=torch.bfloat16)
        E.gemm(N.BF16, 0, 0, M, Nn, K, A, K, W, K, want, Nn, bias=bias, relu=1, keep=keep, ldk=Nn, keep_scale=1.0 / (1.0 - p))
        got = torch.full((M, Nn), float("nan"), device="cuda", dtype=torch.bfloat16)
        N.call("gemm_dropout", 0, 0, M, Nn, K, 1.0, A, K, W, K, got, Nn, bias, 1, seed, p, counter, 1.0 / (1.0 - p))
        torch.cuda.synchronize()
        assert torch.equal(got, want), (M, Nn, K)
        frac = float(keep.float().mean())
        assert abs(frac - (1.0 - p)) < 0.01


def test_gemm_multi_sgd_epilogue_equals_gemm_then_sgd(N):
    """`gemm_multi_sgd` (the optimiser's update in the weight-gradient GEMM's epilogue) against the two-launch form it replaces -
    `gemm_multi` storing the gradient, then `sgd_momentum` with the bf16 shadow - BIT FOR BIT: parameters, momentum, shadow.
    The heads' shapes (256 x 256 tiles, three problems of different width -> 256 x 128 tiles) and a small ragged one."""
    from src import engine as E
    g = torch.Generator().manual_seed(23)
    lr, mu, wd, gs = 0.05, 0.9, 1e-4, 0.5
    for D, ks, M in ((2048, [2048, 2048, 2048], 1024), (2048, [2048, 1024, 512], 1024), (264, [136, 72], 200)):
        nh = len(ks)
        dys = [(torch.randn(M, D, generator=g) * 0.3).cuda().bfloat16() for _ in range(nh)]
        xs = [(torch.randn(M, k, generator=g) * 0.3).cuda().bfloat16() for k in ks]
        P0 = [torch.randn(D, k, generator=g).cuda() for k in ks]
        M0 = [torch.randn(D, k, generator=g).cuda() for k in ks]
        # two launches
        Pa, Ma = [p.clone() for p in P0], [m.clone() for m in M0]
        Ga = [torch.empty(D, k, device="cuda") for k in ks]
        Sa = [torch.empty(D, k, device="cuda", dtype=torch.bfloat16) for k in ks]
        E.gemm_multi(1, 1, D, ks, [M] * nh, dys, [D] * nh, xs, ks, Ga, ks, out_f32=1, atomic=0)
        for p_, g_, m_, s_ in zip(Pa, Ga, Ma, Sa):
            N.call("sgd_momentum", p_, g_, m_, p_.numel(), lr, mu, wd, 0, gs, None, s_, 0)
        # one launch
        Pb, Mb = [p.clone() for p in P0], [m.clone() for m in M0]
        Sb = [torch.empty(D, k, device="cuda", dtype=torch.bfloat16) for k in ks]
        E.gemm_multi_sgd(1, 1, D, ks, [M] * nh, dys, [D] * nh, xs, ks, Pb, Mb, Sb, ks, (lr, mu, wd, gs, None))
        torch.cuda.synchronize()
        for h in range(nh):
            assert torch.equal(Pb[h], Pa[h]) and torch.equal(Mb[h], Ma[h]) and torch.equal(Sb[h], Sa[h]), (D, ks, h)
            assert not torch.equal(Pb[h], P0[h])


_VARIANT_SCRIPT = r"""
import sys, torch
sys.path[:0] = [{root!r}, {pkg!r}]
from src import engine as E
torch.manual_seed(0)
worst = 0.0
for mode, M, Nn, K in {shapes!r}:
    ta, tb = {{"NT": (0, 0), "NN": (0, 1), "TN": (1, 1)}}[mode]
    A = torch.randn((K, M) if ta else (M, K), device="cuda").bfloat16()
    B = torch.randn((K, Nn) if tb else (Nn, K), device="cuda").bfloat16()
    C = torch.empty(M, Nn, device="cuda", dtype=torch.float32)
    E.gemm(1, ta, tb, M, Nn, K, A, A.shape[1], B, B.shape[1], C, Nn, out_f32=1)
    torch.cuda.synchronize()
    ref = (A.float().t() if ta else A.float()) @ (B.float() if tb else B.float().t())
    worst = max(worst, float((C - ref).norm() / ref.norm()))
print("WORST", worst)
"""


@pytest.mark.parametrize("env,shapes", [
    ({"AUDIOSSL_GEMM_RING": "24"}, [("NT", 300, 264, 128), ("NN", 300, 264, 128), ("TN", 296, 264, 128)]),
    ({"AUDIOSSL_GEMM_RING": "13"}, [("NT", 300, 264, 192), ("NN", 300, 264, 192), ("TN", 296, 264, 192)]),
    ({"AUDIOSSL_GEMM_T256": "1"}, [("NT", 1100, 520, 128), ("NN", 1100, 520, 128), ("TN", 1096, 520, 128)]),
    ({"AUDIOSSL_GEMM_W8": "1"}, [("NT", 300, 264, 128), ("NN", 300, 264, 128), ("TN", 296, 264, 128)]),
    ({"AUDIOSSL_GEMM_BK32": "2"}, [("NT", 2048, 4096, 96), ("NN", 2048, 4096, 96), ("TN", 2048, 4096, 96)]),
    # hand-scheduled 256 x 256 kernel: one K-tile, an odd and an even number of K-tiles, partial tiles in M and N
    ({"AUDIOSSL_GEMM_P8": "1"}, [("NT", 1100, 520, 64), ("NT", 1100, 520, 192), ("NT", 700, 264, 512), ("NN", 1100, 520, 192),
                                 ("NN", 700, 264, 512), ("TN", 1096, 520, 192), ("TN", 696, 264, 512), ("NT", 6144, 2048, 2048),
                                 ("NN", 1024, 2048, 1024), ("TN", 2048, 520, 1024)]),
    # software-pipelined 256 x 128 kernel (ring of three 48 KB stages): 1, 2, 3, 4 and 7 K-tiles, partial tiles in M and N, the heads' shapes
    ({"AUDIOSSL_GEMM_P6": "1", "AUDIOSSL_GEMM_P8": "0"},
     [("NT", 1100, 392, 64), ("NT", 1100, 392, 128), ("NT", 700, 264, 192), ("NT", 520, 136, 256), ("NN", 1100, 392, 192),
      ("NN", 700, 264, 448), ("NN", 300, 200, 320), ("TN", 1096, 392, 192), ("TN", 696, 264, 448), ("TN", 296, 136, 128), ("NT", 1024, 2048, 2048),
      ("NN", 1024, 2048, 2048), ("TN", 2048, 1024, 1024)]),
    # hand-scheduled 128 x 128 kernel (ring of five buffers): 1 ... 7 K-tiles, partial tiles, the heads' M = 512 shapes
    ({"AUDIOSSL_GEMM_P6": "5", "AUDIOSSL_GEMM_P8": "0"},
     [("NT", 600, 392, 64), ("NT", 600, 392, 128), ("NT", 300, 264, 192), ("NT", 520, 136, 256), ("NT", 130, 130, 320), ("NN", 600, 392, 384),
      ("NN", 300, 264, 448), ("TN", 600, 392, 192), ("TN", 296, 264, 448), ("NT", 512, 2048, 2048), ("NN", 512, 2048, 2048),
      ("NN", 6144, 512, 2048)]),
    # the software-pipelined form of the 256 x 256 tile (BK 32, ring of five) and of the 128 x 128 tile (BK 64, ring of five): not the
    # default for these tiles (gemm.hip, sp_mask), kept selectable - every layout, 1 ... 7 K-tiles, partial tiles
    ({"AUDIOSSL_GEMM_SP": "8", "AUDIOSSL_GEMM_P8": "1"},
     [("NT", 1100, 520, 64), ("NT", 1100, 520, 192), ("NT", 700, 264, 512), ("NN", 1100, 520, 192), ("NN", 700, 264, 512),
      ("TN", 1096, 520, 192), ("TN", 696, 264, 512), ("NT", 6144, 2048, 2048), ("TN", 2048, 520, 1024)]),
    ({"AUDIOSSL_GEMM_SP": "2", "AUDIOSSL_GEMM_P6": "5", "AUDIOSSL_GEMM_P8": "0"},
     [("NT", 600, 392, 64), ("NT", 600, 392, 128), ("NT", 300, 264, 192), ("NT", 130, 130, 320), ("NN", 600, 392, 384),
      ("NN", 300, 264, 448), ("TN", 600, 392, 192), ("TN", 296, 264, 448), ("NT", 512, 2048, 2048)]),
])
def test_gemm_tile_variants_in_subprocess(env, shapes):
    """The tile variants that the default dispatch does not pick for these shapes (ring of every layout, 256x256 and 256x128
    8-wave tiles, K-step 32 with a transposed A) stay correct: each is forced through its environment switch in a child
    process (the switches are read once per process) on shapes with partial tiles."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _VARIANT_SCRIPT.format(root=root, pkg=os.path.join(root, "audio-ssl_amd"), shapes=shapes)
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    worst = float(out.stdout.strip().split("WORST")[-1])
    assert worst < 1e-5, (env, worst)


def test_gemm_epilogue_large_grid(N):
    """>= 512 tiles: the default dispatch takes the K-step-32 kernel (three workgroups per CU, half-height epilogue tiles);
    bias + ReLU + keep mask + gate + fp32 residual on a 6144-row problem like the encoder's fc layers, partial tiles in M."""
    M, Nn, K = 6100, 2048, 96
    g = torch.Generator().manual_seed(7)
    A = torch.randn(M, K, generator=g).cuda().bfloat16()
    W = torch.randn(Nn, K, generator=g).cuda().bfloat16()
    bias = torch.randn(Nn, generator=g).cuda()
    keep = (torch.rand(M, Nn, generator=g) >= 0.3).to(torch.uint8).cuda()
    gate = torch.randn(M, Nn, generator=g).cuda().bfloat16()
    out = torch.empty(M, Nn, device="cuda", dtype=torch.bfloat16)
    N.call("gemm", 1, 0, 0, M, Nn, K, 1.0, A, K, W, K, out, Nn, bias, 1, keep, Nn, 1.0 / 0.7, gate, Nn, 0, 0, 1, None, 0)
    ref = torch.relu(A.float() @ W.float().T + bias) * keep.float() / 0.7 * (gate.float() > 0)
    resid = torch.randn(M, Nn, generator=g).cuda()
    out32 = torch.empty(M, Nn, device="cuda")
    N.call("gemm", 1, 0, 0, M, Nn, K, 0.5, A, K, W, K, out32, Nn, None, 0, None, 0, 1.0, None, 0, 1, 0, 1, resid, Nn)
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu(), ref.cpu()) < 4e-3
    assert rel_l2(out32.cpu(), (0.5 * (A.float() @ W.float().T) + resid).cpu()) < 1e-5


@pytest.mark.parametrize("sr_orig,n", [(44100, 3000), (22050, 2500), (8000, 1200), (48000, 4801)])
def test_resample_kaiser_best_vs_oracle(N, sr_orig, n):
    """Audio ingest: the windowed-sinc resampler of `librosa.core.load(path, sr=16000)` on the GPU against the
    sequential numpy restatement of resampy 0.2.2 - same taps, same fp64 products, same fp32 accumulation order, so the two
    agree to float32 rounding; plus the spectral sanity of a down-sampled tone and the loader entry point."""
    from src.dataset import ingest
    g = np.random.RandomState(sr_orig)
    t = np.arange(n) / sr_orig
    x = (0.4 * np.sin(2 * np.pi * 1000.0 * t) + 0.05 * g.randn(n)).astype(np.float32)
    got = ingest.resample(torch.from_numpy(x), sr_orig, 16000).cpu().numpy()
    want = FE.resample_kaiser_best(x, sr_orig, 16000)
    assert got.shape == want.shape == (int(np.ceil(n * 16000 / sr_orig)),)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    two = ingest.resample(torch.from_numpy(np.stack([x, x[::-1].copy()])), sr_orig, 16000).cpu().numpy()      # batched clips
    np.testing.assert_array_equal(two[0], got)
    k = int(round(1000.0 * len(got) / 16000))
    spec = np.abs(np.fft.rfft(got * np.hanning(len(got))))
    assert abs(int(spec.argmax()) - k) <= 1                                     # the 1 kHz tone stays at 1 kHz


def test_load_audio_resamples_on_the_gpu(N, tmp_path):
    from scipy.io import wavfile
    from src.dataset.upstream_dataset import load_audio
    sr = 22050
    x = (0.3 * np.sin(2 * np.pi * 440.0 * np.arange(sr) / sr)).astype(np.float32)
    p = str(tmp_path / "a.wav")
    wavfile.write(p, sr, (x * 32767).astype(np.int16))
    y = load_audio(p, 16000)
    assert y.dtype == np.float32 and y.shape == (16000,)
    ref = FE.resample_kaiser_best((x * 32767).astype(np.int16).astype(np.float32) / 32768.0, sr, 16000)
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6)
