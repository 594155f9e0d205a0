#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):

    cd /root/repo && python tests/golden/make_goldens.py

The reference's Python is imported from /root/reference (never copied); two
import-time shims stand in for packages the image lacks:
  * `pytorch_lightning` -> LightningModule = nn.Module + save_hyperparameters /
    log_dict / hparams (the training loop itself carries no arithmetic);
  * `librosa`           -> empty module (only `src.utils` imports it at module
    level; no librosa arithmetic is exercised, log-mel stays "parity unpinned").
Inputs and weights come from `oracle.fill` (integer-exact closed forms), so the
fixtures hold only small outputs.  Dropout masks are made explicit by swapping
the reference *instance's* `nn.Dropout` for a fixed-mask module (the reference's
own mask comes from torch's CPU generator, which no GPU can reproduce).
"""
import importlib
import importlib.util
import inspect
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import fill  # noqa: E402


# ----------------------------------------------------------------------- shims
def _install_shims():
    pl = types.ModuleType("pytorch_lightning")

    class _HP(dict):
        __getattr__ = dict.__getitem__

    class LightningModule(nn.Module):
        def save_hyperparameters(self):
            frame = inspect.currentframe().f_back
            sig = inspect.signature(type(self).__init__)
            loc = frame.f_locals
            self.hparams = _HP({k: loc[k] for k in sig.parameters
                                if k in loc and k not in ("self", "args", "kwargs")})

        def log_dict(self, *a, **k):
            pass

    class LightningDataModule:
        def __init__(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.LightningDataModule = LightningDataModule
    pl.Trainer = object
    cb = types.ModuleType("pytorch_lightning.callbacks")
    cb.ModelCheckpoint = object
    pl.callbacks = cb
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.callbacks"] = cb
    sys.modules["librosa"] = types.ModuleType("librosa")


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class FixedMaskDropout(nn.Module):
    """Pops one explicit keep-mask per call: y = x * mask / (1 - p)."""

    def __init__(self, p=0.3):
        super().__init__()
        self.p = p
        self.masks = []

    def forward(self, x):
        if not self.masks:
            return x
        m = self.masks.pop(0)
        return x * m.to(x.dtype) / (1.0 - self.p)


def drop_mask(shape, salt, p=0.3):
    return torch.from_numpy((fill.uniform01(shape, salt) >= p).astype(np.float32))


def views(B, T, salt):
    return torch.from_numpy(fill.normalish((B, 1, 64, T), salt))


CFG_S = {"pretrain": {"base_encoder": {"type": "AudioNTT2020Task6", "output_dim": 2048, "return_all_layers": False},
                      "projection_dim": 2048, "normalization": "mean_var", "lambda_barlow": 5e-5,
                      "input": {"type": "raw_wav", "sampling_rate": 16000, "length_wave": 1.0, "n_mels": 64},
                      "augmentations": {"MixupBYOLA": {"ratio": 0.4, "log_mixup_exp": True},
                                        "RandomResizeCrop": {"virtual_crop_scale": [1.0, 1.5],
                                                             "freq_crop_scale": [0.6, 1.5],
                                                             "time_crop_scale": [0.6, 1.5]}}}}
CFG_M = {"pretrain": dict(CFG_S["pretrain"], contrastive_dim=128, lambda_barlow=[5e-5, 5e-5, 5e-5], loss_scale="1/32",
                          base_encoder={"type": "AudioNTT2020Task6", "output_dim": 2048, "return_all_layers": True})}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def t2n(t):
    return t.detach().cpu().numpy()


def grad_digest(module):
    """name -> (l2 norm, first 8 flat values) for every parameter gradient."""
    names, norms, heads = [], [], []
    for n, p in module.named_parameters():
        if p.grad is None:
            continue
        names.append(n)
        norms.append(float(p.grad.norm()))
        h = p.grad.flatten()[:8]
        heads.append(np.pad(t2n(h), (0, 8 - h.numel())))
    return dict(names=np.array(names), norms=np.array(norms, np.float64), heads=np.stack(heads))


# ------------------------------------------------------------------ generators
def g1_window(U):
    rows = []
    for seed in (0, 1, 2):
        for n in (100, 16000, 20000, 48000):
            random.seed(seed)
            wav = torch.arange(n, dtype=torch.float32)
            out = U.extract_window(wav, data_size=1.0)
            nz = torch.nonzero(out)
            first = float(out[0])
            # start index recoverable from the ramp; left pad from first non-zero
            rows.append((seed, n, first, int(nz[0]) if len(nz) else -1, float(out[-1]), random.random()))
    save("window", rows=np.array(rows, np.float64))


def g2_runnorm(A):
    rn = A.RunningNorm(epoch_samples=2 * 3)      # freezes after 60 updates
    mus, sds, outs = [], [], []
    for c in range(64):
        x = torch.from_numpy(fill.normalish((1, 64, 101), 500 + c) * (2.0 + 0.1 * c) - 8.0 + 0.05 * c)
        y = rn(x)
        mus.append(float(rn.mean))
        sds.append(float(rn.std))
        if c in (0, 1, 2, 7, 59, 60, 63):
            outs.append(t2n(y)[0, ::8, ::10])
    save("runnorm", mu=np.array(mus, np.float64), sd=np.array(sds, np.float64), outs=np.stack(outs))


def g3_aug(A_pkg):
    for T in (101, 96):
        np.random.seed(31)
        random.seed(31)
        tf = A_pkg.AugmentationModule(CFG_S, 100)
        mix, rrc = tf.train_transform[0], tf.train_transform[1]
        # instrument draws without touching the source: wrap get_params
        trace = []
        orig = type(rrc).get_params

        def spy(*a, **k):
            r = orig(*a, **k)
            trace.append(tuple(int(v) for v in r))
            return r
        rrc.get_params = spy
        v1s, v2s = [], []
        n_clips = 20
        for c in range(n_clips):
            x = torch.from_numpy(fill.normalish((1, 64, T), 1000 + c) * 3.0 - 8.0)
            v1, v2 = tf(x)
            v1s.append(t2n(v1)[0])
            v2s.append(t2n(v2)[0])
        keep = [0, 1, 2, 7, 13, 19]
        digest = np.array([[v.sum(dtype=np.float64), np.abs(v).sum(dtype=np.float64)] for v in v1s + v2s])
        save(f"aug_T{T}", ijhw=np.array(trace, np.int32), keep=np.array(keep),
             v1=np.stack([v1s[c] for c in keep]), v2=np.stack([v2s[c] for c in keep]), digest=digest,
             np_state_after=np.random.random(), py_state_after=random.random())


def g3b_bank_wrap(A):
    """FIFO wrap-around: 2048-deep bank receives each clip twice."""
    np.random.seed(5)
    mix = A.MixupBYOLA(ratio=0.4, n_memory=8, log_mixup_exp=True)   # tiny bank, same code path
    outs = []
    for c in range(12):
        x = torch.from_numpy(fill.normalish((1, 8, 6), 2000 + c))
        outs.append(t2n(mix(x)))
        outs.append(t2n(mix(x)))
    save("mixup_wrap", outs=np.stack(outs))


def g4_specaug(S):
    recs, outs = [], []
    for seed in range(8):
        random.seed(1234 + seed)
        x = torch.from_numpy(fill.normalish((101, 64), 3000 + seed))
        y = S.freq_mask(x, F=30, num_masks=2)
        y = S.time_mask(y, T=40, num_masks=2)
        outs.append(t2n(y))
        recs.append(random.random())
        random.seed(1234 + seed)
        z = S.time_mask(S.freq_mask(x, F=30, num_masks=2, replace_with_zero=True), T=40, num_masks=2,
                        replace_with_zero=True)
        outs.append(t2n(z))
    save("specaug", outs=np.stack(outs), py_state_after=np.array(recs))


def g5_encoder(ENC):
    for T in (101, 96):
        enc = ENC.AudioNTT2020Task6(64, 2048, True)
        fill.fill_state_dict_(enc, seed=1)
        fmd = FixedMaskDropout(0.3)
        enc.fc[2] = fmd
        x = views(2, T, 4000 + T)
        Tp = T // 8
        out = {}
        # eval mode (running stats, no dropout)
        enc.eval()
        with torch.no_grad():
            e1, e2, e3, e = enc(x)
        out.update(eval_x1=t2n(e1), eval_x2=t2n(e2), eval_x3=t2n(e3), eval_x=t2n(e))
        # train mode with explicit dropout mask
        enc.train()
        fmd.masks = [drop_mask((2, Tp, 2048), 4100 + T)]
        a1, a2, a3, a = enc(x)
        out.update(x1=t2n(a1), x2=t2n(a2), x3=t2n(a3), x=t2n(a))
        r = [torch.from_numpy(fill.uniform(tuple(v.shape), 4200 + i)) for i, v in enumerate((a1, a2, a3, a))]
        loss = sum((v * w).sum() for v, w in zip((a1, a2, a3, a), r))
        loss.backward()
        out.update(loss=float(loss))
        gd = grad_digest(enc)
        out.update(g_names=gd["names"], g_norms=gd["norms"], g_heads=gd["heads"])
        out.update(bn1_rm=t2n(enc.features_1[1].running_mean), bn1_rv=t2n(enc.features_1[1].running_var),
                   bn3_rm=t2n(enc.features_3[1].running_mean), bn3_rv=t2n(enc.features_3[1].running_var))
        save(f"encoder_T{T}", **out)


def g6_barlow(XS):
    out = {}
    for in_dim in (2048, 1024, 512):
        p = XS.Projection(in_dim, 5e-5)
        fill.fill_state_dict_(p, seed=in_dim)
        p.train()
        B = 8
        y1 = torch.from_numpy(fill.uniform((B, in_dim), 5000 + in_dim, 0.0, 2.0)).requires_grad_()
        y2 = torch.from_numpy(fill.uniform((B, in_dim), 5001 + in_dim, 0.0, 2.0)).requires_grad_()
        loss = p(y1, y2)
        loss.backward()
        gd = grad_digest(p)
        out[f"loss_{in_dim}"] = float(loss)
        out[f"dy1_{in_dim}"] = t2n(y1.grad)
        out[f"dy2_{in_dim}"] = t2n(y2.grad)
        out[f"gn_{in_dim}"] = gd["norms"]
        out[f"gh_{in_dim}"] = gd["heads"]
        out[f"names_{in_dim}"] = gd["names"]
        out[f"bn_rm_{in_dim}"] = t2n(p.bn.running_mean)[:64]
        out[f"bn_rv_{in_dim}"] = t2n(p.bn.running_var)[:64]
    save("barlow", **out)


class _Trainer:
    use_ddp = False
    use_ddp2 = False

    class datamodule:
        name = "none"


def _closed_queue(emb_dim, K):
    q = torch.from_numpy(fill.normalish((emb_dim, K), 777))
    return nn.functional.normalize(q, dim=0)


def g7_g9_steps(XS, XM, ENC):
    # ---- delores_s: 3 SGD steps, B=8
    B, T, Tp = 8, 101, 12
    torch.manual_seed(0)
    ex = XS.Upstream_Expert(CFG_S, base_encoder=ENC.AudioNTT2020Task6)
    fill.fill_state_dict_(ex, seed=2)
    ex.trainer = _Trainer()
    fmd = FixedMaskDropout(0.3)
    ex.encoder.encoder.fc[2] = fmd
    ex.train()
    opt = ex.configure_optimizers()
    losses = []
    for s in range(3):
        fmd.masks = [drop_mask((B, Tp, 2048), 6100 + 2 * s), drop_mask((B, Tp, 2048), 6101 + 2 * s)]
        opt.zero_grad()
        loss = ex.training_step((views(B, T, 6000 + 2 * s), views(B, T, 6001 + 2 * s)), s)
        loss.backward()
        if s == 0:
            gd0 = grad_digest(ex)
        opt.step()
        losses.append(float(loss))
    sd = ex.state_dict()
    save("step_delores_s", losses=np.array(losses, np.float64), g_names=gd0["names"], g_norms=gd0["norms"],
         g_heads=gd0["heads"],
         w_conv1=t2n(sd["encoder.encoder.features_1.0.weight"]).ravel(),
         w_fc2_head=t2n(sd["encoder.encoder.fc.3.weight"]).ravel()[:256],
         w_p0_head=t2n(sd["p.projector.0.weight"]).ravel()[:256],
         bn_rm=t2n(sd["p.bn.running_mean"])[:64])

    # ---- delores_m: 3 SGD steps, B=8, queue 1024
    torch.manual_seed(0)
    K = 1024
    em = XM.Upstream_Expert(CFG_M, base_encoder=ENC.AudioNTT2020Task6, num_negatives=K)
    fill.fill_state_dict_(em, seed=3)
    for pq, pk in zip(em.encoder_q.parameters(), em.encoder_k.parameters()):
        pk.data.copy_(pq.data)
    em.queue.copy_(_closed_queue(128, K))
    em.trainer = _Trainer()
    fq, fk = FixedMaskDropout(0.3), FixedMaskDropout(0.3)
    em.encoder_q.encoder.fc[2] = fq
    em.encoder_k.encoder.fc[2] = fk
    em.train()
    opt = em.configure_optimizers()
    losses, ptrs = [], []
    logits0 = None
    for s in range(3):
        fq.masks = [drop_mask((B, Tp, 2048), 7100 + 2 * s)]
        fk.masks = [drop_mask((B, Tp, 2048), 7101 + 2 * s)]
        a, b = views(B, T, 7000 + 2 * s), views(B, T, 7001 + 2 * s)
        opt.zero_grad()
        if s == 0:
            # same call the training_step makes, captured for the MoCo-head fixture
            hook = {}
            orig_ce = torch.nn.functional.cross_entropy

            def spy_ce(inp, tgt, *aa, **kk):
                hook["logits"] = inp.detach().clone()
                return orig_ce(inp, tgt, *aa, **kk)
            XM.F.cross_entropy = spy_ce
            loss = em.training_step((a, b), s)
            XM.F.cross_entropy = orig_ce
            logits0 = hook["logits"]
        else:
            loss = em.training_step((a, b), s)
        loss.backward()
        if s == 0:
            gd0 = grad_digest(em)
        opt.step()
        losses.append(float(loss))
        ptrs.append(int(em.queue_ptr))
    sd = em.state_dict()
    save("step_delores_m", losses=np.array(losses, np.float64), ptrs=np.array(ptrs), g_names=gd0["names"],
         g_norms=gd0["norms"], g_heads=gd0["heads"], logits_row0=t2n(logits0[0]),
         ce0=float(torch.nn.functional.cross_entropy(logits0, torch.zeros(B, dtype=torch.long))),
         queue_cols=t2n(sd["queue"][:, :24]),
         wq_conv1=t2n(sd["encoder_q.encoder.features_1.0.weight"]).ravel(),
         wk_conv1=t2n(sd["encoder_k.encoder.features_1.0.weight"]).ravel(),
         wk_fc=t2n(sd["encoder_k.fc.weight"]).ravel()[:256])


def g8_contrastive(CL):
    out = {}
    for B, tau in ((8, 0.5), (16, 0.07)):
        zi = nn.functional.normalize(torch.from_numpy(fill.normalish((B, 128), 8000 + B)), dim=1).requires_grad_()
        zj = nn.functional.normalize(torch.from_numpy(fill.normalish((B, 128), 8001 + B)), dim=1).requires_grad_()
        loss = CL.InstanceLoss(B, tau, "cpu")(zi, zj)
        loss.backward()
        out[f"nt_{B}"] = float(loss)
        out[f"nt_dzi_{B}"] = t2n(zi.grad)
        out[f"nt_dzj_{B}"] = t2n(zj.grad)
    K, B = 16, 24
    ci = torch.softmax(torch.from_numpy(fill.normalish((B, K), 8100)), dim=1).requires_grad_()
    cj = torch.softmax(torch.from_numpy(fill.normalish((B, K), 8101)), dim=1).requires_grad_()
    loss = CL.ClusterLoss(K, 1.0, "cpu")(ci, cj)
    loss.backward()
    out["cl"] = float(loss)
    out["cl_dci"] = t2n(ci.grad)
    save("contrastive", **out)


def g10_lars(MP):
    ps = [nn.Parameter(torch.from_numpy(fill.uniform(s, 9000 + i))) for i, s in enumerate([(16, 8), (16,), (4, 4, 3, 3)])]
    opt = MP.LARS(ps, lr=0.2, weight_decay=1.5e-6, momentum=0.9, eta=0.001,
                  weight_decay_filter=True, lars_adaptation_filter=True)
    for s in range(3):
        for i, p in enumerate(ps):
            p.grad = torch.from_numpy(fill.uniform(tuple(p.shape), 9100 + 10 * s + i))
        opt.step()
    save("lars", p0=t2n(ps[0]), p1=t2n(ps[1]), p2=t2n(ps[2]))


def g13_schedules(MP):
    """LR schedules of the extras trainers, from the reference's own functions: `adjust_learning_rate`
    (extras/delores-s/multi_proc.py:45-57; it uses `math` without importing it - the name is supplied here) and
    `cosine_scheduler` (extras/decar-v2/multi_proc.py:61-72)."""
    import math
    import types
    MP.math = math
    steps_per_epoch, epochs, bs = 7, 30, 512
    opt = types.SimpleNamespace(param_groups=[{"lr": 0.0}, {"lr": 0.0}])
    args = types.SimpleNamespace(epochs=epochs, batch_size=bs)
    loader = [None] * steps_per_epoch
    lw, lb = [], []
    for step in range(epochs * steps_per_epoch):
        MP.adjust_learning_rate(args, opt, loader, step)
        lw.append(opt.param_groups[0]["lr"])
        lb.append(opt.param_groups[1]["lr"])
    MP2 = _load_by_path("ref_multiproc_dc", os.path.join(REF, "extras/decar-v2/multi_proc.py"))
    cs = MP2.cosine_scheduler(4.8, 0.0048, 25, 9, warmup_epochs=10, start_warmup_value=0.3)
    cs0 = MP2.cosine_scheduler(1.0, 0.1, 6, 5)
    save("schedules", lars_lr_weights=np.array(lw), lars_lr_biases=np.array(lb), cosine_warm=cs, cosine_plain=cs0)


def g14_kmix(A):
    """The reference's `Kmix` (src/augmentations/augmentations.py:119-189) run call by call on closed-form inputs with a
    closed-form centroid tensor: 170 calls, memory bank capped at 140 entries, so the random-index branch (< 128 entries), the
    cluster-guided branch (`get_index`) and the FIFO trimming all occur.  Recorded: the chosen bank index of every call (by
    wrapping get_index / np.random.randint from outside - the class itself is untouched), a few outputs, the stream position."""
    import tempfile
    F, T, K, calls = 64, 24, 12, 170
    cent = torch.from_numpy(fill.normalish((K, F), 4242)) + 0.3
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "centroids.pt")
        torch.save(cent, path)
        km = A.Kmix(ratio=0.4, n_memory=140, log_mixup_exp=True, top_k=16, centroid_path=path)
    np.random.seed(77)
    chosen, outs = [], {}
    orig_randint = np.random.randint
    picks = []

    def spy_randint(*a, **k):
        v = orig_randint(*a, **k)
        picks.append((int(a[0]), int(v)))
        return v
    np.random.randint = spy_randint
    try:
        for c in range(calls):
            x = torch.from_numpy(fill.normalish((1, F, T), 5000 + c)) * (0.5 + 0.1 * (c % 7)) + 0.2 * (c % 5)
            n_before = len(picks)
            y = km(x)
            chosen.append(picks[-1] if len(picks) > n_before else (0, -1))       # (randint argument = len(l) or len(bank), value)
            if c in (0, 1, 5, 127, 128, 129, 141, 169):
                outs[f"y{c}"] = t2n(y)
    finally:
        np.random.randint = orig_randint
    tail = np.random.random()
    save("kmix", centroids=t2n(cent), picks=np.array(chosen, np.int64), tail=np.float64(tail), **outs)


CFG_SL = {"pretrain": dict(CFG_S["pretrain"], instance_contrastive_dim=128, cluster_contrastive_dim=128)}


def g11_slicer(CL, ENC):
    """SLICER expert (`src/upstream/slicer/upstream_expert.py`): the plugin imports `ClusterLoss` from `src.utils`, where
    the reference forgot to put it; the class of `extras/slicer/contrastive_loss.py` is attached to the
    imported `src.utils` module object before the plugin is imported.  Two steps, B=8, queue 256."""
    import src.utils as ref_utils
    ref_utils.ClusterLoss = CL.ClusterLoss
    XSL = importlib.import_module("src.upstream.slicer.upstream_expert")
    B, T, Tp, K = 8, 101, 12, 256
    torch.manual_seed(0)
    ex = XSL.Upstream_Expert(CFG_SL, base_encoder=ENC.AudioNTT2020Task6, num_negatives=K)
    fill.fill_state_dict_(ex, seed=5)
    for pq, pk in zip(ex.encoder_q.parameters(), ex.encoder_k.parameters()):
        pk.data.copy_(pq.data)
    ex.queue.copy_(_closed_queue(128, K))
    ex.trainer = _Trainer()
    fq, fk = FixedMaskDropout(0.3), FixedMaskDropout(0.3)
    ex.encoder_q.encoder.fc[2] = fq
    ex.encoder_k.encoder.fc[2] = fk
    ex.train()
    opt = ex.configure_optimizers()
    logged = []
    ex.log_dict = lambda d, *a, **k: logged.append(d)
    returned, combine, sym, cluster, ptrs = [], [], [], [], []
    for s in range(2):
        fq.masks = [drop_mask((B, Tp, 2048), 8100 + 4 * s), drop_mask((B, Tp, 2048), 8102 + 4 * s)]     # q: view 1, view 2
        fk.masks = [drop_mask((B, Tp, 2048), 8101 + 4 * s), drop_mask((B, Tp, 2048), 8103 + 4 * s)]     # k: view 2, view 1
        a, b = views(B, T, 8000 + 2 * s), views(B, T, 8001 + 2 * s)
        opt.zero_grad()
        loss = ex.training_step((a, b), s)
        log = logged[-1]
        log["train_loss"].backward()            # gradient of the logged total (the shipped step returns the first CE only)
        if s == 0:
            gd0 = grad_digest(ex)
        opt.step()
        returned.append(float(loss))
        combine.append(float(log["train_loss"]))
        sym.append(float(log["sym_instance_loss"]))
        cluster.append(float(log["train_loss_cluster"]))
        ptrs.append(int(ex.queue_ptr))
    sd = ex.state_dict()
    save("step_slicer", returned=np.array(returned), combine=np.array(combine), sym=np.array(sym), cluster=np.array(cluster),
         ptrs=np.array(ptrs), g_names=gd0["names"], g_norms=gd0["norms"], g_heads=gd0["heads"],
         queue_cols=t2n(sd["queue"][:, :32]),
         wq_conv1=t2n(sd["encoder_q.encoder.features_1.0.weight"]).ravel(),
         wq_cluster=t2n(sd["encoder_q.cluster_projector.2.weight"]).ravel()[:256],
         wk_inst=t2n(sd["encoder_k.instance_projector.weight"]).ravel()[:256])


def g12_decar(ENC):
    """DeepCluster-v2 model + loss (`extras/decar-v2/models_delores.py:79-122`, `main.py:205-233`): one forward/backward,
    B=8.  The module imports `utils` from its own directory, which imports tensorflow and librosa at module level -
    both absent here and unused by the classes exercised - so empty stand-in modules are registered for the import."""
    sys.modules.setdefault("tensorflow", types.ModuleType("tensorflow"))
    d = os.path.join(REF, "extras/decar-v2")
    sys.path.insert(0, d)
    saved_utils = sys.modules.pop("utils", None)
    try:
        MD = _load_by_path("ref_decar_models", os.path.join(d, "models_delores.py"))
    finally:
        sys.path.remove(d)
        sys.modules.pop("utils", None)
        if saved_utils is not None:
            sys.modules["utils"] = saved_utils
    args = types.SimpleNamespace(nmb_prototypes=[1024], crops_for_assign=[0])
    B, T, Tp = 8, 101, 12
    torch.manual_seed(0)
    m = MD.AudioNTT2020(args, 512, n_mels=64, d=2048, nmb_prototypes=[1024])
    fill.fill_state_dict_(m, seed=6)
    fmd = FixedMaskDropout(0.3)
    m.fc[2] = fmd
    m.train()
    fmd.masks = [drop_mask((B, Tp, 2048), 8500), drop_mask((B, Tp, 2048), 8501)]
    emb, output = m([views(B, T, 8400), views(B, T, 8401)])
    targets = torch.tensor([(37 * i + 5) % 1024 for i in range(B)])
    targets[3] = -100                                            # an unseen clip (ignore_index)
    ce = nn.CrossEntropyLoss(ignore_index=-100)
    loss = ce(output[0] / 1.0, targets)
    loss.backward()
    gd = grad_digest(m)
    sd = m.state_dict()
    save("decar_v2_model", loss=float(loss), emb_head=t2n(emb[:, :16]), emb_norm=float(emb.norm()),
         scores_head=t2n(output[0][:, :16]), scores_norm=float(output[0].norm()), targets=t2n(targets),
         g_names=gd["names"], g_norms=gd["norms"], g_heads=gd["heads"],
         bn_rm=t2n(sd["projection_head.1.running_mean"])[:64], bn_rv=t2n(sd["projection_head.1.running_var"])[:64],
         state_keys=np.array(list(sd.keys())))


def g15_mvit_block():
    """The reference's own `MultiScaleBlock` (extras/mast_new/mast/mvit/models/attention.py:304-393) in three configurations
    (oracle/mvit.py GOLDEN_CONFIGS): outputs + gradient digests.  Only attention.py and common.py of the vendored mvit package
    are executed: the package `__init__`s pull in fvcore / iopath / simplejson (absent here), so `mvit`, `mvit.utils`,
    `mvit.models` are registered as empty namespace modules and `mvit.utils.logging` as a stand-in for its `get_logger`
    (the block does no logging arithmetic)."""
    import logging
    from functools import partial
    from oracle import mvit as OM
    root = os.path.join(REF, "extras/mast_new/mast")
    for name, path in (("mvit", "mvit"), ("mvit.utils", "mvit/utils"), ("mvit.models", "mvit/models")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(root, path)]
        sys.modules[name] = m
    lg = types.ModuleType("mvit.utils.logging")
    lg.get_logger = logging.getLogger
    sys.modules["mvit.utils.logging"] = lg
    ATT = importlib.import_module("mvit.models.attention")
    out = {}
    for cname, cfg in OM.GOLDEN_CONFIGS.items():
        hw = cfg["hw"]
        blk = ATT.MultiScaleBlock(
            dim=cfg["dim"], dim_out=cfg["dim_out"], num_heads=cfg["heads"], input_size=list(hw), mlp_ratio=4.0, qkv_bias=True,
            drop_path=0.0, norm_layer=partial(nn.LayerNorm, eps=1e-6),
            kernel_q=(3, 3) if cfg.get("stride_q") else (), kernel_kv=(3, 3) if cfg.get("stride_kv") else (),
            stride_q=cfg.get("stride_q", ()), stride_kv=cfg.get("stride_kv", ()), mode="conv", has_cls_embed=False,
            pool_first=False, rel_pos_spatial=cfg["rel_pos"], rel_pos_zero_init=False,
            residual_pooling=cfg["residual_pooling"], dim_mul_in_att=cfg.get("dim_mul_in_att", False))
        fill.fill_state_dict_(blk, seed=150 + len(cname))
        with torch.no_grad():
            for n, prm in blk.named_parameters():
                if "rel_pos" in n:                                   # embedding tables: small values like their trunc-normal init
                    prm.copy_(torch.from_numpy(fill.uniform(tuple(prm.shape), fill.salt_of(cname + n), -0.2, 0.2)))
        blk.train()
        B, L = 2, hw[0] * hw[1]
        x = torch.from_numpy(fill.normalish((B, L, cfg["dim"]), 1500 + len(cname))).requires_grad_(True)
        y, hw_out = blk(x, list(hw))
        gy = torch.from_numpy(fill.uniform(tuple(y.shape), 1501 + len(cname)))
        (y * gy).sum().backward()
        gd = grad_digest(blk)
        out.update({f"{cname}.y": t2n(y), f"{cname}.hw_out": np.array(hw_out), f"{cname}.dx": t2n(x.grad),
                    f"{cname}.g_names": gd["names"], f"{cname}.g_norms": gd["norms"], f"{cname}.g_heads": gd["heads"],
                    f"{cname}.state_keys": np.array(list(blk.state_dict().keys()))})
    save("mvit_block", **out)


def g16_kmeans():
    """The reference's own `cluster_memory` (extras/decar-v2/utils.py:276-346) on the CPU.  It is plain torch + scipy, but written
    for a GPU rank: it calls `.cuda(non_blocking=True)` on fresh tensors and uses the process group.  Here: `torch.Tensor.cuda`
    is a no-op for the duration of the call, the group is the 1-process gloo group the reference's own trainers create
    (extras/delores-m/train_moco.py:18), `tensorflow` / `librosa` (imported at module level by utils.py, unused by this
    function) are empty stand-in modules, and `nn.functional.normalize` is wrapped to record the centroids after every M step.
    Case A: N = 4096 clips in a 4160-item data set, d = 64, K = 32, 10 iterations, a noisy mixture around 40 directions.
    Case B: an empty-cluster case - 512 rows that are copies of 20 distinct directions, K = 32: duplicate seeds lose every
    arg-max tie and keep their centroid (`centroids[mask] = ...`)."""
    import socket
    import torch.distributed as dist
    sys.modules.setdefault("tensorflow", types.ModuleType("tensorflow"))
    sys.modules.setdefault("librosa", types.ModuleType("librosa"))
    U = _load_by_path("ref_decar_utils", os.path.join(REF, "extras/decar-v2/utils.py"))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    real_cuda, real_nn = torch.Tensor.cuda, U.nn
    trace = []

    def rec_normalize(x, *a, **k):
        y = real_nn.functional.normalize(x, *a, **k)
        trace.append(y.clone())
        return y
    out = {}
    try:
        torch.Tensor.cuda = lambda self, *a, **k: self
        U.nn = types.SimpleNamespace(functional=types.SimpleNamespace(normalize=rec_normalize))
        for case, (N, size_dataset, d, K, iters, seed) in {"a": (4096, 4160, 64, 32, 10, 1234), "b": (512, 512, 16, 32, 4, 99)}.items():
            if case == "a":
                centres = fill.normalish((40, d), 1601)
                x = centres[np.arange(N) % 40] + 0.45 * fill.normalish((N, d), 1602)
            else:
                x = fill.normalish((20, d), 1603)[(7 * np.arange(N)) % 20]
            mem = torch.nn.functional.normalize(torch.from_numpy(x.astype(np.float32)), dim=1)
            index = torch.from_numpy(((np.arange(N) * 37 + 11) % size_dataset).astype(np.int64))       # 37 is coprime to both sizes
            assert len(set(index.tolist())) == N
            proto = nn.Linear(d, K, bias=False)
            model = types.SimpleNamespace(module=types.SimpleNamespace(prototypes=types.SimpleNamespace(prototypes0=proto)))
            args = types.SimpleNamespace(nmb_prototypes=[K], feat_dim=d, rank=0, world_size=1, crops_for_assign=[0])
            torch.manual_seed(seed)
            seed_idx = torch.randperm(N)[:K]                            # the function's first draw from the global generator
            torch.manual_seed(seed)
            del trace[:]
            assign = U.cluster_memory(args, model, index, mem[None].clone(), size_dataset, nmb_kmeans_iters=iters)
            assert len(trace) == iters and torch.equal(trace[-1], proto.weight.detach())
            out.update({f"{case}.mem": t2n(mem), f"{case}.index": t2n(index), f"{case}.seed_idx": t2n(seed_idx),
                        f"{case}.centroids": np.stack([t2n(c) for c in trace]), f"{case}.assignments": t2n(assign[0]),
                        f"{case}.dims": np.array([N, size_dataset, d, K, iters, seed])})
    finally:
        torch.Tensor.cuda, U.nn = real_cuda, real_nn
        dist.destroy_process_group()
    counts_b = np.bincount(out["b.assignments"][out["b.assignments"] >= 0], minlength=32)
    assert (counts_b == 0).sum() >= 12, counts_b                        # case B really has empty clusters
    save("kmeans_ref", **out)


def main():
    _install_shims()
    sys.path.insert(0, REF)
    A_pkg = importlib.import_module("src.augmentations")
    A = importlib.import_module("src.augmentations.augmentations")
    U = importlib.import_module("src.utils.utils")
    ENC = _load_by_path("ref_audiontt", os.path.join(REF, "src/encoder/audiontt.py"))
    XS = importlib.import_module("src.upstream.delores_s.upstream_expert")
    XM = importlib.import_module("src.upstream.delores_m.upstream_expert")
    S = _load_by_path("ref_specaug", os.path.join(REF, "extras/delores-s/specaugment.py"))
    CL = _load_by_path("ref_closs", os.path.join(REF, "extras/slicer/contrastive_loss.py"))
    MP = _load_by_path("ref_multiproc", os.path.join(REF, "extras/delores-s/multi_proc.py"))
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "slicer":          # regenerate one fixture without touching the others
        g11_slicer(CL, ENC)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "decar":
        g12_decar(ENC)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "schedules":
        g13_schedules(MP)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "kmix":
        g14_kmix(A)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "mvit":
        g15_mvit_block()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "kmeans":
        g16_kmeans()
        return
    g1_window(U)
    g2_runnorm(A)
    g3_aug(A_pkg)
    g3b_bank_wrap(A)
    g4_specaug(S)
    g5_encoder(ENC)
    g6_barlow(XS)
    g8_contrastive(CL)
    g10_lars(MP)
    g7_g9_steps(XS, XM, ENC)
    g11_slicer(CL, ENC)
    g12_decar(ENC)
    g13_schedules(MP)
    g14_kmix(A)
    g15_mvit_block()
    g16_kmeans()


if __name__ == "__main__":
    main()
