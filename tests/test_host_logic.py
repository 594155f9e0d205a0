"""CPU: host-side logic of the product (tables, planner-driven window crops, module / state_dict compatibility)."""
import copy
import os
import random

import numpy as np
import pytest
import torch

from conftest import CFG_M, CFG_S
from oracle import frontend as FE


def test_filterbank_tables_equal_oracle():
    from src.utils.utils import pack_filterbank, slaney_mel_filterbank
    fb = slaney_mel_filterbank(16000, 1024, 64, 60, 7800)
    assert np.array_equal(fb, FE.mel_filterbank())
    st, pk = pack_filterbank(fb)
    assert pk.shape == (64, 45) and int((pk != 0).sum()) == 966
    for m in range(64):
        assert np.array_equal(fb[m, st[m]:st[m] + 45][pk[m] != 0], pk[m][pk[m] != 0])


def test_extract_window_matches_reference_golden(golden):
    from src.utils import extract_window
    for seed, n, first, nz0, last, after in golden("window")["rows"]:
        random.seed(int(seed))
        out = extract_window(torch.arange(int(n), dtype=torch.float32), data_size=1.0)
        assert len(out) == 16000 and float(out[0]) == first and float(out[-1]) == last
        assert random.random() == after


def test_state_dict_keys_match_reference_layout(cfg_s, cfg_m):
    from src.encoder import AudioNTT2020Task6
    from src.upstream.delores_m.upstream_expert import Upstream_Expert as M
    from src.upstream.delores_s.upstream_expert import Upstream_Expert as S
    from oracle import model as OM
    s = S(cfg_s, base_encoder=AudioNTT2020Task6)
    m = M(cfg_m, base_encoder=AudioNTT2020Task6, num_negatives=256)
    # the oracle modules are restated from the reference's constructors and pinned by its goldens
    assert list(s.state_dict()) == list(OM.DeloresSExpert(cfg_s).state_dict())
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == \
        [(k, tuple(v.shape)) for k, v in OM.DeloresMExpert(cfg_m, num_negatives=256).state_dict().items()]
    assert sum(p.numel() for p in s.parameters()) == 17912960
    assert repr(s.encoder.encoder) == "AudioNTT2020Task6" and s.encoder_q is s.encoder
    with pytest.raises(AssertionError):
        from src.augmentations import RandomResizeCrop
        RandomResizeCrop(time_scale=(0.6, 0.9))


def test_flat_group_keeps_views_and_state_dict(cfg_s):
    from src.encoder import AudioNTT2020Task6
    from src.flat import FlatGroup
    from src.upstream.delores_s.upstream_expert import Upstream_Expert as S
    s = S(cfg_s, base_encoder=AudioNTT2020Task6)
    before = {k: v.clone() for k, v in s.state_dict().items()}
    fg = FlatGroup(s.trainable_named())
    assert all(torch.equal(before[k], v) for k, v in s.state_dict().items())
    fg.data.mul_(2.0)
    assert torch.equal(s.state_dict()["p.projector.0.weight"], before["p.projector.0.weight"] * 2)
    s.load_state_dict(before)                       # copies into the views
    assert torch.equal(fg.data[:64 * 9].view(64, 1, 3, 3), before["encoder.encoder.features_1.0.weight"])
    assert all(o % 64 == 0 for o in fg.offsets)


def test_window_collate_consumes_python_stream_like_reference(cfg_s, tmp_path):
    """Crops + augmentation plan made at collate time == the reference's per-clip sequence of draws."""
    from src.augmentations import AugmentationModule
    from src.dataset.upstream_dataset import WindowCollate
    waves = [torch.arange(n, dtype=torch.float32) for n in (100, 16000, 20000, 48000, 16001)]
    np.random.seed(4); random.seed(4)
    tf = AugmentationModule(cfg_s, 10, max_batch=8)
    out, plan = WindowCollate(tf, 16000, 64)(waves)
    np.random.seed(4); random.seed(4)
    tf2 = AugmentationModule(cfg_s, 10, max_batch=8)
    tf2._ensure_ring(5)
    want = []
    ips = []
    for w in waves:                                                   # reference order: window, then both views
        want.append(FE.extract_window(w, data_size=1.0))
        ip, fp, _, _ = tf2.plan_py(1, 64, 101)
        ips.append(ip)
    assert torch.equal(out, torch.stack(want))
    assert np.array_equal(plan[0], np.concatenate(ips))
    assert tf.n_entries == tf2.n_entries == 10


def test_train_upstream_cli_and_config(tmp_path):
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_upstream_hip", os.path.join(root, "audio-ssl_amd", "train_upstream.py"))
    tu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tu)
    args = tu.get_args(["--input", "x.csv", "--upstream", "delores_s"])
    cfg = tu.load_config(args)
    assert cfg["pretrain"]["base_encoder"]["type"] == "AudioNTT2020Task6" and isinstance(cfg["pretrain"]["lambda_barlow"], float)
    assert tu.load_config(tu.get_args(["--input", "x.csv"]))["pretrain"]["loss_scale"] == "1/32"


def test_decar_v2_shards_are_distributed_sampler_shares():
    """`extras/decar-v2/main.py:102,158`: DistributedSampler(train_dataset) (shuffle, seed 0) + set_epoch(epoch) - the harness's
    `ShardedBatches` yields exactly that sampler's indices for every rank and epoch (drop_last batching on top), including the
    wrap-around padding when the data set is not a multiple of the world size."""
    import torch
    from torch.utils.data.distributed import DistributedSampler
    from src.upstream.decar_v2.main import ShardedBatches
    for n, world, batch in ((128, 2, 16), (101, 4, 5), (37, 8, 2)):
        for rank in range(world):
            sb = ShardedBatches(n, batch, lambda i: torch.tensor([float(i)]), rank, world)
            ds = DistributedSampler(range(n), world, rank, shuffle=True, seed=0)
            for epoch in (0, 1, 5):
                sb.set_epoch(epoch)
                ds.set_epoch(epoch)
                want = list(ds)
                assert sb.indices().tolist() == want
                got = [int(i) for ids, _ in sb for i in ids]
                assert got == want[:len(sb) * batch] and len(sb) == len(want) // batch
            assert sb.indices().tolist() != list(range(rank * (n // world), (rank + 1) * (n // world)))
        seen = set()
        for rank in range(world):
            seen |= set(ShardedBatches(n, batch, None, rank, world).indices().tolist())
        assert seen == set(range(n))


def test_flat_optimiser_state_schema_round_trip_and_foreign_checkpoints():
    """Every flat optimiser saves / restores its own state (momentum; Adam moments + step count) under a schema of its own, and
    declines - with a warning, not a KeyError - the per-parameter schema a Lightning / torch.optim checkpoint of the reference
    carries under `optimizer_states` (`resume_from_checkpoint`, train_upstream.py:54)."""
    import warnings
    import torch
    from src.flat import FlatGroup
    from src.optim import FlatState, HipAdamW, HipLARC, HipLARS, HipSGD
    def group():
        ps = [torch.nn.Parameter(torch.randn(5, 7)), torch.nn.Parameter(torch.randn(9))]
        return FlatGroup([("w", ps[0]), ("b", ps[1])]), ps
    for make, bufs in ((lambda f, p: HipSGD([f], p, 0.1), ("momentum",)), (lambda f, p: HipLARS([f], p, 0.2), ("momentum",)),
                       (lambda f, p: HipLARC([f], p, 0.3), ("momentum",)), (lambda f, p: HipAdamW([f], p, lr=1e-3), ("exp_avg", "exp_avg_sq"))):
        fg, ps = group()
        opt = make(fg, ps)
        assert isinstance(opt, FlatState)
        for i, b in enumerate(bufs):
            setattr(fg, b, torch.full((fg.numel,), 1.5 + i))
        opt.steps = 7
        if isinstance(opt, HipAdamW):
            opt.step_count = torch.tensor([7])
        opt.param_groups[0]["lr"] = 0.0625
        sd = opt.state_dict()
        assert sd["schema"] == FlatState.SCHEMA and sd["kind"] == type(opt).__name__
        fg2, ps2 = group()
        opt2 = make(fg2, ps2)
        assert opt2.load_state_dict(sd) is True
        assert opt2.steps == 7 and opt2.param_groups[0]["lr"] == 0.0625
        for i, b in enumerate(bufs):
            assert torch.equal(getattr(fg2, b), torch.full((fg2.numel,), 1.5 + i))
        if isinstance(opt, HipAdamW):
            assert int(opt2.step_count) == 7
        # torch's own schema (what the reference's Lightning checkpoint holds) and another flat optimiser's state are declined
        foreign = torch.optim.SGD(ps, lr=0.1, momentum=0.9).state_dict()
        fg3, ps3 = group()
        opt3 = make(fg3, ps3)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert opt3.load_state_dict(foreign) is False
            other = dict(sd, kind="SomethingElse")
            assert opt3.load_state_dict(other) is False
        assert len(w) == 2 and all(getattr(fg3, b, None) is None for b in bufs)


def test_partial_gradient_clear_table_and_flags():
    """HipSGD.step_tail(stored=...): the clear table lists exactly the runs of the stepped slice that are NOT gradients of the
    stored tensors (adjacent tensors merge, padding between tensors is left alone), and a partially cleared flat gradient is
    only accepted by a caller that stores into the stale tensors (`zero_grad(stores_ok=True)`): anyone else gets a full clear."""
    from src.flat import FlatGroup
    from src.optim import HipSGD
    ps = [torch.nn.Parameter(torch.randn(*sh)) for sh in ((64, 3), (128, 64), (128,), (128,), (128, 128), (100,))]
    names = ["enc.w", "h.0.weight", "h.1.weight", "h.1.bias", "h.3.weight", "h.4.bias"]
    fg = FlatGroup(list(zip(names, ps)))
    opt = HipSGD([fg], ps, 0.1)
    start = fg.offsets[1]
    table, nseg, longest = opt._zero_table(fg, start, ("h.0.weight", "h.3.weight"))
    runs = table.view(-1, 2).tolist()
    assert nseg == len(runs) == 2 and longest == 256
    assert runs[0] == [fg.offsets[2], 256]                          # h.1.weight + h.1.bias: adjacent, merged
    assert runs[1] == [fg.offsets[5], 100]
    fg.grad.fill_(1.0)
    fg.mark_fresh(partial=True)
    fg.zero_grad(stores_ok=True)
    assert float(fg.grad.min()) == 1.0                              # accepted: the stale tensors will be overwritten
    fg.mark_fresh(partial=True)
    fg.zero_grad()
    assert float(fg.grad.abs().max()) == 0.0                        # any other caller: full clear
    fg.grad.fill_(1.0)
    fg.mark_fresh()
    fg.zero_grad()
    assert float(fg.grad.min()) == 1.0 and fg._fresh_grad is False  # a full fused clear buys one skipped sweep, as before

