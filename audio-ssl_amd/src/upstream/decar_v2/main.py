"""DeepCluster-v2 (DECAR-v2) pre-training harness on MI355X - the trainer of `extras/decar-v2/main.py:57-196` (BASELINE
config 3), one process per GPU.

What the reference does per rank and what is done here:
  * model `AudioNTT2020(args, feat_dim, n_mels=64, d=2048)` wrapped in SyncBatchNorm + DistributedDataParallel (`main.py:80-84`)
    -> the HIP model of `decar_v2.model`, parameters in one flat buffer, gradients averaged over ranks with ONE all-reduce of
    the flat gradient per step (RCCL; 1/world folded into the optimiser launch) and SyncBatchNorm as a statistics exchange
    (`engine.SyncBN`): every train-mode BatchNorm - three BatchNorm2d of the encoder, the BatchNorm1d of the projection
    head - all-reduces its sums (54 tap moments / 2 x 64 / 2 x 2048 doubles forward, 2 x C sums backward) and normalises with the
    global batch; `sync_bn=False` keeps per-rank statistics.
  * SGD(momentum 0.9, wd 1e-6) inside apex LARC(trust 0.001, clip False) (`main.py:92-97, 111`) -> `HipLARC`.
  * `lr_schedule` = 10 warm-up epochs + cosine (`main.py:118-122`).  The reference builds it, hands it to `train` and never
    applies it (its optimiser runs at base_lr throughout); `apply_lr_schedule=True` applies it per iteration as the SwAV
    original does, the default False reproduces the reference.
  * per epoch: distributed spherical k-means over the memory bank (`cluster_memory`), then the prototype cross-entropy steps
    (`train`, `main.py:198-292`) with the prototype gradients dropped while it < freeze_prototypes_niters.
  * rank 0 writes `{'epoch', 'state_dict', 'optimizer'}` per epoch (`main.py:172-180`), every rank its memory bank.
Data: `DistributedSampler`'s share of the CSV per rank (seeded permutation per epoch, strided over the ranks, `ShardedBatches`);
every batch is (dataset indices, [view1, view2]) with the two views produced on the GPU by `UpstreamFrontEnd` (log-mel + RunningNorm + MixupBYOLA + RandomResizeCrop).
"""
import argparse
import os
import types

import numpy as np
import torch
import torch.distributed as dist

from src.optim import HipLARC, dcv2_lr_schedule
from src.upstream.decar_v2.model import AudioNTT2020
from src.upstream.decar_v2.train import DeepClusterState, cluster_epoch, init_memory, train_step


def default_args(**kw):
    """The knobs `extras/decar-v2/utils.py:get_upstream_parser` gives the trainer, with its defaults."""
    a = dict(feat_dim=512, nmb_prototypes=[1024], nmb_crops=[2], crops_for_assign=[0], freeze_prototypes_niters=1e10,
             base_lr=4.8, final_lr=0.0048, epochs=100, batch_size=512, nmb_kmeans_iters=10, d=2048, seed=31,
             apply_lr_schedule=False, save_dir=None, length_wave=0.95)
    a.update(kw)
    return types.SimpleNamespace(**a)


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


class ShardedBatches:
    """(indices, waveform batch) over this rank's share of the data set, drop_last; waveforms come from `get_wave(i)`.
    The share is `torch.utils.data.DistributedSampler(dataset)`'s (`extras/decar-v2/main.py:102`, shuffle=True, seed 0): a
    permutation of all items drawn from a generator seeded with seed + epoch (`sampler.set_epoch(epoch)`, `:158`), padded by
    wrapping to a multiple of the world size, rank r taking positions r, r + world, ...  - so every rank sees a fresh, unbiased
    subset every epoch instead of one fixed contiguous slice of the CSV in file order.  shuffle=False: strided, file order."""

    def __init__(self, n_items, batch, get_wave, rank, world, shuffle=True, seed=0):
        self.n, self.batch, self.get, self.rank, self.world = n_items, batch, get_wave, rank, world
        self.shuffle, self.seed, self.epoch = shuffle, seed, 0
        self.per = -(-n_items // world)                               # ceil, as DistributedSampler without drop_last

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def indices(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        total = self.per * self.world
        if total > len(idx):
            idx += (idx * -(-(total - len(idx)) // len(idx)))[:total - len(idx)]
        return np.asarray(idx[self.rank:total:self.world], dtype=np.int64)

    def __len__(self):
        return self.per // self.batch

    def __iter__(self):
        idx = self.indices()
        for b in range(len(self)):
            ids = idx[b * self.batch:(b + 1) * self.batch]
            yield torch.from_numpy(ids), torch.stack([self.get(int(i)) for i in ids])


def run(args, n_items, get_wave, front_end, device=None, max_iters=None, log=print, sync_bn=True):
    """Train for `args.epochs` epochs (or `max_iters` iterations).  front_end(waves[B, L] on the device) -> (view1, view2).
    -> (state, history of (iteration, loss))."""
    rank, world = _world()
    from src import engine as E
    E.set_sync_bn(E.SyncBN() if (sync_bn and world > 1) else None)
    device = device or torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    per_device_batch = args.batch_size // world                       # main.py:102
    model = AudioNTT2020(args, args.feat_dim, n_mels=64, d=args.d, nmb_prototypes=args.nmb_prototypes).to(device).train()
    batches = ShardedBatches(n_items, per_device_batch, get_wave, rank, world)
    state = DeepClusterState(model, HipLARC, args.feat_dim, len(batches) * per_device_batch, len(args.crops_for_assign),
                             lr=args.base_lr, momentum=0.9, weight_decay=1e-6, trust_coefficient=0.001, clip=False)
    schedule = dcv2_lr_schedule(args.base_lr, args.final_lr, max(args.epochs, 11), len(batches))

    def loader():
        for ids, waves in batches:
            v1, v2 = front_end(waves.to(device, non_blocking=True))
            yield ids, [v1, v2]
    init_memory(state, loader())
    history = []
    for epoch in range(args.epochs):
        batches.set_epoch(epoch)                                      # main.py:158
        assignments = cluster_epoch(state, n_items, tuple(args.nmb_prototypes), args.nmb_kmeans_iters, tuple(args.crops_for_assign))
        start = 0
        for i, (ids, inputs) in enumerate(loader()):
            it = len(batches) * epoch + i
            if args.apply_lr_schedule:
                state.optimizer.param_groups[0]["lr"] = float(schedule[min(it, len(schedule) - 1)])
            loss, start = train_step(state, ids, inputs, assignments, start, tuple(args.nmb_crops), tuple(args.crops_for_assign),
                                     args.freeze_prototypes_niters)
            history.append((it, float(loss)))
            if rank == 0 and it % 50 == 0:
                log(f"Epoch: [{epoch}][{it}]\tLoss {float(loss):.4f}\tLr: {state.optimizer.param_groups[0]['lr']:.4f}")
            if max_iters and it + 1 >= max_iters:
                break
        if args.save_dir:
            os.makedirs(os.path.join(args.save_dir, "checkpoints_deepcluster"), exist_ok=True)
            if rank == 0:
                torch.save({"epoch": epoch + 1, "state_dict": model.state_dict(),
                            "optimizer": {"momentum": state.flat.momentum, "lr": state.optimizer.param_groups[0]["lr"]}},
                           os.path.join(args.save_dir, "checkpoints_deepcluster", f"checkpoint_{epoch + 1}_.pth.tar"))
            torch.save({"local_memory_embeddings": state.local_memory_embeddings, "local_memory_index": state.local_memory_index},
                       os.path.join(args.save_dir, f"mb{rank}.pth"))
        if max_iters and len(history) >= max_iters:
            break
    return state, history


def main(argv=None):
    """python -m src.upstream.decar_v2.main --input train.csv [--epochs N ...]; launch under torch.distributed.run for N GPUs."""
    import pandas as pd
    from src.augmentations import AugmentationModule
    from src.dataset.upstream_dataset import UpstreamFrontEnd, load_audio
    from src.utils import extract_window
    ap = argparse.ArgumentParser(allow_abbrev=False)
    ap.add_argument("--input", required=True)
    ap.add_argument("--save_dir", default=None)
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--batch_size", type=int, default=512)
    ap.add_argument("--base_lr", type=float, default=4.8)
    ap.add_argument("--final_lr", type=float, default=0.0048)
    ap.add_argument("--apply_lr_schedule", action="store_true")
    ap.add_argument("--max_iters", type=int, default=None)
    a = ap.parse_args(argv)
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    files = list(pd.read_csv(a.input)["files"])
    args = default_args(epochs=a.epochs, batch_size=a.batch_size, base_lr=a.base_lr, final_lr=a.final_lr,
                        apply_lr_schedule=a.apply_lr_schedule, save_dir=a.save_dir)
    cfg = {"pretrain": {"normalization": "mean_var", "input": {"n_mels": 64, "length_wave": args.length_wave},
                        "augmentations": {"MixupBYOLA": {"ratio": 0.4, "log_mixup_exp": True},
                                          "RandomResizeCrop": {"virtual_crop_scale": [1.0, 1.5], "freq_crop_scale": [0.6, 1.5],
                                                               "time_crop_scale": [0.6, 1.5]}}}}
    tfms = AugmentationModule(cfg, len(files), max_batch=args.batch_size // world)
    front = UpstreamFrontEnd(cfg, tfms)
    get = lambda i: extract_window(torch.from_numpy(load_audio(files[i])), data_size=args.length_wave)
    return run(args, len(files), get, front, max_iters=a.max_iters)


if __name__ == "__main__":
    main()
