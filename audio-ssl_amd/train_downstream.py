#!/usr/bin/env python3
"""Downstream linear probe / fine-tuning harness for the MI355X path (BASELINE config 5).

Same CLI, config keys and flow as the reference's `train_downstream.py:19-204`: labelled CSVs -> `DownstreamEncoder`
(base encoder -> mean over time -> Linear) -> optional `freeze_encoder` + `load_pretrained_encoder` -> Adam(lr) on the
trainable parameters with CrossEntropyLoss, `train_one_epoch` then a rank-0 `eval` (loss, accuracy) per epoch, one JSON
stats line per epoch in `<exp_dir>/<task>/downstream_stats.txt`.
MI355X shape of it: one process per GPU (launch under torch.distributed.run for N > 1; the reference spawns), log-mels of a
whole batch in one `logmel_fwd` launch, encoder / head / loss through the HIP kernels, the trainable parameters in one flat
buffer updated by one `adamw` launch (weight_decay 0 = torch.optim.Adam), gradients averaged with one all-reduce.
As in the reference the model is in train() mode while training even when the encoder is frozen (BatchNorm uses batch
statistics and updates its running buffers, dropout is active) and in eval() mode for the evaluation pass.
"""
import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from src.dataset.downstream_dataset import DownstreamDataset, DownstreamFrontEnd  # noqa: E402
from src.downstream.downstream_encoder import DownstreamEncoder  # noqa: E402
from src.flat import FlatGroup  # noqa: E402
from src.optim import HipAdamW  # noqa: E402
from src.upstream.decar_v2.kmeans import prototype_cross_entropy as cross_entropy  # noqa: E402  (ce_rows kernel, mean over rows)
from src.utils import AverageMeter, Metric, freeze_encoder, load_pretrained_encoder  # noqa: E402


class ProbeTrainer:
    """Model + flat trainable parameters + Adam, with the reference's `train_one_epoch` / `eval`."""

    def __init__(self, model, lr, front_end=None):
        self.model = model
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.flat = FlatGroup(named)
        self.opt = HipAdamW([self.flat], [p for _, p in named], lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
        self.front = front_end or DownstreamFrontEnd()
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def step(self, x, target):
        """One optimisation step on a batch of log-mels [B, 1, n_mels, T] (device) and int64 targets -> loss tensor."""
        flat = self.flat
        flat.zero_grad()
        for p in flat.params:
            p.grad = None
        flat.attach_grads()
        out = self.model(x)
        loss = cross_entropy(out, target)
        loss.backward()
        if self.world > 1:
            dist.all_reduce(flat.grad)
            self.opt.grad_scale = 1.0 / self.world
        self.opt.step()
        return loss.detach()

    def train_one_epoch(self, loader, epoch, rank=0, log=print):
        losses, batch_time, end = AverageMeter(), AverageMeter(), time.time()
        self.model.train()
        for i, (waves, target) in enumerate(loader):
            x = self.front(waves)
            loss = self.step(x, torch.as_tensor(target).to(x.device))
            losses.update(loss, waves.size(0))
            batch_time.update(time.time() - end)
            end = time.time()
            if rank == 0 and log is not None:
                log(f"Epoch: [{epoch}][{i}/{len(loader)}]\tTime: {batch_time.val:.3f} ({batch_time.avg:.3f})\t"
                    f"Loss: {float(losses.val):.4f} ({float(losses.avg):.4f})")
        return dict(epoch=epoch, loss=losses)

    @torch.no_grad()
    def eval(self, loader, epoch=0):
        self.model.eval()
        losses, accuracy = AverageMeter(), Metric()
        for waves, targets in loader:
            x = self.front(waves)
            targets = torch.as_tensor(targets).to(x.device)
            outputs = self.model(x)
            loss = cross_entropy(outputs, targets)
            accuracy.update((torch.argmax(outputs, dim=1) == targets).cpu())
            losses.update(loss.cpu(), waves.size(0))
        return dict(epoch=epoch, loss=losses, accuracy=accuracy)


def sync_replicas(model, world=None):
    """What `DistributedDataParallel(model)` + `nn.SyncBatchNorm.convert_sync_batchnorm` give the reference
    (`train_downstream.py:80-85`): every parameter and buffer of the model becomes rank 0's (the random `final` Linear - and the
    encoder when no checkpoint is loaded - would otherwise differ per rank and only the gradients are all-reduced), and every
    train-mode BatchNorm of the encoder normalises with the statistics of the GLOBAL batch (`engine.SyncBN`).  No-op on one rank."""
    from src import engine as E
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    E.set_sync_bn(E.SyncBN() if world > 1 else None)
    if world > 1:
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0)
    return model


def main(args):
    cfg_path = args.config or os.path.join(HERE, "src", "downstream", "downstream_config.yaml")
    with open(cfg_path, "r") as f:
        config = yaml.load(f, Loader=yaml.SafeLoader)
    print(config)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise RuntimeError("train_downstream.py (MI355X path) needs a GPU; the HIP kernels have no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    exp_root = Path(args.exp_dir) / args.task
    exp_root.mkdir(parents=True, exist_ok=True)
    stats_file = open(exp_root / "downstream_stats.txt", "a", buffering=1) if rank == 0 else None
    assert config["run"]["batch_size"] % world == 0
    per_device_batch_size = config["run"]["batch_size"] // world
    train_dataset = DownstreamDataset(args, config, split="train")
    test_dataset = DownstreamDataset(args, config, split="test", labels_dict=train_dataset.labels_dict)
    sampler = torch.utils.data.distributed.DistributedSampler(train_dataset, world, rank, shuffle=True, seed=1)
    train_loader = torch.utils.data.DataLoader(train_dataset, batch_size=per_device_batch_size, pin_memory=True, sampler=sampler,
                                               num_workers=0)
    test_loader = torch.utils.data.DataLoader(test_dataset, batch_size=per_device_batch_size, pin_memory=True, num_workers=0)
    if args.encoder is not None:
        config["downstream"]["base_encoder"]["type"] = args.encoder
    base_encoder = getattr(importlib.import_module("src.encoder"), config["downstream"]["base_encoder"]["type"])
    model = DownstreamEncoder(config, args, base_encoder, no_of_classes=train_dataset.no_of_classes).cuda()
    if args.freeze:
        freeze_encoder(model)
    if args.checkpoint is not None:
        load_pretrained_encoder(model, args)
    sync_replicas(model, world)                                   # DDP's rank-0 broadcast + SyncBatchNorm (reference :80-85)
    trainer = ProbeTrainer(model, config["run"]["lr"])
    test_accuracy, history = [], []
    for epoch in range(config["run"]["epochs"]):
        sampler.set_epoch(epoch)
        train_stats = trainer.train_one_epoch(train_loader, epoch, rank, log=print if args.verbose else None)
        if rank == 0:
            ev = trainer.eval(test_loader, epoch)
            test_accuracy.append(float(ev["accuracy"].avg))
            stats = dict(epoch=epoch, Train_loss=float(train_stats["loss"].avg), Test_Loss=float(ev["loss"].avg),
                         Test_Accuracy=float(ev["accuracy"].avg), Best_Test_Acc=max(test_accuracy))
            print(stats)
            print(json.dumps(stats), file=stats_file)
            history.append(stats)
    if rank == 0:
        print("max valid accuracy : {}".format(max(test_accuracy)))
    if world > 1:
        from src import engine as E
        E.set_sync_bn(None)
    return trainer, history


def get_args(argv=None):
    parser = argparse.ArgumentParser(allow_abbrev=False)
    parser.add_argument("--task", type=str, default="test_task")
    parser.add_argument("--train_csv", type=str, required=True)
    parser.add_argument("--valid_csv", type=str, default=None)
    parser.add_argument("--test_csv", type=str, required=True)
    parser.add_argument("--checkpoint", type=str, help="path to pre-trained checkpoint", default=None)
    parser.add_argument("--encoder", type=str, default="AudioNTT2020Task6")
    parser.add_argument("--freeze", type=lambda s: str(s).lower() not in ("0", "false", "no"), default=True)
    parser.add_argument("--exp_dir", default="./exp", type=Path)
    parser.add_argument("--upstream", type=str, default="delores_m")
    parser.add_argument("-c", "--config", metavar="CONFIG_PATH", default=None)
    parser.add_argument("--verbose", action="store_true")
    return parser.parse_args(argv)


if __name__ == "__main__":
    main(get_args())
