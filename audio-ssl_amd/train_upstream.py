#!/usr/bin/env python3
"""Upstream pre-training harness for the MI355X path.

Same CLI and plugin lookup as the reference's `train_upstream.py:18-80`
(`--input CSV --upstream NAME [-c CONFIG] [--load_checkpoint CKPT]`; the expert is
`src.upstream.<name>.upstream_expert.Upstream_Expert`, the backbone `getattr(src.encoder, cfg.base_encoder.type)`).
pytorch-lightning is not in the image, so the loop is `HipTrainer`: one process per GPU (launch with
`python -m torch.distributed.run --nproc-per-node N train_upstream.py ...` for N > 1), RCCL through
torch.distributed, a single flat-buffer gradient all-reduce per step, Lightning-shaped checkpoints
(`{'state_dict', 'hyper_parameters', 'epoch', 'global_step'}`) that `load_pretrained_encoder` reads back.
"""
import argparse
import importlib
import os
import random
import sys
import time

import numpy as np
import pandas as pd
import torch
import torch.distributed as dist
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from src.augmentations import AugmentationModule  # noqa: E402
from src.dataset import BaselineDataModule  # noqa: E402
from src.optim import FlatState  # noqa: E402


class HipTrainer:
    def __init__(self, max_epochs=1, save_dir=None, resume_from_checkpoint=None, log_every=10, max_steps=None, use_graph=True):
        self.use_graph = use_graph
        self.max_epochs, self.save_dir, self.resume, self.log_every = max_epochs, save_dir, resume_from_checkpoint, log_every
        self.max_steps = max_steps
        self.rank = int(os.environ.get("RANK", 0))
        self.world = int(os.environ.get("WORLD_SIZE", 1))
        self.local_rank = int(os.environ.get("LOCAL_RANK", 0))
        self.global_step = 0
        self.epoch = 0
        self.history = []

    @property
    def world_size(self):
        return self.world

    use_ddp = property(lambda self: self.world > 1)
    current_epoch = property(lambda self: self.epoch)
    use_ddp2 = False

    def _init_dist(self):
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world)
        torch.cuda.set_device(self.local_rank)

    def fit(self, model, dm):
        self._init_dist()
        dev = torch.device("cuda", self.local_rank)
        model.to(dev).train()
        model.trainer = self
        self.datamodule = dm
        dm.setup("fit")
        sampler = None
        if self.world > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(dm.train_dataset, self.world, self.rank, shuffle=True)
        loader = dm.train_dataloader(sampler=sampler)
        opt = self.optimizer = model.configure_optimizers()
        # run.sync_batchnorm: `nn.SyncBatchNorm.convert_sync_batchnorm` of the extras trainers (extras/delores-s/main.py:79): every
        # train-mode BatchNorm - stem, conv blocks, projector - takes its statistics (forward and backward) over the global batch
        from src import engine as E
        sync = bool(getattr(model, "config", {}).get("run", {}).get("sync_batchnorm", False))
        E.set_sync_bn(E.SyncBN() if (sync and self.world > 1) else None)
        if self.resume:
            ck = torch.load(self.resume, map_location=dev, weights_only=True)
            model.load_state_dict(ck["state_dict"], strict=False)
            self.epoch, self.global_step = ck.get("epoch", 0), ck.get("global_step", 0)
            # Lightning's resume_from_checkpoint restores the optimiser too.  The flat optimisers keep their state in FlatGroup
            # buffers under their own schema (key `hip_optimizer_states`); a reference / Lightning checkpoint carries torch's
            # per-parameter schema under `optimizer_states`: its weights load, its optimiser state is declined with a warning
            states = ck.get("hip_optimizer_states") or ck.get("optimizer_states")
            if states and hasattr(opt, "load_state_dict"):
                if hasattr(model, "ensure_flat"):
                    model.ensure_flat()
                if isinstance(opt, FlatState):
                    opt.load_state_dict(states[0])
                else:
                    try:
                        opt.load_state_dict(states[0])
                    except (KeyError, ValueError) as e:
                        print(f"optimiser state of the checkpoint not restored ({e!r}); fresh optimiser state")
        best = float("inf")
        # one rank: the step is a single hipGraph replay; data-parallel: one graph per collective-free phase
        gstep = None
        if self.use_graph and hasattr(model, "graphed_step") and (self.world == 1 or model.graph_phases_supported()):
            gstep = model.graphed_step(opt)
        for epoch in range(self.epoch, self.max_epochs):
            self.epoch = epoch
            if sampler is not None:
                sampler.set_epoch(epoch)
            t0, clips = time.time(), 0
            batches = iter(loader)
            nxt = next(batches, None)
            ticket = None if nxt is None else dm.front_end.submit(nxt[0].to(dev, non_blocking=True), nxt[1])
            i = -1
            while nxt is not None:
                i += 1
                waves = nxt[0]
                img_1, img_2 = dm.front_end.collect(ticket)
                nxt = next(batches, None)               # the next batch's log-mel + augmentation runs under this step
                if nxt is not None:
                    ticket = dm.front_end.submit(nxt[0].to(dev, non_blocking=True), nxt[1])
                if gstep is not None:
                    loss = gstep(img_1, img_2)
                else:
                    opt.zero_grad()
                    loss = model.training_step((img_1, img_2), i)
                    loss.backward()
                    model.all_reduce_grads()
                    opt.step()
                self.global_step += 1
                clips += waves.shape[0] * self.world
                if self.rank == 0 and self.global_step % self.log_every == 0:
                    lv = float(loss)
                    self.history.append((self.global_step, lv))
                    print(f"epoch {epoch} step {self.global_step} train_loss {lv:.6f} clips/s {clips / (time.time() - t0):.1f}", flush=True)
                if self.max_steps and self.global_step >= self.max_steps:
                    break
            last = float(loss)
            if self.rank == 0 and self.save_dir and last < best:
                best = last
                self.save_checkpoint(os.path.join(self.save_dir, f"epoch={epoch}.ckpt"), model)
            if self.max_steps and self.global_step >= self.max_steps:
                break
        self.model, self.optimizer = model, opt
        return model

    def save_checkpoint(self, path, model=None):
        model = model or self.model
        if self.rank != 0:
            return
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        ck = model.checkpoint(self.epoch, self.global_step)
        ck["hyper_parameters"] = {k: v for k, v in ck["hyper_parameters"].items()}
        ck["hyper_parameters"]["config"] = model.config
        ck["hyper_parameters"]["base_encoder"] = model.config["pretrain"]["base_encoder"]["type"]
        opt = getattr(self, "optimizer", None)
        if opt is not None and hasattr(opt, "state_dict"):
            # a list with one entry per optimiser, like Lightning's `optimizer_states`, but under a key of its own: the flat
            # optimisers' schema is not torch's, and a Lightning loader must not mistake one for the other
            ck["hip_optimizer_states" if isinstance(opt, FlatState) else "optimizer_states"] = [opt.state_dict()]
        torch.save(ck, path)


def load_config(args):
    path = args.config or os.path.join(HERE, "src", "upstream", args.upstream, "config.yaml")
    with open(path, "r") as f:
        return yaml.load(f, Loader=yaml.SafeLoader)


def main(args):
    config = load_config(args)
    print(config)
    if args.seed is not None:                         # the extras trainers seed all three generators (main.py:59-64)
        np.random.seed(args.seed)
        random.seed(args.seed)
        torch.manual_seed(args.seed)
    tfms = AugmentationModule(config, len(pd.read_csv(args.input)), max_batch=config["run"]["batch_size"])
    dm = BaselineDataModule(config, args, tfms, data_csv=args.input, num_workers=config["run"]["num_dataloader_workers"],
                            batch_size=config["run"]["batch_size"])
    expert = getattr(importlib.import_module(f'src.upstream.{args.upstream}.upstream_expert'), 'Upstream_Expert')
    base_encoder = getattr(importlib.import_module('src.encoder'), config["pretrain"]["base_encoder"]["type"])
    model = expert(config, base_encoder=base_encoder, datamodule=dm)
    if not torch.cuda.is_available():
        raise RuntimeError("train_upstream.py (MI355X path) needs a GPU; the HIP kernels have no CPU fallback")
    trainer = HipTrainer(max_epochs=config["run"].get("max_epochs", 1), save_dir=config["run"]["save_path"] + '_chkp',
                         resume_from_checkpoint=args.load_checkpoint, max_steps=args.max_steps)
    trainer.fit(model, dm)
    trainer.save_checkpoint(args.final_checkpoint or os.path.join(config["run"]["save_path"], "final.ckpt"), model)
    return trainer


def get_args(argv=None):
    parser = argparse.ArgumentParser(allow_abbrev=False)
    parser.add_argument("--input", help="csv with a `files` column", type=str, required=True)
    parser.add_argument('--load_checkpoint', type=str, help='load checkpoint', default=None)
    parser.add_argument('-c', '--config', metavar='CONFIG_PATH', help='yaml config of the whole experiment', default=None)
    parser.add_argument('--upstream', type=str, help='define the type of upstream', default='delores_m')
    parser.add_argument('--seed', type=int, default=31)
    parser.add_argument('--max_steps', type=int, default=None)
    parser.add_argument('--final_checkpoint', type=str, default=None)
    return parser.parse_args(argv)


if __name__ == "__main__":
    main(get_args())
