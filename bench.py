#!/usr/bin/env python3
"""Headline benchmark: upstream clips/s of the DeLoRes-M pre-training step on MI355X (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one synthetic batch already resident in HBM:
  waveforms [B,16000] -> log-mel -> running norm + two augmented views -> q/k encoders, MoCo InfoNCE (65,536-key
  queue), three Barlow heads, full backward -> [all-reduce] -> SGD(momentum) step.
Contract: `python bench.py --gpus N --steps K --warmup W`.  For N > 1 either launch it under torch.distributed.run (one
rank per GPU, RCCL; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or run it directly: the parent then
starts the N ranks itself as fresh child processes BEFORE touching the GPU, relays rank 0's JSON line and exits non-zero
if any rank failed.  Weak scaling (per-GPU batch fixed).  Rank 0 prints ONE JSON line.
"""
import argparse
import copy
import json
import os
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd")]

CFG = {"run": {"batch_size": 512, "world_size": 1, "num_dataloader_workers": 0, "save_path": "/tmp/audiossl_bench/",
               "precision": "bf16"},
       "pretrain": {"base_encoder": {"type": "AudioNTT2020Task6", "output_dim": 2048, "return_all_layers": True},
                    "projection_dim": 2048, "contrastive_dim": 128, "normalization": "mean_var",
                    "lambda_barlow": [5e-5, 5e-5, 5e-5], "loss_scale": "1/32",
                    "input": {"type": "raw_wav", "sampling_rate": 16000, "length_wave": 1.0, "n_mels": 64},
                    "augmentations": {"MixupBYOLA": {"ratio": 0.4, "log_mixup_exp": True},
                                      "RandomResizeCrop": {"virtual_crop_scale": [1.0, 1.5], "freq_crop_scale": [0.6, 1.5],
                                                           "time_crop_scale": [0.6, 1.5]}}}}
PEAK_TFLOPS = {0: 157.3, 1: 2500.0}          # MI355X_MICROARCH.md: f32-in MFMA = vector peak; bf16 dense MFMA
PEAK_HBM_GBS = 8000.0                        # HBM3E
ES = {0: 4, 1: 2, 2: 2}                      # bytes per element of the dtype codes of the C ABI


def _work(entry, a):
    """Algorithmic work of one launch from the scalar arguments of its C-ABI call (include/audiossl_hip.h order):
    -> (bound, bytes or flops, dtype code) or None.  HBM figures = bytes every operand is read / written ONCE (DESIGN.md
    section 4); MFMA figures = 2 * MACs."""
    if entry == "logmel_fwd":
        B, L, T, _, _, nm = a[:6]
        return "hbm", B * (4.0 * L + 4.0 * nm * T), 0
    if entry == "clip_moments":
        return "hbm", 4.0 * a[0] * a[1], 0
    if entry == "aug_normalize":
        return "hbm", 8.0 * a[2] * a[3], 0
    if entry == "aug_views":                      # self + partner rows read, two views written
        return "hbm", 16.0 * a[1] * a[2] * a[3], 0
    if entry == "conv1_stats":
        return "hbm", 4.0 * a[0] * a[1] * a[2], 0
    if entry == "conv1_fwd":
        d, n, f, t = a[:4]
        return "hbm", 4.0 * n * f * t + n * (t // 2) * (f // 2) * 64.0 * ES[d], 0
    if entry == "conv1_bwd":
        n, f, t = a[2:5]
        return "hbm", 4.0 * n * f * t + n * (t // 2) * (f // 2) * 64.0 * 4, 0
    if entry == "colstats":
        return "hbm", float(a[1] * a[2] * a[3] * ES[a[0]]), 0
    if entry == "bn_relu_pool_fwd":
        d, yd, n, ti, fi = a[:5]
        return "hbm", n * ti * fi * 64.0 * ES[yd] + n * (ti // 2) * (fi // 2) * 64.0 * ES[d], 0
    if entry == "bn_relu_pool_train_fwd":       # (dtype, ydtype, replicas, count, momentum, eps, N, Ti, Fi)
        d, yd = a[:2]
        n, ti, fi = a[-3:]
        return "hbm", n * ti * fi * 64.0 * ES[yd] + n * (ti // 2) * (fi // 2) * 64.0 * ES[d], 0
    if entry == "tmean3_fwd":                   # (dtype, out_f32, To1, Fo1, To2, Fo2, To3, Fo3, N); x_1 comes from the stem's parts
        return "hbm", a[8] * 64.0 * (ES[a[0]] * (a[4] * a[5] + a[6] * a[7]) + 16.0 * a[3]), 0
    if entry == "bn_relu_pool_bwd":
        d, yd, gd, n, ti, fi = a[:6]
        return "hbm", n * ti * fi * 64.0 * (ES[yd] + ES[d]) + n * (ti // 2) * (fi // 2) * 64.0 * ES[gd], 0
    if entry == "bn_relu_pool_bwd_p":           # sums from the pooled output P (+ dP), then the apply sweep (Y, dP -> dY)
        d, yd, gd, n, ti, fi = a[:6]
        return "hbm", n * ti * fi * 64.0 * (ES[yd] + ES[d]) + n * (ti // 2) * (fi // 2) * 64.0 * (2 * ES[gd] + ES[d]), 0
    if entry == "tmean_fwd":
        return "hbm", a[2] * a[3] * a[4] * 64.0 * ES[a[0]], 0
    if entry == "maxmean_fwd":
        return "hbm", float(a[2] * a[3] * a[4] * ES[a[0]]), 0
    if entry == "maxmean_bwd":
        return "hbm", float(a[2] * a[3] * a[4] * 2 * ES[a[0]]), 0
    if entry in ("conv3x3_fwd", "conv3x3_wgrad"):
        n, ti, fi = a[-3:]
        return "mfma", 2.0 * n * ti * fi * 64 * 576, 1
    if entry == "gemm":
        return "mfma", 2.0 * a[3] * a[4] * a[5], a[0]
    if entry == "gemm_dropout":                 # (trans_a, trans_b, M, N, K, ...): the fc GEMM that draws its dropout mask itself
        return "mfma", 2.0 * a[2] * a[3] * a[4], 1
    if entry == "colbn_train_fwd":
        return "hbm", float(a[5] * a[6] * a[7] * (ES[a[1]] + ES[a[0]])), 0
    if entry == "colbn_train_fwd_multi":
        return "hbm", float(a[0] * a[5] * a[6] * a[7] * (ES[a[1]] + 2)), 0
    if entry == "colbn_bwd_multi":
        return "hbm", float(a[0] * a[4] * a[5] * a[6] * (ES[a[1]] + ES[a[2]] + 2)), 0
    if entry == "center_cast":
        return "hbm", 6.0 * a[0] * a[1] * a[2], 0
    if entry == "moco_ce_fwd":
        return "hbm", 4.0 * a[0] * a[1], 0
    if entry == "moco_ce_bwd":
        return "hbm", 6.0 * a[1] * a[2], 0
    if entry == "sgd_momentum":
        return "hbm", 20.0 * a[0], 0
    if entry == "cast":
        return "hbm", 6.0 * a[1], 0
    if entry == "ema_update":
        return "hbm", 12.0 * a[0], 0
    if entry == "dropout_mask":
        return "hbm", 1.0 * a[0], 0
    return None


GEMM_SYMBOL = {(0, 0): "NT", (0, 1): "NN", (1, 1): "TN", (1, 0): "TT"}
# C-ABI entry -> substring of the kernel name rocprofv3 prints for it (profiles/*kernel_stats.csv); the GEMM family is resolved
# at run time (audiossl_last_kernel: the dispatch picks among several instantiations), everything else launches <entry>_kernel
SYMBOLS = {"sgd_momentum": ["sgd_kernel"], "conv3x3_fwd": ["conv3x3_ws_kernel"], "conv3x3_wgrad": ["conv3x3_wgrad_kernel", "wgrad_reduce_kernel"],
           "conv1_fwd": ["conv1_fwd_mfma_kernel"], "conv1_bwd": ["conv1_bwd_mfma_kernel", "conv1_bwd_finalize_kernel"],
           "conv1_stats": ["conv1_moments_kernel", "conv1_finalize_kernel"], "logmel_fwd": ["logmel2_kernel"], "ema_update": ["ema_kernel"],
           "bn_relu_pool_bwd": ["bn_relu_pool_bwd_kernel"], "bn_relu_pool_bwd_p": ["bn_relu_pool_bwd_kernel", "bn_pool_bwd_stats_p_kernel"],
           "cast": ["cast_kernel"], "sgd_momentum_segments": ["sgd_segments_kernel"], "zero_segments": ["zero_segments_kernel"]}


def _symbols(entry):
    base = entry.split("[")[0]
    return SYMBOLS.get(base, [base + "_kernel"])


def per_kernel_report(prof, prof_steps, step_ms):
    """-> list of rows sorted by time.  One row per C-ABI entry point (GEMM family: per entry, operand layout AND kernel the
    dispatch picked), each with the kernel symbols that join it to rocprofv3's kernel_stats.csv.  Durations are HIP events around
    each call on the stream it was issued on, during an eager re-issue of the step."""
    rows = {}
    for full, recs in prof.items():
        entry = full[len("audiossl_"):]
        for e0, e1, a, note in recs:
            sec = e0.elapsed_time(e1) * 1e-3
            sym = None
            if isinstance(note, tuple):
                note, sym = note
            w = _work(entry, a)
            key = entry
            if entry == "gemm":
                key = f"gemm<{'bf16' if a[0] else 'f32'},{GEMM_SYMBOL[(a[1], a[2])]}>"
            elif entry == "gemm_dropout":
                key = f"gemm<bf16,{GEMM_SYMBOL[(a[0], a[1])]}>"
            elif entry == "gemm_multi":
                key = f"gemm_multi<{GEMM_SYMBOL[(a[1], a[2])]}>"
                w = ("mfma", note, 1) if note else None
            elif entry == "gemm_multi_sgd":                   # weight gradients applied in the epilogue: flops AND optimiser bytes
                key = f"gemm_multi_sgd<{GEMM_SYMBOL[(a[1], a[2])]}>"
                w = ("mfma", note[0], 1) if note else None
                byts = note[1] if note else 0.0
            elif entry == "gemm_multi_barlow":
                w = ("mfma", note, 1) if note else None
            elif entry == "moco_logits":                      # (mode, B, K, dim, 1/T, gscale)
                key = f"moco_logits[mode={a[0]}]"
                w = ("mfma", 2.0 * a[1] * a[2] * a[3], 1)
            elif entry in ("conv3x3_fwd", "bn_relu_pool_fwd", "bn_relu_pool_train_fwd", "bn_relu_pool_bwd", "bn_relu_pool_bwd_p", "conv3x3_wgrad",
                           "tmean_fwd"):
                key = f"{entry}[F={a[-1]}]"
            syms = [sym] if sym else _symbols(entry)
            if entry == "conv3x3_fwd":                        # the instantiations of this tile width only (stats / plain / fp32-output forms)
                syms = [f"conv3x3_ws_kernel<{a[-1]},"]
            elif entry == "conv3x3_wgrad":
                syms = [f"conv3x3_wgrad_kernel<{a[-1]},", "wgrad_reduce_kernel"]
            r = rows.setdefault((key, tuple(syms)), {"sec": 0.0, "work": 0.0, "n": 0, "bound": None, "dt": 0, "bytes": 0.0})
            r["sec"] += sec; r["n"] += 1
            if entry == "gemm_multi_sgd":
                r["bytes"] += byts
            if w is not None:
                r["bound"], r["dt"] = w[0], w[2]
                r["work"] += w[1]
    out = []
    for (key, syms), r in sorted(rows.items(), key=lambda kv: -kv[1]["sec"]):
        row = {"entry": key, "kernel_symbols": list(syms), "launches_per_step": round(r["n"] / prof_steps, 2),
               "avg_us": round(r["sec"] / r["n"] * 1e6, 2), "share_of_step": round(r["sec"] / prof_steps / (step_ms * 1e-3), 4)}
        if r["bound"] == "hbm" and r["sec"] > 0:
            ach = r["work"] / r["sec"] / 1e9
            row.update(bound="hbm", achieved=round(ach, 1), unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4), peak=PEAK_HBM_GBS,
                       work_per_launch=r["work"] / r["n"])
        elif r["bound"] == "mfma" and r["sec"] > 0:
            ach = r["work"] / r["sec"] / 1e12
            row.update(bound="mfma", achieved=round(ach, 1), unit="TFLOP/s", frac=round(ach / PEAK_TFLOPS[r["dt"]], 4),
                       peak=PEAK_TFLOPS[r["dt"]], work_per_launch=r["work"] / r["n"])
            if r["bytes"]:                                    # a GEMM that also carries the optimiser update: its byte side
                gbs = r["bytes"] / r["sec"] / 1e9
                row.update(epilogue="SGD update of the weights in the epilogue (parameter + momentum read and written, bf16 shadow)",
                           bytes_per_launch=r["bytes"] / r["n"], hbm_GBs=round(gbs, 1), hbm_frac=round(gbs / PEAK_HBM_GBS, 4))
        out.append(row)
    return out


def _latest_profile(suffix):
    """newest committed profiles/rNN_<suffix> (by round number), or None"""
    import glob
    import re
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)):
        n = int(re.match(r"r(\d+)_", os.path.basename(path)).group(1))
        if best is None or n > best[0]:
            best = (n, path)
    return best[1] if best else None


def in_graph_stats(symbols):
    """(avg ns, calls, csv path, name of the CSV's top row) of the kernels whose rocprofv3 name contains one of `symbols`, from the
    committed `rocprofv3 --kernel-trace --stats` summary of this same command (graph replays: kernels of different branches of
    the step share the machine there, so these averages are longer than the isolated HIP-event ones)."""
    import csv
    path = _latest_profile("bench_b512_kernel_stats.csv")
    if path is None:
        return None
    tot, calls, top = 0.0, 0, None
    with open(path, newline="") as f:
        for rec in csv.DictReader(f):
            if top is None:
                top = rec["Name"]
            if any(sy in rec["Name"] for sy in symbols):
                tot += float(rec["TotalDurationNs"]); calls += int(rec["Calls"])
    return (tot / calls if calls else None, calls, os.path.relpath(path, ROOT), top)


def pmc_traffic(symbols):
    """HBM bytes (read + written) per launch of the kernels named by `symbols`, from the committed PMC passes
    (profiles/rNN_pmc_traffic.json, made by tools/pmc_summary.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs of this
    command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes; counters cannot be read live from inside the process)."""
    path = _latest_profile("pmc_traffic.json")
    if path is None:
        return None
    tot, n = 0.0, 0
    for name, v in json.load(open(path))["kernels"].items():
        if v.get("read_bytes_per_launch") is None or not any(sy in name for sy in symbols):
            continue
        tot += (v["read_bytes_per_launch"] + (v["write_bytes_per_launch"] or 0)) * v["launches"]
        n += v["launches"]
    return round(tot / n) if n else None


def synth_waves(B, L, seed):
    """Seeded uniform noise x0.1 + 440 Hz + 3 kHz tones; last two clips: silence / full scale."""
    g = np.random.RandomState(seed)
    t = np.arange(L) / 16000.0
    w = g.uniform(-0.1, 0.1, (B, L)) + 0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t)
    if B >= 4:
        w[-2] = 0.0
        w[-1] = 1.0
    return w.astype(np.float32)


def cpu_baseline(B, steps, queue):
    """The CPU oracle (port of the reference path) timed on this host: log-mel + aug + delores_m step."""
    from oracle import augment as OA, frontend as FE, model as OM
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))                 # a 1-GPU box owns a 16-core share of the host
    torch.set_num_threads(cores)
    cfg = copy.deepcopy(CFG)
    np.random.seed(31)
    random.seed(31)
    torch.manual_seed(0)
    ex = OM.DeloresMExpert(cfg, num_negatives=queue).train()
    tf = OA.AugmentationModule(cfg, 100000)
    mel = FE.MelSpectrogram()
    waves = torch.from_numpy(synth_waves(B, 16000, 1234))
    params = [p for p in ex.parameters() if p.requires_grad]
    bufs = {}

    def step():
        lms = FE.log_mel_batch(waves, mel)
        v = [tf(lms[b][None]) for b in range(B)]
        a = torch.stack([x[0] for x in v])
        b = torch.stack([x[1] for x in v])
        keep_q = (torch.rand(B, 12, 2048) >= 0.3).float()
        keep_k = (torch.rand(B, 12, 2048) >= 0.3).float()
        for p in params:
            p.grad = None
        loss = ex.training_loss(a, b, keep_q, keep_k)
        loss.backward()
        OM.sgd_momentum_step(params, bufs, 0.03, 0.9, 1e-4)
        return float(loss)
    tw = time.perf_counter()
    step()
    print(f"[bench] cpu_baseline warm-up step: {time.perf_counter() - tw:.1f} s on {cores} threads", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    done = 0
    for _ in range(steps):
        step()
        done += 1
        if done % 8 == 0:
            print(f"[bench] cpu_baseline step {done}: {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
        if time.perf_counter() - t0 > 20.0:
            break
    steps = done
    dt = time.perf_counter() - t0
    return {"value": B * steps / dt, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} steps of batch {B} (log-mel + two views + delores_m fwd/bwd/SGD, queue {queue}), fp32, torch-CPU oracle"}


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this process never touches the GPU and
    is not replaced), stream their stderr through, print rank 0's stdout (the JSON line), return the worst exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print(f"[bench] ranks failed (rank, exit code): {bad}", file=sys.stderr)
    return max((abs(rc) for rc in rcs), default=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="per-GPU batch (weak scaling)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16_hp", "fp32"])
    ap.add_argument("--queue", type=int, default=65536)
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=80, help="upper bound; the CPU leg also stops after ~20 s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="issue the step kernel by kernel instead of replaying the hipGraph")
    ap.add_argument("--graph-phases", action="store_true", help="single rank: use the data-parallel variant (one graph per phase)")
    ap.add_argument("--h2d", action="store_true", help="upload a fresh pinned host batch (B x 16000 fp32) every step on a copy stream, "
                    "two batches ahead of the training step, instead of re-using the batch resident in HBM (reported as "
                    "`h2d`; the contract's `value` is measured without it)")
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))         # no GPU call has happened in this process
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("AUDIOSSL_SHARE_GPU") == "1":                # rehearsal of the N > 1 path on a one-GPU box: every rank on
        local = 0                                                  # cuda:0, gloo transport (RCCL refuses two ranks per device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rccl = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("AUDIOSSL_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                       # every rank answers: the collective layer is alive
        rccl = {"backend": "rccl (torch 'nccl')" if backend == "nccl" else backend, "ranks_in_all_reduce": int(ones.item())}

    from src import _native as N
    from src.augmentations import AugmentationModule
    from src.dataset import UpstreamFrontEnd
    from src.encoder import AudioNTT2020Task6
    from src.upstream.delores_m.upstream_expert import Upstream_Expert
    N.lib()

    cfg = copy.deepcopy(CFG)
    cfg["run"]["precision"] = args.precision
    cfg["run"]["batch_size"] = args.batch
    B = args.batch
    np.random.seed(31 + rank)
    random.seed(31 + rank)
    torch.manual_seed(0)                                           # identical initial weights on every rank
    model = Upstream_Expert(cfg, base_encoder=AudioNTT2020Task6, num_negatives=args.queue).to(dev).train()
    tfms = AugmentationModule(cfg, 100000, max_batch=B)
    front = UpstreamFrontEnd(cfg, tfms)
    opt = model.configure_optimizers()
    waves = torch.from_numpy(synth_waves(B, 16000, 1234 + rank)).to(dev)

    # single rank: the product's step is one hipGraph replay (zero_grad + fused fwd/bwd + SGD) behind the eager front end;
    # data-parallel ranks replay one graph per collective-free phase with the RCCL calls in between.
    gstep = None if args.no_graph else model.graphed_step(opt, phases=args.graph_phases)

    # --h2d: the loader's side of the boundary - a pinned host batch per step, uploaded on a copy stream two batches ahead of the
    # training step that consumes it (ring of three device buffers), the front end waits for the upload's event only
    if args.h2d:
        host = [torch.from_numpy(synth_waves(B, 16000, 1234 + rank + 7 * j)).pin_memory() for j in range(3)]
        ring = [torch.empty(B, 16000, device=dev) for _ in range(3)]
        copy_stream = torch.cuda.Stream(device=dev)
        uploads, fe_done = {}, {}

        def upload(j):
            if j - 3 in fe_done:
                copy_stream.wait_event(fe_done.pop(j - 3))                # the buffer's previous reader: the front end of batch j - 3
            with torch.cuda.stream(copy_stream):
                ring[j % 3].copy_(host[j % 3], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            uploads[j] = ev

        def submit_next(j):
            t = front.submit(ring[j % 3], after=uploads.pop(j))
            fe_done[j] = t[1]
            upload(j + 2)
            return t
        upload(0); upload(1)
        nxt = [0]
    else:
        def submit_next(j):
            return front.submit(waves)
        nxt = [0]

    def next_ticket():
        t = submit_next(nxt[0])
        nxt[0] += 1
        return t
    ticket = [next_ticket()]

    def step(i):
        # the front end of the NEXT batch is submitted (own stream) before this batch's training step is launched, so it
        # runs underneath it; every call does exactly one front end and one training step
        img_1, img_2 = front.collect(ticket[0])
        ticket[0] = next_ticket()
        if gstep is not None:
            return gstep(img_1, img_2)
        opt.zero_grad()
        loss = model.training_step((img_1, img_2), i)
        loss.backward()
        model.all_reduce_grads()
        opt.step()
        return loss

    if gstep is not None:
        for i in range(gstep.eager_steps + 1):                     # untimed: eager priming steps + the capture itself
            step(0)
    for i in range(args.warmup):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if gstep is None:
        N.PROFILE, N.PROFILE_ALL = {}, True
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    t_host = time.perf_counter() - t0                              # launch-side time, before the device drains
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    final_loss = float(loss.detach())
    if gstep is not None:
        # kernels inside a replayed graph cannot carry HIP events: time the dominant kernel on an eager re-issue of the
        # very same step (same buffers, same streams) right after the timed region
        N.PROFILE, N.PROFILE_ALL = {}, True
        prof_steps = min(args.steps, 5)
        for i in range(prof_steps):
            img_1, img_2 = front.collect(ticket[0])
            ticket[0] = next_ticket()
            gstep._eager(img_1, img_2)
        torch.cuda.synchronize()
    else:
        prof_steps = args.steps
    prof, N.PROFILE, N.PROFILE_ALL = N.PROFILE, None, False
    if rank == 0:
        print(f"[bench] gpu: {B * world * args.steps / dt:.1f} clips/s, {dt / args.steps * 1e3:.2f} ms/step "
              f"(host launch side {t_host / args.steps * 1e3:.2f} ms/step)", file=sys.stderr, flush=True)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    if rank != 0:
        return

    # ---- per-kernel roofline (north_star: GB/s for the mel / augment kernels, TFLOP/s for the encoder GEMMs); the headline
    #      `roofline` object = the row with the largest share of the step (the dominant kernel), named by its rocprofv3 symbol
    per_kernel = per_kernel_report(prof, prof_steps, dt / args.steps * 1e3)
    # Which kernel is "the dominant one": the top row of the committed rocprofv3 summary of this same command (kernel time inside
    # the replayed graph) when one exists and this run launched that kernel; otherwise the live row with the largest share.  All
    # live rows that launch that kernel symbol (e.g. the heads' weight gradients AND the Barlow correlation: the same
    # instantiation) are merged into the headline.
    rows = [r for r in per_kernel if r.get("bound")]
    csv_top = (in_graph_stats(["\0"]) or (None, 0, None, None))[3]
    chosen, how = [], None
    if csv_top:
        chosen = [r for r in rows if any(sy in csv_top for sy in r["kernel_symbols"])]
        how = "top row of the committed kernel_stats.csv"
    if not chosen:
        chosen, how = rows[:1], "largest share of the step in this run"
    top = chosen[0]
    symbols = sorted({sy for r in chosen for sy in r["kernel_symbols"] if not csv_top or sy in csv_top} or set(top["kernel_symbols"]))
    n_l = sum(r["launches_per_step"] for r in chosen)
    t_us = sum(r["avg_us"] * r["launches_per_step"] for r in chosen)
    work = sum(r["work_per_launch"] * r["launches_per_step"] for r in chosen)
    scale = 1e12 if top["bound"] == "mfma" else 1e9
    achieved = work / (t_us * 1e-6) / scale
    roofline = {"bound": top["bound"], "achieved": round(achieved, 1), "peak": top["peak"], "unit": top["unit"],
                "frac": round(achieved / top["peak"], 4), "traffic": pmc_traffic(symbols), "kernel": symbols[0],
                "entry": [r["entry"] for r in chosen], "selected_by": how, "launches_per_step": round(n_l, 2),
                "avg_launch_us": round(t_us / n_l, 2), "work_per_launch": work / n_l,
                "share_of_step": round(sum(r["share_of_step"] for r in chosen), 4),
                "timed_on": "eager re-issue of the step after the timed region" if gstep is not None else "the timed region"}
    fused = [r for r in chosen if r.get("bytes_per_launch")]
    if fused:
        # some launches of the headline kernel also apply the optimiser's update to their result: their HBM side, for the reader who
        # prices the kernel by flops alone
        fb = sum(r["bytes_per_launch"] * r["launches_per_step"] for r in fused)
        ft = sum(r["avg_us"] * r["launches_per_step"] for r in fused)
        roofline["epilogue"] = {"what": fused[0]["epilogue"], "launches_per_step": round(sum(r["launches_per_step"] for r in fused), 2),
                                "bytes_per_launch": fb / sum(r["launches_per_step"] for r in fused),
                                "hbm_GBs": round(fb / (ft * 1e-6) / 1e9, 1), "hbm_frac": round(fb / (ft * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)}
    ig = in_graph_stats(symbols)
    if ig is not None and ig[0]:
        # the same algorithmic work over the average duration rocprofv3 saw for this kernel INSIDE the replayed graph
        per_s = roofline["work_per_launch"] / (ig[0] * 1e-9)
        roofline.update(frac_in_graph=round(per_s / (top["peak"] * scale), 4), in_graph_avg_us=round(ig[0] * 1e-3, 2),
                        in_graph_source=ig[2], profile_top_kernel=ig[3])
    for r in per_kernel:
        r.pop("peak", None)
    roofline["per_kernel"] = per_kernel[:32]
    out = {"metric": "upstream clips/sec (1s@16kHz, 64-mel)", "value": round(B * world * args.steps / dt, 1), "unit": "clips/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.precision.startswith("bf16") else "f32",
           "precision_mode": args.precision, "data": "synthetic", "hip_graph": gstep is not None,
           "config": {"workload": "delores-m upstream step (log-mel + 2 views + q/k conv encoders + MoCo + 3 Barlow heads + bwd + SGD), "
                                  f"1 s @ 16 kHz, 64 mel, batch {B}/GPU, queue {args.queue}", "global_batch": B * world,
                      "parallelism": f"dp{world}"},
           "final_loss": final_loss, "roofline": roofline}
    if gstep is not None:
        ph = gstep.phases
        out["graph_mode"] = ({"mode": "one hipGraph per collective-free phase", "phases": sorted(ph.graphs), "eager_fallback": ph.broken}
                             if gstep.use_phases and ph is not None else {"mode": "single hipGraph"})
    else:
        out["graph_mode"] = {"mode": "eager"}
    if rccl is not None:
        out["collectives"] = rccl
    if args.h2d:
        out["h2d"] = {"bytes_per_step": B * 16000 * 4, "pinned": True, "prefetch_batches": 2}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.cpu_steps, args.queue)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
