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


def per_kernel_report(prof, prof_steps, step_ms):
    """-> (list of per-entry-point rows sorted by time, the GEMM groups).  Durations are HIP events around each C-ABI call on
    the stream it was issued on, during an eager re-issue of the step; rocprofv3 --kernel-trace --stats of the same command
    (profiles/) is the cross-check."""
    rows, gemm_groups = {}, {}
    for full, recs in prof.items():
        entry = full[len("audiossl_"):]
        for e0, e1, a, note in recs:
            sec = e0.elapsed_time(e1) * 1e-3
            w = _work(entry, a)
            if entry == "gemm_multi" and note:
                w = ("mfma", note, 1)
            key = entry
            if entry == "gemm":
                key = f"gemm<{'bf16' if a[0] else 'f32'},{GEMM_SYMBOL[(a[1], a[2])]}>"
                g = gemm_groups.setdefault((a[0], a[1], a[2]), [0.0, 0.0, 0])
                g[0] += sec; g[1] += 2.0 * a[3] * a[4] * a[5]; g[2] += 1
            elif entry in ("conv3x3_fwd", "bn_relu_pool_fwd", "bn_relu_pool_train_fwd", "bn_relu_pool_bwd", "conv3x3_wgrad", "tmean_fwd"):
                key = f"{entry}[F={a[-1]}]"
            r = rows.setdefault(key, {"sec": 0.0, "work": 0.0, "n": 0, "bound": None, "dt": 0})
            r["sec"] += sec; r["n"] += 1
            if w is not None:
                r["bound"], r["dt"] = w[0], w[2]
                r["work"] += w[1]
    out = []
    for key, r in sorted(rows.items(), key=lambda kv: -kv[1]["sec"]):
        row = {"entry": key, "launches_per_step": round(r["n"] / prof_steps, 2), "avg_us": round(r["sec"] / r["n"] * 1e6, 2),
               "share_of_step": round(r["sec"] / prof_steps / (step_ms * 1e-3), 4)}
        if r["bound"] == "hbm" and r["sec"] > 0:
            ach = r["work"] / r["sec"] / 1e9
            row.update(bound="hbm", achieved=round(ach, 1), unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4))
        elif r["bound"] == "mfma" and r["sec"] > 0:
            ach = r["work"] / r["sec"] / 1e12
            row.update(bound="mfma", achieved=round(ach, 1), unit="TFLOP/s", frac=round(ach / PEAK_TFLOPS[r["dt"]], 4))
        out.append(row)
    return out, gemm_groups
GEMM_SYMBOL = {(0, 0): "NT", (0, 1): "NN", (1, 1): "TN", (1, 0): "TT"}


def pmc_traffic(ta, tb):
    """HBM bytes per launch of the bf16 GEMM instantiations with these transposes, from the committed PMC passes
    (profiles/r02_pmc_traffic.json, made by tools/pmc_summary.py; counters cannot be read live from inside the process)."""
    import re
    path = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
    if not os.path.exists(path):
        path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    tot, n = 0.0, 0
    for name, v in json.load(open(path))["kernels"].items():
        if not any(k in name for k in ("gemm_kernel", "gemm_ring_kernel", "gemm_bk32_kernel", "gemm_p8_kernel")) or v["read_bytes_per_launch"] is None:
            continue
        m = (re.search(r"gemm_kernelIDF16bLb(\d)ELb(\d)ELi\d+E", name) or re.search(r"gemm_ring_kernelILb(\d)ELb(\d)ELi\d+E", name)
             or re.search(r"gemm_bk32_kernelILb(\d)ELb(\d)E", name) or re.search(r"gemm_p8_kernelILb(\d)ELb(\d)E", name))
        key = (int(m.group(1)), int(m.group(2))) if m else ((1, 1) if re.search(r"E, true, \d+(, \d+)*>", name) else None)
        d = re.search(r"gemm_(?:ring|bk32)_kernel<(false|true), (false|true)", name)    # demangled form of the ring / BK=32 kernels
        if d:
            key = (int(d.group(1) == "true"), int(d.group(2) == "true"))
        if key == (ta, tb):
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("AUDIOSSL_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

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

    ticket = [front.submit(waves)]

    def step(i):
        # the front end of the NEXT batch is submitted (own stream) before this batch's training step is launched, so it
        # runs underneath it; every call does exactly one front end and one training step
        img_1, img_2 = front.collect(ticket[0])
        ticket[0] = front.submit(waves)
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
            ticket[0] = front.submit(waves)
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

    # ---- per-kernel roofline (north_star: GB/s for the mel / augment kernels, TFLOP/s for the encoder GEMMs) and the
    #      headline `roofline` object = the dominant GEMM instantiation group (largest total time)
    per_kernel, groups = per_kernel_report(prof, prof_steps, dt / args.steps * 1e3)
    (dtype, ta, tb), (tsec, flops, launches) = max(groups.items(), key=lambda kv: kv[1][0])
    achieved = flops / tsec / 1e12
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_TFLOPS[dtype], 4), "traffic": pmc_traffic(ta, tb) if dtype == 1 else None,
                "kernel": f"gemm_kernel<{'bf16' if dtype else 'f32'},{GEMM_SYMBOL[(ta, tb)]}>", "launches_per_step": launches / prof_steps,
                "avg_launch_us": round(tsec / launches * 1e6, 2), "flop_per_launch": flops / launches,
                "share_of_step": round(tsec / prof_steps / (dt / args.steps), 3),
                "timed_on": "eager re-issue of the step after the timed region" if gstep is not None else "the timed region",
                "per_kernel": per_kernel[:28]}
    out = {"metric": "upstream clips/sec (1s@16kHz, 64-mel)", "value": round(B * world * args.steps / dt, 1), "unit": "clips/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.precision.startswith("bf16") else "f32",
           "precision_mode": args.precision, "data": "synthetic", "hip_graph": gstep is not None,
           "config": {"workload": "delores-m upstream step (log-mel + 2 views + q/k conv encoders + MoCo + 3 Barlow heads + bwd + SGD), "
                                  f"1 s @ 16 kHz, 64 mel, batch {B}/GPU, queue {args.queue}", "global_batch": B * world,
                      "parallelism": f"dp{world}"},
           "final_loss": final_loss, "roofline": roofline}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.cpu_steps, args.queue)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if world > 1:
        pass


if __name__ == "__main__":
    main()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
