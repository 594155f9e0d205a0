"""Diagnostic (GPU box): where does the two-rank graph-phase step go wrong?  (VERDICT r1 "weak" 4)

Runs the two-rank delores_s / delores_m step of tests/test_gpu_ddp.py (both ranks on cuda:0, gloo) with stream-ordered
snapshots of every buffer the collectives and the optimiser touch, and checks per step, on each rank:
  local gradient (snapshot taken on the launch stream right before each all-reduce starts)  vs the eagerly issued twin run
  all-reduced gradient (snapshot right before the SGD launch)                                vs sum over ranks of the locals
  weights / momentum after SGD                                                               vs the update recomputed from the snapshots
  weights before SGD of step s+1                                                             vs weights after SGD of step s
Usage: python tools/ddp_diag.py [delores_s|delores_m] [reps] [drain 0|1] [inline_wgrad 0|1]
"""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd"), os.path.join(ROOT, "tests")]
os.environ["PYTHONPATH"] = os.pathsep.join(sys.path[:3] + [os.environ.get("PYTHONPATH", "")])

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, which, steps, graph, drain, inline, ret):
    import copy
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(0)
    torch.manual_seed(77)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import CFG_M, CFG_S
        from oracle import fill
        from helpers import closed_queue, views
        from src import engine as E
        from src.encoder import AudioNTT2020Task6
        from src.upstream import common as C
        B, T, K = 16, 96, 256
        if which == "delores_m":
            from src.upstream.delores_m.upstream_expert import Upstream_Expert
            cfg = copy.deepcopy(CFG_M); cfg["run"]["precision"] = "bf16"
            m = Upstream_Expert(cfg, base_encoder=AudioNTT2020Task6, num_negatives=K)
        else:
            from src.upstream.delores_s.upstream_expert import Upstream_Expert
            cfg = copy.deepcopy(CFG_S); cfg["run"]["precision"] = "bf16"
            m = Upstream_Expert(cfg, base_encoder=AudioNTT2020Task6)
        fill.fill_state_dict_(m, seed=33)
        if which == "delores_m":
            for pq, pk in zip(m.encoder_q.parameters(), m.encoder_k.parameters()):
                pk.data.copy_(pq.data)
            m.queue.copy_(closed_queue(128, K))
        m = m.cuda().train()
        if inline:
            E.SideStream.run = lambda self, device, fn: fn()
            E.SideStream.join = lambda self, device: None
        opt = m.configure_optimizers()
        step = m.graphed_step(opt, eager_steps=1 if graph else 10 ** 9)
        snaps = {}
        orig_begin = m.reduce_begin

        def reduce_begin(which_seg):
            ho = m.head_offset()
            lo, hi = (ho, m.flat.numel) if which_seg == "heads" else (0, ho)
            snaps["local_" + which_seg] = (lo, hi, m.flat.grad[lo:hi].clone())
            orig_begin(which_seg)
        m.reduce_begin = reduce_begin

        def eager(img_1, img_2, runner=None):
            kw = {} if runner is None else {"runner": runner}
            opt.grad_scale_tensor = None
            loss = m.fused_loss(img_1, img_2, True, **kw)
            m.all_reduce_grads()
            if runner is not None and drain:
                torch.cuda.current_stream().synchronize()
            snaps["g_red"] = m.flat.grad.clone()
            snaps["w_before"] = m.flat.data.clone()
            snaps["m_before"] = None if m.flat.momentum is None else m.flat.momentum.clone()
            opt.step()
            snaps["w_after"] = m.flat.data.clone()
            snaps["m_after"] = m.flat.momentum.clone()
            return loss
        step._eager = eager
        trace, prev_after = [], None
        g0 = opt.param_groups[0]
        for s in range(steps):
            a = views(B, T, 9300 + 10 * s + rank).cuda()
            b = views(B, T, 9350 + 10 * s + rank).cuda()
            loss = float(step(a, b))
            torch.cuda.synchronize()
            rec = {"loss": loss}
            local = torch.zeros(m.flat.numel)
            for k in ("local_heads", "local_enc"):
                lo, hi, t = snaps[k]
                local[lo:hi] = t.cpu()
            want = local.clone()
            dist.all_reduce(want)
            g_red = snaps["g_red"].cpu()
            rec["reduce_err"] = float((g_red - want).abs().max() / want.abs().max())
            rec["local_norms"] = [float(local[o:o + p.numel()].norm()) for p, o in zip(m.flat.params, m.flat.offsets)]
            wb, wa = snaps["w_before"].cpu().double(), snaps["w_after"].cpu().double()
            g = g_red.double() / world + g0["weight_decay"] * wb
            mb = snaps["m_before"]
            buf = g if mb is None else g0["momentum"] * mb.cpu().double() + g
            rec["sgd_w_err"] = float((wa - (wb - g0["lr"] * buf)).abs().max() / (g0["lr"] * buf.abs().max()))
            rec["sgd_m_err"] = float((snaps["m_after"].cpu().double() - buf).abs().max() / buf.abs().max())
            rec["carry_err"] = 0.0 if prev_after is None else float((snaps["w_before"].cpu() - prev_after).abs().max())
            prev_after = snaps["w_after"].cpu()
            rec["w_norm"] = float(wa.norm())
            trace.append(rec)
        ret[rank] = {"trace": trace, "names": list(m.flat.names),
                     "graphs": None if step.phases is None else (sorted(step.phases.graphs), step.phases.broken)}
    finally:
        dist.destroy_process_group()


def run(which, steps, graph, drain, inline):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, which, steps, graph, drain, inline, ret), nprocs=2, join=True)
    return [ret[r] for r in range(2)]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "delores_s"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    drain = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
    inline = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
    steps = 5
    ref = run(which, steps, False, False, False)
    print(f"eager twin: losses r0 {[round(t['loss'], 5) for t in ref[0]['trace']]}", flush=True)
    for r in (0, 1):
        for s, t in enumerate(ref[r]["trace"]):
            if max(t["reduce_err"], t["sgd_w_err"], t["sgd_m_err"], t["carry_err"]) > 1e-4:
                print(f"  EAGER rank {r} step {s}: reduce {t['reduce_err']:.2e} sgd_w {t['sgd_w_err']:.2e} sgd_m {t['sgd_m_err']:.2e} carry {t['carry_err']:.2e}")
    for rep in range(reps):
        got = run(which, steps, True, drain, inline)
        print(f"rep {rep} graph (drain={drain}, inline_wgrad={inline}) {got[0]['graphs']}: losses r0 {[round(t['loss'], 5) for t in got[0]['trace']]}", flush=True)
        for r in (0, 1):
            for s, (t, e) in enumerate(zip(got[r]["trace"], ref[r]["trace"])):
                ln, en = np.array(t["local_norms"]), np.array(e["local_norms"])
                dev = np.abs(ln - en) / (en + 1e-12 * en.max() + 1e-30)
                worst = np.argsort(-dev)[:3]
                flags = []
                if dev.max() > 2e-2:
                    flags.append("LOCAL-GRAD " + ", ".join(f"{got[r]['names'][i]} {ln[i]:.3e} vs {en[i]:.3e}" for i in worst if dev[i] > 2e-2))
                if t["reduce_err"] > 1e-4:
                    flags.append(f"ALLREDUCE err {t['reduce_err']:.2e}")
                if t["sgd_w_err"] > 1e-3 or t["sgd_m_err"] > 1e-3:
                    flags.append(f"SGD w {t['sgd_w_err']:.2e} m {t['sgd_m_err']:.2e}")
                if t["carry_err"] > 0:
                    flags.append(f"CARRY {t['carry_err']:.2e}")
                if abs(t["loss"] - e["loss"]) > 2e-3 * abs(e["loss"]):
                    flags.append(f"LOSS {t['loss']:.5f} vs {e['loss']:.5f}")
                if flags:
                    print(f"  rank {r} step {s}: " + " | ".join(flags), flush=True)


if __name__ == "__main__":
    main()
