"""DeLoRes-M expert on MI355X: MoCo-v2 (query/key encoders, EMA, negatives queue, InfoNCE) plus three Barlow heads on
the intermediate layer means.

Same class name, constructor, buffers (`queue`, `queue_ptr`) and parameter names as
`src/upstream/delores_m/upstream_expert.py:51-317` of the reference.  `training_step` is one fused sequence of HIP
launches: q-encoder fwd, EMA of the key encoder (one launch over the flat parameter buffer), k-encoder fwd,
InfoNCE against the queue, enqueue, three projector heads, and the complete backward.
Multi-GPU (torch.distributed over RCCL): batch shuffle / unshuffle for the key encoder's BatchNorm, key all-gather
before the enqueue (`:156-219`), keyed off the process group instead of the removed `trainer.use_ddp`.
"""
import os

import torch
import torch.nn as nn

from src import _native as N
from src import engine as E
from src.encoder.audiontt import default_precision
from src.flat import FlatGroup
from src.module_base import UpstreamModule
from src.upstream.common import EAGER, FusedExpertMixin, FusedStepFn, MocoQueueMixin, Projection, strip
from src.upstream.delores_m.upstream_encoder import DELORES_M as DELORES_M_ENCODER
from src.utils import concat_all_gather


_PREP_ASIDE = os.environ.get("AUDIOSSL_PREP_ASIDE", "1") != "0"
_HEADS_ASIDE = os.environ.get("AUDIOSSL_HEADS_ASIDE", "1") != "0"        # 0: the grouped Barlow heads are issued on the main stream
_FUSED_SGD = os.environ.get("AUDIOSSL_FUSED_SGD", "1") != "0"          # 0: the heads' weight gradients go through memory to the SGD pass
_SKIP_ZERO = os.environ.get("AUDIOSSL_SKIP_ZERO", "1") != "0"            # 0: the head-segment SGD clears the whole gradient slice again
_GRADS_ZERO = os.environ.get("AUDIOSSL_GRADS_ZERO", "1") != "0"          # 0: weight-gradient GEMMs of the heads add to the (zero) buffers
_DY_EVENT = os.environ.get("AUDIOSSL_DY_EVENT", "1") != "0"              # 0: the main stream waits for the whole heads stream
_LATE_JOIN = os.environ.get("AUDIOSSL_LATE_JOIN", "1") != "0"            # 0: join the heads before the whole encoder backward
_SGD_ASIDE = os.environ.get("AUDIOSSL_SGD_ASIDE", "1") != "0"            # 0: the early head-segment SGD is issued on the main stream


def _world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class Upstream_Expert(MocoQueueMixin, FusedExpertMixin, UpstreamModule):
    prefer_phases = False         # single rank: ONE graph measured faster than one graph per phase (4.1 vs 4.3 ms at B=512)

    def __init__(self, config, base_encoder, datamodule=None, emb_dim: int = 128, num_negatives: int = 65536,
                 encoder_momentum: float = 0.999, softmax_temperature: float = 0.07, learning_rate: float = 0.03,
                 momentum: float = 0.9, weight_decay: float = 1e-4, data_dir: str = './', batch_size: int = 256,
                 use_mlp: bool = False, num_workers: int = 8, *args, **kwargs):
        super().__init__()
        self.save_hyperparameters()
        if use_mlp:
            raise NotImplementedError("use_mlp is not part of the HIP path")
        self.config = config
        self.base_encoder = base_encoder
        self.datamodule = datamodule
        self.encoder_q, self.encoder_k = self.init_encoders(self.base_encoder)
        for param_q, param_k in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            param_k.data.copy_(param_q.data)
            param_k.requires_grad = False
        self.register_buffer("queue", nn.functional.normalize(torch.randn(emb_dim, num_negatives), dim=0))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        lam, scale = self.config["pretrain"]["lambda_barlow"], self.config["pretrain"]["loss_scale"]
        self.p1 = Projection(2048, lam[0], scale)
        self.p2 = Projection(1024, lam[1], scale)
        self.p3 = Projection(512, lam[2], scale)
        prec = config.get("run", {}).get("precision")
        self.precision = {"fp32": N.F32, "bf16": N.BF16, "bf16_hp": N.BF16}.get(prec, default_precision())
        self.high_precision = prec == "bf16_hp"
        self.grouped_heads = bool(config.get("run", {}).get("grouped_heads", os.environ.get("AUDIOSSL_GROUPED_HEADS", "1") != "0"))
        self.encoder_q.encoder.precision = self.encoder_k.encoder.precision = self.precision
        self.flat_k = None
        self._key_stream = E.SideStream()
        self._prep_stream = E.SideStream()

    def init_encoders(self, base_encoder):
        return DELORES_M_ENCODER(self.config, base_encoder), DELORES_M_ENCODER(self.config, base_encoder)

    # ------------------------------------------------------------------ flat storage
    def trainable_named(self):
        # encoder_q first, in the same order as encoder_k's parameters, so that one EMA launch covers both
        named = [(n, p) for n, p in self.named_parameters() if n.startswith("encoder_q.")]
        named += [(n, p) for n, p in self.named_parameters() if n.startswith(("p1.", "p2.", "p3."))]
        return named

    # ------------------------------------------------------------------ fused step
    def fused_loss(self, img_q, img_k, need_grad=True, parts=None, runner=None, optimizer=None):
        with E.ARENA.step(img_q.device):          # every zero-initialised scratch of the step: one arena, one memset
            return self._fused_loss(img_q, img_k, need_grad, parts, runner, optimizer)

    def _fused_loss(self, img_q, img_k, need_grad, parts, runner, optimizer):
        """Forward + backward of the whole step.  The work is written as collective-free *phases* (query encoder, key
        encoder, loss heads, encoder backward) handed to `runner`: the default runs them in place; the data-parallel
        graph step captures each one into a hipGraph (common.GraphPhases).  Collectives sit between the phases."""
        R = runner or EAGER
        dt = self.precision
        E.set_high_precision(self.high_precision)
        flat = self.ensure_flat()
        dev = img_q.device
        B = img_q.shape[0]
        eq, ek = self.encoder_q, self.encoder_k
        ddp = _world() > 1
        img_k = img_k.float().contiguous()
        img_q = img_q.float().contiguous()
        pending_k = self._shuffle_begin(img_k) if ddp else None
        main = torch.cuda.current_stream()
        G = flat.grad_dict
        # stacked projector inputs: rows [0,B) from the query encoder, [B,2B) from the key encoder
        Ys = R.alloc("Ys", lambda: [torch.empty(2 * B, f, dtype=E.pooled_dtype(dt), device=dev) for f in (2048, 1024, 512)])

        # ---- key encoder (no gradient) on its own stream, concurrent with the query encoder: EMA first, then forward
        def key_phase(x):
            self._momentum_update_key_encoder(shadow_dtype=dt)
            self.flat_k.refresh_shadow(dt)
            Wk = self.flat_k.shadow_dict()
            keepk = ek.encoder.next_keep_mask(B, x.shape[-1])
            _, _, _, Hk, _ = E.encoder_forward(ek.encoder.param_dict(), x, dt, keep=keepk, p_drop=0.3, train=self.training,
                                               Wc=strip(Wk, "encoder."), layer_out=tuple(y[B:] for y in Ys))
            yk, _ = E.maxmean_forward(dt, Hk)
            wk = Wk["fc.weight"]
            return E.linear_fwd(dt, yk, wk, B, wk.shape[0], wk.shape[1], bias=ek.fc.bias.data, out_f32=1)

        def key_branch():
            x, idx_unshuffle = img_k, None
            if ddp:
                x, idx_unshuffle = self._shuffle_end(pending_k, B)
                x = R.static("key_in", x)
            kk = R.phase("key", lambda: key_phase(x))
            if ddp:
                kk = R.static("key_out", self._batch_unshuffle_ddp(kk, idx_unshuffle))
            return kk
        k = self._key_stream.run(dev, key_branch)

        # the grouped heads STORE their nine weight gradients (single unsplit writer each): those need no clearing between steps
        heads = (self.p1, self.p2, self.p3)
        grouped = not (self.high_precision or dt == N.F32 or not self.grouped_heads or E.SYNC_BN is not None)   # SyncBatchNorm: per-head chains
        D_ = self.p1.param_dict()["projector.3.weight"].shape[0]
        kins_ = [p.param_dict()["projector.0.weight"].shape[1] for p in heads]
        stored_grads = tuple(f"p{i + 1}.projector.{j}.weight" for i in range(3) for j in (0, 3, 6)) \
            if (need_grad and grouped and _GRADS_ZERO and _SKIP_ZERO and E.heads_wgrads_store(D_, kins_, 2 * B)) else ()

        # one rank with the flat SGD: those nine weights are updated in the epilogue of their weight-gradient GEMMs (no gradient written,
        # none read back); the head-segment SGD pass then covers the small tensors only
        sgd_ctx = None
        if stored_grads and _FUSED_SGD and optimizer is not None and not ddp and hasattr(optimizer, "fused_wgrad"):
            sgd_ctx = optimizer.fused_wgrad(flat, [f"p{i + 1}." for i in range(3)])

        # ---- query encoder (main stream)
        def query_phase():
            # clearing the flat gradient and refreshing the bf16 weight shadow are two HBM-bound sweeps (152 MB written,
            # 228 MB moved) that nothing needs before the encoder's fc layers: they run on a side stream beside the stem and
            # the convolutions (which read the fp32 parameters) and are joined right before the first shadow weight is used
            def prep():
                if need_grad:
                    flat.zero_grad(stores_ok=bool(stored_grads))
                flat.refresh_shadow(dt)
            # ... when there is anything to sweep: after a step with the fused SGD tail both are no-ops (two flags flipped), and an empty
            # branch in the captured graph is not free (fork + join edges: 1.92 -> 1.90-1.92 ms without it)
            busy = (need_grad and not (flat._fresh_grad is True or (flat._fresh_grad == "partial" and stored_grads))) or \
                   not (flat._fresh_shadow or dt == N.F32)
            if _PREP_ASIDE and busy:
                self._prep_stream.run(dev, prep)
            else:
                prep()
            Wq = flat.shadow_dict("encoder_q.")
            wq = Wq["fc.weight"]
            # [ce, barlow1, barlow2, barlow3]; grouped heads: the Barlow terms are accumulated by the correlation GEMM's epilogue
            # into 32 replicas each, stored behind the four slots (the step's loss is the sum of everything either way)
            fold = not (self.high_precision or dt == N.F32 or not self.grouped_heads)
            loss = torch.zeros(4 + (3 * E.LOSS_REPLICAS if fold else 0), dtype=torch.float32, device=dev)
            keep = eq.encoder.next_keep_mask(B, img_q.shape[-1])
            _, _, _, Hq, cq = E.encoder_forward(eq.encoder.param_dict(), img_q, dt, keep=keep, p_drop=0.3, train=self.training,
                                                Wc=strip(Wq, "encoder."), layer_out=tuple(y[:B] for y in Ys),
                                                before_fc=(lambda: self._prep_stream.join(dev)) if (_PREP_ASIDE and busy) else None)
            yq, argq = E.maxmean_forward(dt, Hq)
            q = E.linear_fwd(dt, yq, wq, B, wq.shape[0], wq.shape[1], bias=eq.fc.bias.data, out_f32=1)
            return loss, wq, Hq, cq, yq, argq, q
        loss, wq, Hq, cq, yq, argq, q = R.phase("query", query_phase)
        self._key_stream.join(dev)
        k.record_stream(main)

        # ---- loss heads: the three Barlow heads are independent of each other and of the MoCo head.  Each is its own
        #      phase on its own stream (a single captured graph would serialise the four branches: ROCm's graph executor
        #      ran them back to back, 2.1 ms instead of 0.5 ms - tools/timeline.py)
        streams = self._streams(dev)
        dys = [None, None, None]
        dy_event = None

        def head_phase(i, p):
            Wp = flat.shadow_dict(f"p{i + 1}.")
            return E.barlow_forward_backward(p.param_dict(), G(f"p{i + 1}."), Ys[i], dt, p.lambd, p.scale_loss,
                                             loss[i + 1:i + 2], need_dy1=True, need_dy2=False,
                                             update_running=self.training, backward=need_grad,
                                             Wc=tuple(Wp[f"projector.{j}.weight"] for j in (0, 3, 6)))
        if not grouped:
            for i, p in enumerate(heads):
                if not E.ONE_STREAM:
                    E.fork(streams[i], main)
                with torch.cuda.stream(streams[i]):
                    dys[i] = R.phase(f"head{i + 1}", lambda i=i, p=p: head_phase(i, p))
        else:
            # run.grouped_heads: the three chains in lock-step on ONE side stream, every GEMM a single multi-problem launch.
            # Measured (tools/phase_times.py, B=512): one head chain 375 us alone, the three on three streams 1,050 us
            # (concurrent queues do not overlap here: 1,430 us with 8 hardware queues), grouped 940 us in isolation - but the
            # whole single-graph step came out slower (3.54 vs 3.27 ms), so the three-stream form stays the default.
            def heads_phase():
                Wps = [flat.shadow_dict(f"p{i + 1}.") for i in range(3)]
                return E.barlow_heads_forward_backward(
                    [p.param_dict() for p in heads], [G(f"p{i + 1}.") for i in range(3)], Ys, dt, [p.lambd for p in heads],
                    [p.scale_loss for p in heads],
                    [loss[4 + i * E.LOSS_REPLICAS:4 + (i + 1) * E.LOSS_REPLICAS] for i in range(3)], update_running=self.training,
                    backward=need_grad, Wcs=[tuple(Wp[f"projector.{j}.weight"] for j in (0, 3, 6)) for Wp in Wps],
                    grads_zero=need_grad and _GRADS_ZERO,          # prep() cleared the flat gradient; each dW has one writer
                    dy_ready=(lambda: dy_event.record()) if dy_event is not None else None, sgd=sgd_ctx)
            hs = streams[0] if _HEADS_ASIDE else main
            if not E.ONE_STREAM and _HEADS_ASIDE:
                E.fork(streams[0], main)
            # one rank: the encoder backward only waits for the heads' data gradients (an event recorded before their weight-
            # gradient launches); the head-segment SGD is issued on the heads' own stream, i.e. after those launches
            if need_grad and not ddp and not E.ONE_STREAM and _HEADS_ASIDE and _SGD_ASIDE and _DY_EVENT:
                dy_event = torch.cuda.Event()
            with torch.cuda.stream(hs):
                dys = list(R.phase("heads", heads_phase))

        def moco_phase():
            shadow = self.queue_shadow(dt)              # maintained by enqueue: no per-step cast of the 16.8 MB queue
            dq, kn32 = E.moco_forward_backward(dt, q, k, self.queue, shadow, float(self.hparams.softmax_temperature),
                                               loss[0:1], backward=need_grad)
            dA2 = None
            if need_grad:
                Gq = G("encoder_q.")
                E.linear_bwd_w(dt, dq, yq, Gq["fc.weight"], B, wq.shape[0], wq.shape[1])
                E.colsum_add(dt, dq, B, wq.shape[0], Gq["fc.bias"])
                dyq = E.linear_bwd_x(dt, dq, wq, B, wq.shape[0], wq.shape[1], out_f32=1)
                dA2 = E.maxmean_backward(dt, dyq, argq, Hq)
            return kn32, dA2
        kn32, dA2 = R.phase("moco", moco_phase)
        # after the logits and dq GEMMs have read the queue; the new keys go into the fp32 queue and its bf16 shadow
        self._dequeue_and_enqueue(kn32, self.queue_shadow(dt) if dt != N.F32 else None)
        early_box = [None]

        def join_heads():
            """Order the main stream after the loss heads, then start what only waited for them: the all-reduce of the head
            gradients (data-parallel) or, on one rank, the SGD step of the head segment on a side stream."""
            for i, st in enumerate(streams):
                if E.ONE_STREAM:
                    continue
                if i == 0 and dy_event is not None:
                    main.wait_event(dy_event)               # the data gradients are there; the weight gradients may still run
                else:
                    main.wait_stream(st)
            for d in dys:
                if d is not None:
                    d.record_stream(main)
            if not need_grad:
                return dys
            self.reduce_begin("heads")                      # p1-p3 gradients complete: their all-reduce overlaps the encoder bwd
            if optimizer is not None and not ddp and hasattr(optimizer, "step_tail"):
                # single rank: the head parameters can be updated right now, on a side stream under the encoder backward
                # (nothing reads the fp32 head weights or their gradients again in this step)
                early = streams[0] if _SGD_ASIDE else main
                if not E.ONE_STREAM and _SGD_ASIDE and dy_event is None:
                    E.fork(early, main)
                with torch.cuda.stream(early):
                    if R.phase("sgd_heads", lambda: optimizer.step_tail(flat, self.head_offset(), stored=stored_grads,
                                                                        stepped=sgd_ctx is not None)):
                        optimizer.mark_early(flat, self.head_offset())
                        early_box[0] = early
                    elif sgd_ctx is not None:
                        raise RuntimeError("the projector weights were stepped by their weight-gradient GEMMs but the head-segment "
                                           "SGD pass declined: the optimiser's step() would step them again")
            return dys

        # One rank (one captured graph): the fully connected part of the encoder backward needs only the MoCo gradient, so it is
        # issued BEFORE the main stream joins the heads - the heads (0.9 ms on their stream, GEMMs that fill 40-75 % of the CUs)
        # then overlap MoCo AND that part instead of MoCo alone.  Data-parallel phases keep the join between the phases.
        late = need_grad and not ddp and _LATE_JOIN
        if not late:
            join_heads()
        if need_grad:
            def backward_phase():
                if late:
                    # every loss term is final once the main stream has joined the heads: the sum is taken there, in front of the
                    # convolution backward, not as one more small launch (+ its gaps) at the serial end of the step
                    box = [None]

                    def join_and_sum():
                        d = join_heads()
                        box[0] = loss.sum()
                        return d
                    E.encoder_backward(cq, G("encoder_q.encoder."), dA2=dA2, dx_late=join_and_sum, grads_zero=_GRADS_ZERO)
                    return box[0]
                E.encoder_backward(cq, G("encoder_q.encoder."), dA2=dA2, dx1=dys[0], dx2=dys[1], dx3=dys[2],
                                   grads_zero=_GRADS_ZERO)
                return loss.sum()
            total = R.phase("encoder_bwd", backward_phase)
            if early_box[0] is not None and not E.ONE_STREAM and _SGD_ASIDE:
                main.wait_stream(early_box[0])
            elif dy_event is not None:
                # main has only waited for the event recorded BEFORE the heads' deferred weight-gradient GEMMs; without an early
                # head-segment SGD on their stream (eager training_step, optimiser without step_tail, first graph step: no
                # momentum yet) nothing else joins them before publish_grads / opt.step() read p1-p3's gradients
                main.wait_stream(streams[0])
            self.reduce_begin("enc")
        else:
            total = loss.sum()
        if parts is not None:
            parts["losses"] = loss if loss.numel() == 4 else torch.cat([loss[:1], loss[4:].view(3, E.LOSS_REPLICAS).sum(1)])
        return total

    def _streams(self, dev):
        if E.ONE_STREAM:
            return [torch.cuda.current_stream(dev)] * 3
        if getattr(self, "_side_streams", None) is None or self._side_streams[0].device != dev:
            self._side_streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
        return self._side_streams

    def forward(self, img_q=None, img_k=None):
        raise NotImplementedError("the HIP expert fuses forward and loss; call training_step((img_1, img_2), i)")

    def training_step(self, batch, batch_idx):
        img_1, img_2 = batch
        params = self.ensure_flat().params
        loss = FusedStepFn.apply(self, torch.is_grad_enabled(), img_1, img_2, *params)
        self.log_dict({'train_loss': loss})
        return loss
