"""SS-MAST expert on MI355X (BASELINE config 4): MoCo-v3-style symmetric InfoNCE on the AST transformer.

Follows `Moco_v2` of `src/upstream/ssmast/upstream_expert.py:62-379` (= `extras/mast_new/mast/moco_model.py`) of the reference:
`training_step` runs the model twice with the views swapped (each call: query encoder, EMA of the key encoder with the
cosine momentum schedule `adjust_moco_momentum(epoch + 1)` of `extras/mast_new/mast/utils.py:55-57`, key encoder,
InfoNCE against the queue, enqueue), the loss is the sum of the two cross-entropies, the optimiser is AdamW.
The shipped plugin imports modules that are not in the reference tree, so the class keeps its constructor
keywords and method names but takes the usual `(config, base_encoder)` pair of the `src/upstream` plugins.

One fused launch sequence per step: both query passes keep their activations, both InfoNCE heads produce dq on the spot,
the two transformer backwards accumulate straight into the flat gradient buffer, AdamW is one launch over it.
"""
import math

import torch
import torch.nn as nn

from src import _native as N
from src import engine as E
from src.module_base import UpstreamModule
from src.upstream.common import EAGER, FusedExpertMixin, FusedStepFn, MocoQueueMixin, _world
from src.upstream.ssmast.upstream_encoder import SSMAST


def adjust_moco_momentum(epoch, epochs=200, base=0.99):
    return 1. - 0.5 * (1. + math.cos(math.pi * epoch / epochs)) * (1. - base)


class Upstream_Expert(MocoQueueMixin, FusedExpertMixin, UpstreamModule):
    def __init__(self, config, base_encoder=None, emb_dim: int = 256, num_negatives: int = 65536,
                 encoder_momentum: float = 0.999, softmax_temperature: float = 0.07, learning_rate: float = 0.0003,
                 momentum: float = 0.9, weight_decay: float = 0, data_dir: str = './', batch_size: int = 256,
                 use_mlp: bool = False, num_workers: int = 8, *args, **kwargs):
        super().__init__()
        self.save_hyperparameters()
        if use_mlp:
            raise NotImplementedError("use_mlp is not part of the HIP path")
        self.config = config
        self.base_encoder = base_encoder
        self.encoder_q, self.encoder_k = self.init_encoders(base_encoder)
        for param_q, param_k in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            param_k.data.copy_(param_q.data)
            param_k.requires_grad = False
        self.register_buffer("queue", nn.functional.normalize(torch.randn(emb_dim, num_negatives), dim=0))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        self.precision = N.BF16                   # the transformer path is bf16 MFMA with an fp32 residual stream
        self.current_epoch = 0

    def init_encoders(self, base_encoder):
        return SSMAST(self.config, self.hparams.emb_dim), SSMAST(self.config, self.hparams.emb_dim)

    # ---- MoCo pieces ------------------------------------------------------------------------------------------------
    def _epoch(self):
        return int(getattr(self.trainer, "current_epoch", self.current_epoch))

    @torch.no_grad()
    def _momentum_update_key_encoder(self, epoch=None):
        self.ensure_flat()
        em = adjust_moco_momentum((self._epoch() if epoch is None else epoch) + 1)
        N.call("ema_update", self.flat_k.data, self.flat.data, self.flat_k.numel, float(em), None)

    def graph_key(self):
        return (self._epoch(),)                   # the EMA momentum is a launch argument: a new epoch re-captures the graph

    # ---- fused step ----------------------------------------------------------------------------------------------------
    def fused_loss(self, img_1, img_2, need_grad=True, runner=None):
        """Both directions of `Moco_v2.training_step` (`extras/mast_new/mast/moco_model.py:253-340`).  The work is written as
        collective-free phases handed to `runner` (query forward, key EMA + forward, InfoNCE, enqueue per direction, then the two
        backwards): eager issue runs them in place, the data-parallel graph step captures each into its own hipGraph
        (`common.GraphPhases`) with the collectives - batch shuffle / unshuffle, key gather, gradient all-reduce - in between."""
        R = runner or EAGER
        flat = self.ensure_flat()
        ddp = _world() > 1
        eq, ek = self.encoder_q, self.encoder_k
        T = float(self.hparams.softmax_temperature)
        dev = img_1.device
        img_1, img_2 = img_1.float().contiguous(), img_2.float().contiguous()
        Pq = eq.param_dict()
        G = flat.grad_dict("encoder_q.")

        def prep_phase():
            if need_grad:
                flat.zero_grad()
            flat.refresh_shadow(N.BF16)               # one cast launch: bf16 copies of every query-encoder weight
            return torch.zeros(2, dtype=torch.float32, device=dev)
        loss = R.phase("prep", prep_phase)
        Wq = flat.shadow_dict("encoder_q.")
        shadow = self.queue_shadow(N.BF16)            # bf16 copy of the queue, kept current by `enqueue` (no per-pass cast of 65,536 keys)
        ctxs, dqs = [], []
        for d, (xq, xk) in enumerate(((img_1, img_2), (img_2, img_1))):
            q, c = R.phase(f"query{d}", lambda xq=xq: eq.engine_forward(Pq, Wq, xq, eq.cfg, need_ctx=need_grad))
            idx_unshuffle = None
            if ddp:
                xk, idx_unshuffle = self._batch_shuffle_ddp(xk)
                xk = R.static(f"key_in{d}", xk.contiguous())

            def key_phase(xk=xk):
                self._momentum_update_key_encoder()
                self.flat_k.refresh_shadow(N.BF16)    # the key weights just moved
                return ek.engine_forward(ek.param_dict(), self.flat_k.shadow_dict(), xk, ek.cfg, need_ctx=False)[0]
            k = R.phase(f"key{d}", key_phase)
            if ddp:
                k = R.static(f"key_out{d}", self._batch_unshuffle_ddp(k.contiguous(), idx_unshuffle))
            dq, kn32 = R.phase(f"moco{d}", lambda q=q, k=k, d=d: E.moco_forward_backward(
                N.BF16, q, k, self.queue, shadow, T, loss[d:d + 1], backward=need_grad))
            if ddp:
                from src.utils import concat_all_gather
                kn32 = R.static(f"keys{d}", concat_all_gather(kn32))
            R.phase(f"enqueue{d}", lambda kn32=kn32: self._enqueue_local(kn32, shadow))
            ctxs.append(c)
            dqs.append(dq)
        if need_grad:
            def backward_phase():
                for c, dq in zip(ctxs, dqs):
                    eq.engine_backward(c, Pq, Wq, G, dq.float())
                return loss.sum()
            total = R.phase("backward", backward_phase)
            self.reduce_begin("enc")
            return total
        return loss.sum()

    @torch.no_grad()
    def _enqueue_local(self, keys32, shadow):
        """`_dequeue_and_enqueue` (`moco_model.py:187-204`) after the key gather: keys32 holds the keys of EVERY rank."""
        K = self.hparams.num_negatives
        assert K % keys32.shape[0] == 0  # for simplicity
        N.call("enqueue", self.precision, keys32, keys32.shape[0], keys32.shape[1], K, 0, self.queue_ptr, self.queue, shadow)

    def graph_phases_supported(self):
        return True

    def forward(self, img_q=None, img_k=None, epoch=None):
        raise NotImplementedError("the HIP expert fuses forward and loss; call training_step((img_1, img_2), i)")

    def training_step(self, batch, batch_idx):
        img_1, img_2 = batch
        params = self.ensure_flat().params
        loss = FusedStepFn.apply(self, torch.is_grad_enabled(), img_1, img_2, *params)
        self.log_dict({'train_loss': loss})
        return loss

    def configure_optimizers(self):
        from src.optim import HipAdamW
        self.ensure_flat()
        self.hip_optimizer = HipAdamW([self.flat], self.flat.params, lr=self.hparams.learning_rate,
                                      weight_decay=self.hparams.weight_decay)
        return self.hip_optimizer


Moco_v2 = Upstream_Expert      # the class name of the reference's plugin file
