"""DeLoRes-S expert on MI355X: one encoder, two views, Barlow-Twins style cross-correlation loss.

Same class name, constructor and methods as `src/upstream/delores_s/upstream_expert.py:52-243` of the reference
(`Upstream_Expert(config, base_encoder, datamodule=...)`, `forward`, `training_step`, `configure_optimizers`,
attribute `encoder`; `encoder_q` is an alias so `load_pretrained_encoder` works).
`training_step` runs the whole step - both encoder passes, the projector, the loss and the complete backward - as
one fused sequence of HIP launches (`src.engine`); `loss.backward()` only publishes the gradients.
"""
import torch

from src import _native as N
from src import engine as E
from src.encoder.audiontt import default_precision
from src.module_base import UpstreamModule
from src.upstream.common import EAGER, FusedExpertMixin, FusedStepFn, Projection
from src.upstream.delores_s.upstream_encoder import DELORES_S as DELORES_S_ENCODER


class Upstream_Expert(FusedExpertMixin, UpstreamModule):
    def __init__(self, config, base_encoder, datamodule=None, emb_dim: int = 128, num_negatives: int = 65536,
                 encoder_momentum: float = 0.999, softmax_temperature: float = 0.07, learning_rate: float = 0.03,
                 momentum: float = 0.9, weight_decay: float = 1e-4, data_dir: str = './', batch_size: int = 256,
                 use_mlp: bool = False, num_workers: int = 8, *args, **kwargs):
        super().__init__()
        self.save_hyperparameters()
        self.config = config
        self.base_encoder = base_encoder
        self.datamodule = datamodule
        self.encoder = self.init_encoders(self.base_encoder)
        self.p = Projection(self.config["pretrain"]["projection_dim"], self.config["pretrain"]["lambda_barlow"])
        prec = config.get("run", {}).get("precision")
        self.precision = {"fp32": N.F32, "bf16": N.BF16, "bf16_hp": N.BF16}.get(prec, default_precision())
        self.high_precision = prec == "bf16_hp"
        self.encoder.encoder.precision = self.precision
        self.cross_gpu_barlow = bool(config.get("run", {}).get("cross_gpu_barlow", False))

    def init_encoders(self, base_encoder):
        return DELORES_S_ENCODER(self.config, base_encoder)

    @property
    def encoder_q(self):
        return self.encoder

    def forward(self, img_q=None, img_k=None):
        return self.encoder(img_q), self.encoder(img_k)

    # ------------------------------------------------------------------ fused step
    def fused_loss(self, img_1, img_2, need_grad=True, runner=None):
        with E.ARENA.step(img_1.device):          # every zero-initialised scratch of the step: one arena, one memset
            return self._fused_loss(img_1, img_2, need_grad, runner)

    def _fused_loss(self, img_1, img_2, need_grad, runner):
        """Forward + backward of the step as two collective-free phases (see common.GraphPhases): both views through the
        encoder + the Barlow head, then the encoder backward; the gradient all-reduces start between / after them."""
        R = runner or EAGER
        dt = self.precision
        E.set_high_precision(self.high_precision)
        enc = self.encoder.encoder
        flat = self.ensure_flat()
        B = img_1.shape[0]
        G = flat.grad_dict("encoder.encoder.")

        def forward_phase():
            if need_grad:
                flat.zero_grad()
            flat.refresh_shadow(dt)
            loss = torch.zeros(1, dtype=torch.float32, device=img_1.device)
            P = enc.param_dict()
            Wenc = flat.shadow_dict("encoder.encoder.")
            Wp = flat.shadow_dict("p.")
            Y = torch.empty(2 * B, enc.d, dtype=E.pooled_dtype(dt), device=img_1.device)     # both views, stacked
            views = []
            for v, img in enumerate((img_1, img_2)):
                img = img.float().contiguous()
                keep = enc.next_keep_mask(B, img.shape[-1])
                _, _, _, H, c = E.encoder_forward(P, img, dt, keep=keep, p_drop=enc.fc[2].p, train=self.training,
                                                  want_layers=False, Wc=Wenc)
                _, arg = E.maxmean_forward(dt, H, out=Y[v * B:(v + 1) * B])
                views.append((c, H, arg))
            ar, gb = self._barlow_reduce()
            dY = E.barlow_forward_backward(self.p.param_dict(), flat.grad_dict("p."), Y, dt, self.p.lambd, self.p.scale_loss,
                                           loss, update_running=self.training, all_reduce=ar, global_batch=gb(B),
                                           backward=need_grad, Wc=tuple(Wp[f"projector.{i}.weight"] for i in (0, 3, 6)))
            dA = [E.maxmean_backward(dt, dY[v * B:(v + 1) * B], arg, H) for v, (c, H, arg) in enumerate(views)] if need_grad else []
            return loss, views, dA
        loss, views, dA = R.phase("forward", forward_phase)
        if need_grad:
            self.reduce_begin("heads")                      # projector gradients are complete: start their all-reduce

            def backward_phase():
                for (c, H, arg), d in zip(views, dA):
                    E.encoder_backward(c, G, dA2=d)
            R.phase("encoder_bwd", backward_phase)
            self.reduce_begin("enc")
        return loss[0]

    def graph_phases_supported(self):
        ar, _ = self._barlow_reduce()
        return ar is None and E.SYNC_BN is None             # cross-GPU Barlow / SyncBatchNorm: all-reduces inside the phases

    def _barlow_reduce(self):
        import torch.distributed as dist
        if self.cross_gpu_barlow and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return (lambda c: dist.all_reduce(c)), (lambda b: b * dist.get_world_size())
        return None, (lambda b: None)

    def training_step(self, batch, batch_idx):
        img_1, img_2 = batch
        params = self.ensure_flat().params
        loss = FusedStepFn.apply(self, torch.is_grad_enabled(), img_1, img_2, *params)
        self.log_dict({'train_loss': loss})
        return loss
