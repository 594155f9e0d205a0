"""SLICER expert on MI355X: symmetric MoCo-v2 InfoNCE (two forward calls with the views swapped; each updates the key
encoder and enqueues its keys) plus the cluster-level contrastive loss on the query-side soft assignments.

Same class name, constructor, buffers and parameter names as `src/upstream/slicer/upstream_expert.py:13-276` of the
reference.  Two defects of the shipped plugin are handled as SURVEY 2.4 lists: it imports a `ClusterLoss` that
`src.utils` does not define (taken from `extras/slicer/contrastive_loss.py:45-92`, here `src.upstream.slicer.losses`),
and its `training_step` returns the first direction's cross-entropy only (`:237`) - this one returns the total it logs
as `train_loss` (`returns_logged_total=False` restores the shipped behaviour).
The step is composed from autograd bridges over the C ABI (encoder launch sequences, MFMA GEMMs, InfoNCE, NT-Xent);
gradients accumulate straight into the flat gradient buffer the fused SGD launch consumes.
"""
import torch
import torch.nn as nn

from src import _native as N
from src.encoder.audiontt import default_precision
from src.functional import MocoCEFn
from src.module_base import UpstreamModule
from src.upstream.common import FusedExpertMixin, MocoQueueMixin, _world
from src.upstream.slicer.losses import ClusterLoss
from src.upstream.slicer.upstream_encoder import SLICER as SLICER_ENCODER


class Upstream_Expert(MocoQueueMixin, FusedExpertMixin, UpstreamModule):
    def __init__(self, config, base_encoder, emb_dim: int = 128, num_negatives: int = 65536,
                 encoder_momentum: float = 0.999, softmax_temperature: float = 0.07, learning_rate: float = 0.03,
                 momentum: float = 0.9, weight_decay: float = 1e-4, data_dir: str = './', batch_size: int = 256,
                 use_mlp: bool = False, num_workers: int = 8, returns_logged_total: bool = True, *args, **kwargs):
        super().__init__()
        self.save_hyperparameters()
        if use_mlp:
            raise NotImplementedError("use_mlp is not part of the HIP path")
        self.config = config
        self.base_encoder = base_encoder
        self.encoder_q, self.encoder_k = self.init_encoders(self.base_encoder)
        for param_q, param_k in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            param_k.data.copy_(param_q.data)
            param_k.requires_grad = False
        self.register_buffer("queue", nn.functional.normalize(torch.randn(emb_dim, num_negatives), dim=0))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        self.loss_cluster = ClusterLoss(self.config["pretrain"]["cluster_contrastive_dim"], 1, None)
        prec = config.get("run", {}).get("precision")
        self.precision = {"fp32": N.F32, "bf16": N.BF16, "bf16_hp": N.BF16}.get(prec, default_precision())
        self.encoder_q.encoder.precision = self.encoder_k.encoder.precision = self.precision

    def init_encoders(self, base_encoder):
        return SLICER_ENCODER(self.config, base_encoder), SLICER_ENCODER(self.config, base_encoder)

    def forward(self, img_q=None, img_k=None):
        """-> (InfoNCE of this direction, q_cluster, k_cluster); the reference returns the logits and labels and takes
        the cross-entropy in training_step - here the [B, 1+K] logits never leave the fused InfoNCE launch sequence."""
        ddp = _world() > 1
        q_instance, q_cluster = self.encoder_q(img_q.float().contiguous())
        with torch.no_grad():
            self._momentum_update_key_encoder()
            img_k = img_k.float().contiguous()
            idx_unshuffle = None
            if ddp:
                img_k, idx_unshuffle = self._batch_shuffle_ddp(img_k)
            k_instance, k_cluster = self.encoder_k(img_k)
            if ddp:
                k_instance = self._batch_unshuffle_ddp(k_instance.contiguous(), idx_unshuffle)
        ce, kn32 = MocoCEFn.apply(q_instance, k_instance, self.queue, float(self.hparams.softmax_temperature), self.precision)
        self._dequeue_and_enqueue(kn32, None)
        return ce, q_cluster, k_cluster

    def training_step(self, batch, batch_idx):
        img_1, img_2 = batch
        flat = self.ensure_flat()
        if torch.is_grad_enabled():
            flat.zero_grad()
            flat.attach_grads()                 # p.grad = views of the flat buffer: autograd accumulates in place
        loss, q_cluster, _ = self(img_q=img_1, img_k=img_2)
        loss_1, q_cluster_1, _ = self(img_q=img_2, img_k=img_1)
        sym_loss_instance = loss + loss_1
        loss_cluster = self.loss_cluster(q_cluster, q_cluster_1)
        loss_combine = sym_loss_instance + loss_cluster
        self.log_dict({'train_loss': loss_combine, 'sym_instance_loss': sym_loss_instance, 'train_loss_cluster': loss_cluster})
        return loss_combine if self.hparams.get("returns_logged_total", True) else loss
