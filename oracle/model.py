"""Oracle: encoder, loss heads, experts, optimisers (test infrastructure).

torch-CPU fp32 restatement; backward comes from torch.autograd on these
formulas.  Follows (reference file:line)
  * `src/encoder/audiontt.py:37-107`                       AudioNTT2020Task6
  * `src/upstream/delores_s/upstream_encoder.py:10-30`     DELORES_S
  * `src/upstream/delores_m/upstream_encoder.py:11-36`     DELORES_M
  * `src/upstream/delores_s/upstream_expert.py:11-46`      Projection (Barlow)
  * `src/utils/utils.py:185-189`                           off_diagonal
  * `src/upstream/delores_m/upstream_expert.py:115-172, 222-278`  MoCo step
  * `extras/slicer/contrastive_loss.py:6-92`               NT-Xent / ClusterLoss
  * `extras/delores-s/multi_proc.py:4-43`                  LARS
  * `extras/delores-s/models_byol.py:92-118`               cross-GPU Barlow
Pinned by tests/golden/{encoder,barlow,moco,ntxent,step_*}.npz.

Dropout: the reference draws its mask from torch's CPU generator, which a GPU
cannot reproduce; here the mask is an explicit input (`drop_mask`, 1 = keep)
so that both sides of a parity test use the same one.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------- encoder
class AudioNTT2020Task6(nn.Module):
    def __init__(self, n_mels=64, d=2048, return_all_layers=False, p_drop=0.3):
        super().__init__()
        self.return_all_layers = return_all_layers
        self.p_drop = p_drop

        def block(cin):
            return nn.Sequential(nn.Conv2d(cin, 64, 3, stride=1, padding=1),
                                 nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(2, stride=2))
        self.features_1 = block(1)
        self.features_2 = block(64)
        self.features_3 = block(64)
        self.fc = nn.Sequential(nn.Linear(64 * (n_mels // 8), d), nn.ReLU(), nn.Dropout(p=p_drop),
                                nn.Linear(d, d), nn.ReLU())
        self.d = d

    @staticmethod
    def _tmean(x):
        x = x.permute(0, 3, 2, 1)
        B, T, D, C = x.shape
        return x.reshape(B, T, C * D)

    def forward(self, x, drop_mask=None):
        """drop_mask: None -> no dropout (eval semantics for that layer only);
        else float/bool [B, T', d] keep-mask, applied as x*mask/(1-p)."""
        x = self.features_1(x)
        x_1 = self._tmean(x).mean(dim=1)
        x = self.features_2(x)
        x_2 = self._tmean(x).mean(dim=1)
        x = self.features_3(x)
        x_3 = self._tmean(x).mean(dim=1)
        x = self._tmean(x)
        x = F.relu(self.fc[0](x))
        if drop_mask is not None:
            x = x * drop_mask.to(x.dtype) / (1.0 - self.p_drop)
        x = F.relu(self.fc[3](x))
        if self.return_all_layers:
            return x_1, x_2, x_3, x
        return x

    def __repr__(self):
        return "AudioNTT2020Task6"


def _maxmean(x):
    return x.max(dim=1).values + x.mean(dim=1)


class DELORES_S(nn.Module):
    def __init__(self, config, base_encoder=AudioNTT2020Task6):
        super().__init__()
        be = config["pretrain"]["base_encoder"]
        self.return_all_layers = be["return_all_layers"]
        self.encoder = base_encoder(config["pretrain"]["input"]["n_mels"], be["output_dim"], self.return_all_layers)

    def forward(self, x, drop_mask=None):
        x = self.encoder(x, drop_mask)
        if self.return_all_layers:
            x = x[-1]
        return _maxmean(x)


class DELORES_M(nn.Module):
    def __init__(self, config, base_encoder=AudioNTT2020Task6):
        super().__init__()
        be = config["pretrain"]["base_encoder"]
        self.encoder = base_encoder(config["pretrain"]["input"]["n_mels"], be["output_dim"], True)
        self.fc = nn.Linear(be["output_dim"], config["pretrain"]["contrastive_dim"])

    def forward(self, x, drop_mask=None):
        l1, l2, l3, x = self.encoder(x, drop_mask)
        return self.fc(_maxmean(x)), l1, l2, l3


# -------------------------------------------------------------- Barlow head
def off_diagonal(x):
    n, m = x.shape
    assert n == m
    return x.flatten()[:-1].view(n - 1, n + 1)[:, 1:].flatten()


class Projection(nn.Module):
    def __init__(self, in_dim, lambd=5e-5, scale_loss=1 / 32, width=2048):
        super().__init__()
        sizes = [in_dim, width, width, width]
        layers = []
        for i in range(len(sizes) - 2):
            layers += [nn.Linear(sizes[i], sizes[i + 1], bias=False), nn.BatchNorm1d(sizes[i + 1]), nn.ReLU()]
        layers.append(nn.Linear(sizes[-2], sizes[-1], bias=False))
        self.projector = nn.Sequential(*layers)
        self.lambd = float(lambd)
        self.scale_loss = float(scale_loss)
        self.bn = nn.BatchNorm1d(sizes[-1], affine=False)

    def correlation(self, y1, y2):
        z1 = self.projector(y1)
        z2 = self.projector(y2)
        return (self.bn(z1).T @ self.bn(z2)) / z1.shape[0]

    def loss_from_c(self, c):
        on_diag = (torch.diagonal(c) - 1).pow(2).sum() * self.scale_loss
        off_diag = off_diagonal(c).pow(2).sum() * self.scale_loss
        if self.lambd:
            return self.lambd * on_diag + self.lambd * off_diag
        return on_diag + off_diag

    def forward(self, y1, y2, all_reduce=None):
        """all_reduce: optional callable(c)->c summing c over ranks with the
        batch divisor already applied (models_byol.py:108-112 semantics: the
        reduce is not differentiated; gradient flows as identity)."""
        c = self.correlation(y1, y2)
        if all_reduce is not None:
            c = c + (all_reduce(c.detach()) - c.detach())
        return self.loss_from_c(c)


# ------------------------------------------------------------------- MoCo head
def moco_logits(q, k, queue, temperature):
    l_pos = torch.einsum('nc,nc->n', [q, k]).unsqueeze(-1)
    l_neg = torch.einsum('nc,ck->nk', [q, queue.clone().detach()])
    return torch.cat([l_pos, l_neg], dim=1) / temperature


def moco_enqueue(queue, ptr, keys):
    """delores_m/upstream_expert.py:156-172; returns new ptr."""
    b = keys.shape[0]
    K = queue.shape[1]
    assert K % b == 0
    queue[:, ptr:ptr + b] = keys.T
    return (ptr + b) % K


# ------------------------------------------------------------ NT-Xent / cluster
def nt_xent(z_i, z_j, temperature):
    """InstanceLoss.forward, contrastive_loss.py:26-42."""
    B = z_i.shape[0]
    N = 2 * B
    z = torch.cat((z_i, z_j), dim=0)
    sim = (z @ z.T) / temperature
    pos = torch.cat((torch.diag(sim, B), torch.diag(sim, -B)), dim=0).reshape(N, 1)
    mask = torch.ones(N, N, dtype=torch.bool)
    mask.fill_diagonal_(False)
    for i in range(B):
        mask[i, B + i] = False
        mask[B + i, i] = False
    neg = sim[mask].reshape(N, -1)
    logits = torch.cat((pos, neg), dim=1)
    return F.cross_entropy(logits, torch.zeros(N, dtype=torch.long), reduction="sum") / N


def cluster_loss(c_i, c_j, temperature):
    """ClusterLoss.forward, contrastive_loss.py:66-92 (entropy term computed by the
    reference but not returned)."""
    K = c_i.shape[1]
    ci, cj = c_i.t(), c_j.t()
    N = 2 * K
    c = torch.cat((ci, cj), dim=0)
    sim = F.cosine_similarity(c.unsqueeze(1), c.unsqueeze(0), dim=2) / temperature
    pos = torch.cat((torch.diag(sim, K), torch.diag(sim, -K)), dim=0).reshape(N, 1)
    mask = torch.ones(N, N, dtype=torch.bool)
    mask.fill_diagonal_(False)
    for i in range(K):
        mask[i, K + i] = False
        mask[K + i, i] = False
    neg = sim[mask].reshape(N, -1)
    logits = torch.cat((pos, neg), dim=1)
    return F.cross_entropy(logits, torch.zeros(N, dtype=torch.long), reduction="sum") / N


# ------------------------------------------------------------------- experts
class DeloresSExpert(nn.Module):
    """`src/upstream/delores_s/upstream_expert.py:52-243` (lambda coerced to float)."""

    def __init__(self, config, learning_rate=0.03, momentum=0.9, weight_decay=1e-4):
        super().__init__()
        self.encoder = DELORES_S(config)
        self.p = Projection(config["pretrain"]["projection_dim"], float(config["pretrain"]["lambda_barlow"]))
        self.hp = dict(lr=learning_rate, momentum=momentum, weight_decay=weight_decay)

    def training_loss(self, img_1, img_2, mask_1=None, mask_2=None):
        q = self.encoder(img_1, mask_1)
        k = self.encoder(img_2, mask_2)
        return self.p(q, k)


class DeloresMExpert(nn.Module):
    """`src/upstream/delores_m/upstream_expert.py:51-317`, single process."""

    def __init__(self, config, emb_dim=128, num_negatives=65536, encoder_momentum=0.999,
                 softmax_temperature=0.07, learning_rate=0.03, momentum=0.9, weight_decay=1e-4):
        super().__init__()
        self.encoder_q = DELORES_M(config)
        self.encoder_k = DELORES_M(config)
        for pq, pk in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            pk.data.copy_(pq.data)
            pk.requires_grad = False
        self.register_buffer("queue", F.normalize(torch.randn(emb_dim, num_negatives), dim=0))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        lam = config["pretrain"]["lambda_barlow"]
        s = config["pretrain"]["loss_scale"]
        s = eval(s) if isinstance(s, str) else s
        self.p1 = Projection(2048, lam[0], s)
        self.p2 = Projection(1024, lam[1], s)
        self.p3 = Projection(512, lam[2], s)
        self.m = encoder_momentum
        self.T = softmax_temperature
        self.K = num_negatives
        self.hp = dict(lr=learning_rate, momentum=momentum, weight_decay=weight_decay)

    @torch.no_grad()
    def momentum_update(self):
        for pq, pk in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            pk.data = pk.data * self.m + pq.data * (1.0 - self.m)

    def training_loss(self, img_1, img_2, mask_q=None, mask_k=None, parts=None):
        q, q1, q2, q3 = self.encoder_q(img_1, mask_q)
        q = F.normalize(q, dim=1)
        with torch.no_grad():
            self.momentum_update()
            k, k1, k2, k3 = self.encoder_k(img_2, mask_k)
            k = F.normalize(k, dim=1)
        logits = moco_logits(q, k, self.queue, self.T)
        with torch.no_grad():
            self.queue_ptr[0] = moco_enqueue(self.queue, int(self.queue_ptr), k)
        ce = F.cross_entropy(logits.float(), torch.zeros(logits.shape[0], dtype=torch.long))
        b1, b2, b3 = self.p1(q1, k1), self.p2(q2, k2), self.p3(q3, k3)
        if parts is not None:
            parts.update(ce=ce.detach(), b1=b1.detach(), b2=b2.detach(), b3=b3.detach(),
                         q=q.detach(), k=k.detach(), logits0=logits[0].detach())
        return ce + b1 + b2 + b3


class SLICER(nn.Module):
    """`src/upstream/slicer/upstream_encoder.py:4-35`: encoder -> max_T + mean_T -> instance Linear and cluster MLP+Softmax."""

    def __init__(self, config, base_encoder=AudioNTT2020Task6):
        super().__init__()
        pre = config["pretrain"]
        d = pre["base_encoder"]["output_dim"]
        self.encoder = base_encoder(pre["input"]["n_mels"], d, pre["base_encoder"]["return_all_layers"])
        self.instance_projector = nn.Linear(d, pre["instance_contrastive_dim"])
        self.cluster_projector = nn.Sequential(nn.Linear(d, d), nn.ReLU(), nn.Linear(d, pre["cluster_contrastive_dim"]),
                                               nn.Softmax(dim=1))

    def forward(self, x, drop_mask=None):
        x = _maxmean(self.encoder(x, drop_mask))
        return self.instance_projector(x), self.cluster_projector(x)


class SlicerExpert(nn.Module):
    """`src/upstream/slicer/upstream_expert.py:13-276`, single process: symmetric MoCo InfoNCE (two forward calls with the
    views swapped; each call updates the key encoder and enqueues its keys) + ClusterLoss(class_num, temperature 1) on
    the query-side soft assignments of the two calls.  The shipped training_step returns the first call's CE only
    (`:237`); `parts` carries every term it logs, the return value is the logged `train_loss`."""

    def __init__(self, config, emb_dim=128, num_negatives=65536, encoder_momentum=0.999, softmax_temperature=0.07,
                 learning_rate=0.03, momentum=0.9, weight_decay=1e-4):
        super().__init__()
        self.encoder_q = SLICER(config)
        self.encoder_k = SLICER(config)
        for pq, pk in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            pk.data.copy_(pq.data)
            pk.requires_grad = False
        self.register_buffer("queue", F.normalize(torch.randn(emb_dim, num_negatives), dim=0))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        self.m, self.T, self.K = encoder_momentum, softmax_temperature, num_negatives
        self.hp = dict(lr=learning_rate, momentum=momentum, weight_decay=weight_decay)

    @torch.no_grad()
    def momentum_update(self):
        for pq, pk in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            pk.data = pk.data * self.m + pq.data * (1.0 - self.m)

    def one_direction(self, img_q, img_k, mask_q, mask_k):
        q, q_cluster = self.encoder_q(img_q, mask_q)
        q = F.normalize(q, dim=1)
        with torch.no_grad():
            self.momentum_update()
            k, _ = self.encoder_k(img_k, mask_k)
            k = F.normalize(k, dim=1)
        logits = moco_logits(q, k, self.queue, self.T)
        with torch.no_grad():
            self.queue_ptr[0] = moco_enqueue(self.queue, int(self.queue_ptr), k)
        ce = F.cross_entropy(logits.float(), torch.zeros(logits.shape[0], dtype=torch.long))
        return ce, q_cluster

    def training_loss(self, img_1, img_2, masks=(None, None, None, None), parts=None):
        """masks: dropout keep-masks in call order (q on view 1, k on view 2, q on view 2, k on view 1)."""
        ce_a, qc_a = self.one_direction(img_1, img_2, masks[0], masks[1])
        ce_b, qc_b = self.one_direction(img_2, img_1, masks[2], masks[3])
        cl = cluster_loss(qc_a, qc_b, 1.0)
        if parts is not None:
            parts.update(ce_first=ce_a.detach(), sym=(ce_a + ce_b).detach(), cluster=cl.detach())
        return ce_a + ce_b + cl


class DecarV2Model(nn.Module):
    """`extras/decar-v2/models_delores.py:33-122` (AudioNTT2020): the Task6 encoder written as ONE `features` Sequential
    (state_dict keys `features.{0,1,4,5,8,9}.*`, `fc.{0,3}.*`), max_T + mean_T, projection head
    Linear(d, 2048)-BN-ReLU-Linear(2048, out_dim), one bias-free prototype layer per k-means head (`MultiPrototypes`,
    utils.py:134-145: `prototypes.prototypes<i>.weight`).  forward(batch) -> (embedding of view 1, [scores of view 2])."""

    def __init__(self, out_dim=512, n_mels=64, d=2048, nmb_prototypes=(1024,), p_drop=0.3):
        super().__init__()
        layers = []
        for cin in (1, 64, 64):
            layers += [nn.Conv2d(cin, 64, 3, stride=1, padding=1), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(2, stride=2)]
        self.features = nn.Sequential(*layers)
        self.fc = nn.Sequential(nn.Linear(64 * (n_mels // 8), d), nn.ReLU(), nn.Dropout(p=p_drop), nn.Linear(d, d), nn.ReLU())
        self.p_drop = p_drop
        self.projection_head = nn.Sequential(nn.Linear(d, 2048), nn.BatchNorm1d(2048), nn.ReLU(inplace=True),
                                             nn.Linear(2048, out_dim))
        self.prototypes = nn.Module()
        for i, k in enumerate(nmb_prototypes):
            self.prototypes.add_module(f"prototypes{i}", nn.Linear(out_dim, k, bias=False))
        self.n_heads = len(nmb_prototypes)

    def encode(self, x, drop_mask=None):
        x = self.features(x)                                        # (batch, ch, mel, time)
        x = x.permute(0, 3, 2, 1)
        B, T, D, C = x.shape
        x = F.relu(self.fc[0](x.reshape(B, T, C * D)))
        if drop_mask is not None:
            x = x * drop_mask.to(x.dtype) / (1.0 - self.p_drop)
        return F.relu(self.fc[3](x))

    def forward(self, batch, masks=(None, None)):
        z = _maxmean(self.encode(batch[0], masks[0]))
        z_new = _maxmean(self.encode(batch[1], masks[1]))
        x = self.projection_head(z)
        x_new = self.projection_head(z_new)
        return x, [getattr(self.prototypes, f"prototypes{i}")(x_new) for i in range(self.n_heads)]


def decar_v2_loss(scores, targets):
    """`extras/decar-v2/main.py:205, 226-233`: mean over heads of CrossEntropy(ignore_index=-100)(scores / 1.0, target)."""
    loss = 0
    for h, sc in enumerate(scores):
        loss = loss + F.cross_entropy(sc / 1.0, targets[h], ignore_index=-100)
    return loss / len(scores)


# ---------------------------------------------------------------- optimisers
@torch.no_grad()
def sgd_momentum_step(params, bufs, lr, momentum, weight_decay):
    """torch.optim.SGD (no dampening, no nesterov): first step buf = g."""
    for p in params:
        if p.grad is None:
            continue
        g = p.grad + weight_decay * p if weight_decay else p.grad
        key = id(p)
        if key not in bufs:
            bufs[key] = g.clone()
        else:
            bufs[key].mul_(momentum).add_(g)
        p.add_(bufs[key], alpha=-lr)


@torch.no_grad()
def lars_step(params, bufs, lr, weight_decay=0.0, momentum=0.9, eta=0.001,
              weight_decay_filter=False, lars_adaptation_filter=False):
    """`extras/delores-s/multi_proc.py:16-43`."""
    for p in params:
        dp = p.grad
        if dp is None:
            continue
        is1d = p.ndim == 1
        if not weight_decay_filter or not is1d:
            dp = dp.add(p, alpha=weight_decay)
        if not lars_adaptation_filter or not is1d:
            pn, un = torch.norm(p), torch.norm(dp)
            one = torch.ones_like(pn)
            q = torch.where(pn > 0., torch.where(un > 0, eta * pn / un, one), one)
            dp = dp.mul(q)
        key = id(p)
        if key not in bufs:
            bufs[key] = torch.zeros_like(p)
        bufs[key].mul_(momentum).add_(dp)
        p.add_(bufs[key], alpha=-lr)


def lars_lr(step, epochs, steps_per_epoch, batch_size):
    """adjust_learning_rate, multi_proc.py:45-57 -> (lr_weights, lr_biases)."""
    max_steps = epochs * steps_per_epoch
    warmup = 10 * steps_per_epoch
    base_lr = batch_size / 256
    if step < warmup:
        lr = base_lr * step / warmup
    else:
        s, m = step - warmup, max_steps - warmup
        q = 0.5 * (1 + math.cos(math.pi * s / m))
        lr = base_lr * q + base_lr * 0.001 * (1 - q)
    return lr * 0.2, lr * 0.0048


@torch.no_grad()
def larc_sgd_step(params, bufs, lr, weight_decay=1e-6, momentum=0.9, trust_coefficient=0.001, eps=1e-8, clip=False):
    """apex.parallel.LARC.step around torch.optim.SGD, as `extras/decar-v2/main.py:92-97, 111` builds it.  apex is a third-party
    dependency that is not in the reference tree or this image: restated from its published LARC.py (parity unpinned) -
    per parameter with a gradient: if |p| != 0 and |g| != 0: alr = tc |p| / (|g| + |p| wd + eps) [clip: min(alr / lr, 1)],
    g <- (g + wd p) alr; then SGD(momentum, weight_decay = 0)."""
    for p in params:
        if p.grad is None:
            continue
        g = p.grad
        pn, gn = torch.norm(p), torch.norm(g)
        if pn != 0 and gn != 0:
            alr = trust_coefficient * pn / (gn + pn * weight_decay + eps)
            if clip:
                alr = min(alr / lr, 1)
            g = (g + weight_decay * p) * alr
        key = id(p)
        if key not in bufs:
            bufs[key] = g.clone()
        else:
            bufs[key].mul_(momentum).add_(g)
        p.add_(bufs[key], alpha=-lr)


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """`extras/decar-v2/multi_proc.py:61-72`."""
    import numpy as np
    warmup_schedule = np.array([])
    warmup_iters = warmup_epochs * niter_per_ep
    if warmup_epochs > 0:
        warmup_schedule = np.linspace(start_warmup_value, base_value, warmup_iters)
    iters = np.arange(epochs * niter_per_ep - warmup_iters)
    schedule = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * iters / len(iters)))
    schedule = np.concatenate((warmup_schedule, schedule))
    assert len(schedule) == epochs * niter_per_ep
    return schedule


def dcv2_lr_schedule(base_lr, final_lr, epochs, niter_per_ep):
    """`extras/decar-v2/main.py:118-122`: 10 warm-up epochs (linear from 0) then cosine to final_lr."""
    import numpy as np
    warm = np.linspace(0, base_lr, niter_per_ep * 10)
    iters = np.arange(niter_per_ep * (epochs - 10))
    cos = np.array([final_lr + 0.5 * (base_lr - final_lr) * (1 + math.cos(math.pi * t / (niter_per_ep * (epochs - 10)))) for t in iters])
    return np.concatenate((warm, cos))


def train_steps(expert, batches, masks=None):
    """Run fwd/bwd/SGD over a list of (img_1, img_2); returns per-step losses."""
    bufs, losses = {}, []
    params = [p for p in expert.parameters() if p.requires_grad]
    for s, (a, b) in enumerate(batches):
        for p in params:
            p.grad = None
        mk = masks[s] if masks is not None else (None, None)
        loss = expert.training_loss(a, b, mk[0], mk[1])
        loss.backward()
        sgd_momentum_step(params, bufs, expert.hp["lr"], expert.hp["momentum"], expert.hp["weight_decay"])
        losses.append(float(loss.detach()))
    return losses
