"""Host side of the front end and the small helpers of the reference's `src/utils/utils.py`.

Same names and call signatures as the reference; the arithmetic runs in libaudiossl_hip.so:
  MelSpectrogramLibrosa        utils.py:20-28   -> audiossl_logmel_fwd (batched, one launch per batch)
  extract_log_mel_spectrogram  utils.py:43-49
  extract_window               utils.py:166-182 (host index math; python `random`, same draw order)
  off_diagonal, loss_fn_mse, concat_all_gather, freeze_encoder, load_pretrained_encoder, AverageMeter, Metric
The filterbank / window / twiddle tables are host constants (librosa 0.8.1's published formulas: Slaney
mel scale + area normalisation, periodic Hann) uploaded once per device.
"""
import importlib
import logging
import random

import numpy as np
import torch
import torch.nn.functional as F

from src import _native as N

F32_EPS = float(np.finfo(np.float32).eps)
F64_EPS = float(np.finfo(np.float64).eps)


# ------------------------------------------------------------------------------------- host tables
def _hz_to_mel(f):
    f = np.atleast_1d(np.asarray(f, dtype=np.float64))
    f_sp, brk = 200.0 / 3, 1000.0
    out = f / f_sp
    step = np.log(6.4) / 27.0
    hi = f >= brk
    out[hi] = brk / f_sp + np.log(f[hi] / brk) / step
    return out


def _mel_to_hz(m):
    m = np.atleast_1d(np.asarray(m, dtype=np.float64))
    f_sp, brk = 200.0 / 3, 1000.0
    out = f_sp * m
    step = np.log(6.4) / 27.0
    hi = m >= brk / f_sp
    out[hi] = brk * np.exp(step * (m[hi] - brk / f_sp))
    return out


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """float32 [n_mels, 1+n_fft/2] triangular filters, Slaney scale and area norm."""
    bins = 1 + n_fft // 2
    freqs = np.linspace(0.0, sr / 2.0, bins)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin)[0], _hz_to_mel(fmax)[0], n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    fb = np.zeros((n_mels, bins), dtype=np.float32)
    for i in range(n_mels):
        fb[i] = np.maximum(0.0, np.minimum(-ramps[i] / width[i], ramps[i + 2] / width[i + 1]))
    norm = 2.0 / (edges[2:] - edges[:-2])
    return (fb.astype(np.float64) * norm[:, None]).astype(np.float32)


def pack_filterbank(fb):
    """Each row's non-zero run as (start, values padded to a common length)."""
    n_mels = fb.shape[0]
    starts = np.zeros(n_mels, np.int32)
    runs = []
    for m in range(n_mels):
        nz = np.nonzero(fb[m])[0]
        if len(nz) == 0:
            runs.append(np.zeros(1, np.float32))
            continue
        starts[m] = nz[0]
        runs.append(fb[m, nz[0]:nz[-1] + 1])
    taps = max(len(r) for r in runs)
    packed = np.zeros((n_mels, taps), np.float32)
    for m, r in enumerate(runs):
        packed[m, :len(r)] = r
    return starts, packed


class MelSpectrogramLibrosa:
    """Mel spectrogram with the reference's constructor; computed on the GPU.

    `__call__(audio[L]) -> [n_mels, T]` keeps the reference's contract (mel *power*); `logmel(wave[B, L])`
    is the batched, log-fused entry point the training path uses."""

    def __init__(self, fs=16000, n_fft=1024, shift=160, n_mels=64, fmin=60, fmax=7800):
        if n_fft != 1024:
            raise NotImplementedError("the HIP front end is specialised for n_fft=1024 (the reference's setting)")
        self.fs, self.n_fft, self.shift, self.n_mels, self.fmin, self.fmax = fs, n_fft, shift, n_mels, fmin, fmax
        self.mfb = slaney_mel_filterbank(fs, n_fft, n_mels, fmin, fmax)
        self._starts, self._packed = pack_filterbank(self.mfb)
        k = np.arange(n_fft, dtype=np.float64)
        self._win = (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n_fft)).astype(np.float32)
        ang = -2.0 * np.pi * k / n_fft
        self._tw = np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)
        self._dev = {}

    def _tables(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(torch.from_numpy(a).to(device) for a in (self._win, self._tw, self._packed, self._starts))
        return self._dev[key]

    def n_frames(self, n_samples):
        return 1 + n_samples // self.shift

    def _run(self, wave, apply_log):
        wave = torch.as_tensor(wave, dtype=torch.float32)
        squeeze = wave.dim() == 1
        src_device = wave.device
        if not wave.is_cuda:
            wave = wave.cuda()
        wave = wave.reshape(-1, wave.shape[-1]).contiguous()
        B, L = wave.shape
        T = self.n_frames(L)
        out = torch.empty(B, self.n_mels, T, dtype=torch.float32, device=wave.device)
        win, tw, packed, starts = self._tables(wave.device)
        N.call("logmel_fwd", wave, out, B, L, T, self.n_fft, self.shift, self.n_mels, packed.shape[1], win, tw, packed,
               starts, np.float32(F64_EPS).item(), F32_EPS, int(apply_log))
        out = out[0] if squeeze else out
        return out if src_device.type == "cuda" else out.to(src_device)

    def __call__(self, audio):
        return self._run(audio, False)

    def logmel(self, wave):
        """[B, L] (or [L]) float32 waveform -> log(mel + eps) [B, n_mels, T]."""
        return self._run(wave, True)


def extract_log_mel_spectrogram(waveform, to_mel_spec):
    """waveform [L] or [B, L] -> log-mel, one fused launch."""
    return to_mel_spec.logmel(waveform)


# ------------------------------------------------------------------------------------- windows
def window_start(n_samples, unit_length):
    """(start, left_pad) of the reference's random crop; draws from python `random` only when the clip is
    longer than the window, exactly like `extract_window`."""
    short = unit_length - n_samples
    left = short // 2 if short > 0 else 0
    over = max(n_samples, unit_length) - unit_length
    start = random.randint(0, over) if over > 0 else 0
    return start, left


def extract_window(wav, duration=16000, data_size=None):
    """Random (or centre zero-padded) window of `data_size` seconds / `duration` samples."""
    unit_length = int(data_size * 16000) if data_size else duration
    start, left = window_start(len(wav), unit_length)
    if left or len(wav) < unit_length:
        wav = F.pad(wav, (left, unit_length - len(wav) - left))
    return wav[start:start + unit_length]


# ------------------------------------------------------------------------------------- small helpers
def off_diagonal(x):
    n, m = x.shape
    assert n == m
    return x.flatten()[:-1].view(n - 1, n + 1)[:, 1:].flatten()


def loss_fn_mse(x, y):
    x = F.normalize(x, dim=-1, p=2)
    y = F.normalize(y, dim=-1, p=2)
    return (2 - 2 * (x * y).sum(dim=-1)).mean()


@torch.no_grad()
def concat_all_gather(tensor):
    """All-gather along dim 0 into one pre-allocated buffer (no list + cat copy); no gradient."""
    world = torch.distributed.get_world_size()
    out = torch.empty((world * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
    torch.distributed.all_gather_into_tensor(out, tensor.contiguous())
    return out


def freeze_encoder(model):
    logging.getLogger("__main__").info("freezing encoder weights")
    for param in model.encoder.parameters():
        param.requires_grad = False


def load_pretrained_encoder(model, args):
    expert_cls = getattr(importlib.import_module(f"src.upstream.{args.upstream}.upstream_expert"), "Upstream_Expert")
    backbone = expert_cls.load_from_checkpoint(args.checkpoint, strict=False)
    wts = backbone.encoder_q.state_dict()
    target = model.module if hasattr(model, "module") else model
    missing, unexpected = target.load_state_dict(wts, strict=False)
    print("Missing Keys:  ", missing)
    print("Unexpected Keys:  ", unexpected)
    return model


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class Metric(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val):
        if isinstance(val, torch.Tensor):
            val = val.detach().cpu().numpy()
        self.val = val
        self.sum += np.sum(val)
        self.count += np.size(val)
        self.avg = self.sum / self.count
