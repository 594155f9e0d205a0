"""Oracle: waveform -> random window -> log-mel (test infrastructure).

Follows (reference file:line):
  * `src/utils/utils.py:166-182`  extract_window
  * `src/utils/utils.py:20-28`    MelSpectrogramLibrosa  (calls librosa 0.8.1)
  * `src/utils/utils.py:43-49`    extract_log_mel_spectrogram

librosa==0.8.1 (requirements.txt:71) is NOT in /root/reference and not in the
image.  Its published algorithm is restated here:
  librosa.stft(y, n_fft, hop): centre reflect-pad n_fft//2, frames of n_fft at
    `hop`, periodic Hann (scipy get_window(fftbins=True), float64), float64
    rFFT of window*frame, result stored as complex64.
  librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax): Slaney mel scale
    (htk=False), triangular weights, Slaney area normalisation, float32.
PARITY UNPINNED vs librosa (no fixture of it exists in the reference); the
STFT is cross-checked against torch.stft in tests/test_oracle_frontend.py.
"""
import math
import random

import numpy as np
import torch

F32_EPS = float(np.finfo(np.float32).eps)      # torch.finfo().eps, utils.py:48
F64_EPS = float(np.finfo(np.float64).eps)      # np.finfo(float).eps, utils.py:28


# ----------------------------------------------------------------- window crop
def extract_window(wav, duration=16000, data_size=None, rng=random):
    """utils.py:166-182.  `rng` must expose randint(a, b) (inclusive)."""
    unit_length = int(data_size * 16000) if data_size else duration
    length_adj = unit_length - len(wav)
    if length_adj > 0:
        half_adj = length_adj // 2
        wav = torch.nn.functional.pad(wav, (half_adj, length_adj - half_adj))
    length_adj = len(wav) - unit_length
    start = rng.randint(0, length_adj) if length_adj > 0 else 0
    return wav[start:start + unit_length]


def window_start(n_samples, unit_length, rng=random):
    """Index-only form of extract_window: returns (start, left_pad).

    Consumes one python-`random` draw only when the clip is longer than the
    window."""
    length_adj = unit_length - n_samples
    left = length_adj // 2 if length_adj > 0 else 0
    padded = max(n_samples, unit_length)
    over = padded - unit_length
    start = rng.randint(0, over) if over > 0 else 0
    return start, left


# ------------------------------------------------------------- mel filterbank
def _hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    big = f >= min_log_hz
    out = mels.copy()
    out[big] = min_log_mel + np.log(f[big] / min_log_hz) / logstep
    return out


def _mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    big = m >= min_log_mel
    out = freqs.copy()
    out[big] = min_log_hz * np.exp(logstep * (m[big] - min_log_mel))
    return out


def mel_filterbank(sr=16000, n_fft=1024, n_mels=64, fmin=60.0, fmax=7800.0):
    """librosa.filters.mel(..., htk=False, norm='slaney', dtype=float32)."""
    n_bins = 1 + n_fft // 2
    weights = np.zeros((n_mels, n_bins), dtype=np.float32)
    fftfreqs = np.linspace(0, float(sr) / 2, n_bins, endpoint=True)
    mmin = _hz_to_mel_slaney(np.array([fmin]))[0]
    mmax = _hz_to_mel_slaney(np.array([fmax]))[0]
    mel_f = _mel_to_hz_slaney(np.linspace(mmin, mmax, n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))   # f64 -> f32 store
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    # in-place float32 *= float64 : product in f64, rounded once to f32
    weights = (weights.astype(np.float64) * enorm[:, None]).astype(np.float32)
    return weights


def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True) in float64."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


# ------------------------------------------------------------------ STFT / mel
def stft_c64(y, n_fft=1024, hop=160):
    """librosa.stft(y, n_fft, hop_length=hop) -> complex64 [1+n_fft/2, T]."""
    y = np.asarray(y, dtype=np.float32)
    pad = n_fft // 2
    yp = np.pad(y, pad, mode="reflect")
    n_frames = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    frames = yp[idx]                                       # [n_fft, T] float32
    win = hann_periodic(n_fft)[:, None]                    # float64
    spec = np.fft.rfft(win * frames, axis=0)               # float64 complex
    return spec.astype(np.complex64)


class MelSpectrogram:
    """Restatement of MelSpectrogramLibrosa (utils.py:20-28)."""

    def __init__(self, fs=16000, n_fft=1024, shift=160, n_mels=64, fmin=60, fmax=7800):
        self.fs, self.n_fft, self.shift, self.n_mels = fs, n_fft, shift, n_mels
        self.mfb = mel_filterbank(fs, n_fft, n_mels, fmin, fmax)

    def __call__(self, audio):
        X = stft_c64(np.array(audio), self.n_fft, self.shift)
        # numpy 1.20.3 (requirements.txt:94): float32 array + float64 scalar
        # stays float32 (value-based casting) -> emulate explicitly.
        power = (np.abs(X) ** 2).astype(np.float32)
        power = (power + np.float32(F64_EPS)).astype(np.float32)
        return torch.tensor(np.matmul(self.mfb, power))


def log_mel(waveform, to_mel=None):
    """extract_log_mel_spectrogram (utils.py:43-49): [L] f32 -> [n_mels, T]."""
    to_mel = to_mel or MelSpectrogram()
    return (to_mel(waveform) + torch.finfo().eps).log()


def log_mel_batch(waves, to_mel=None):
    """[B, L] -> [B, n_mels, T] (the batched entry point the product exposes)."""
    to_mel = to_mel or MelSpectrogram()
    return torch.stack([log_mel(w, to_mel) for w in waves])


def n_frames(n_samples, hop=160):
    return 1 + n_samples // hop


def l2_normalize_wave(w):
    """upstream_dataset.py:61-62 (F.normalize(p=2, dim=-1), eps 1e-12)."""
    return w / w.norm(p=2, dim=-1, keepdim=True).clamp_min(1e-12)


# ------------------------------------------------------------------ f2 audio ingest: resampy 0.2.2 kaiser_best (librosa.core.load)
def resample_kaiser_best(x, sr_orig, sr_new):
    """`resampy.resample(x, sr_orig, sr_new, filter='kaiser_best')` as `librosa.core.load(path, sr=16000)` calls it
    (src/dataset/upstream_dataset.py:55), followed by librosa's fix_length to ceil(n * ratio).  resampy / librosa are absent:
    restated from the published algorithm (interpn.resample_f: sequential float64 time register, table + linear interpolation,
    float32 output accumulated product by product) and filter parameters - parity unpinned."""
    import math
    x = np.asarray(x, np.float32)
    ratio = float(sr_new) / sr_orig
    num_zeros, num_table, beta, rolloff = 64, 512, 14.769656459379492, 0.9475937167399596
    n = num_table * num_zeros
    win = np.kaiser(2 * n + 1, beta)[n:] * (rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True)))
    if ratio < 1:
        win = win * ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    n_out = int(len(x) * ratio)
    y = np.zeros(n_out, np.float32)
    scale = min(1.0, ratio)
    inc = 1.0 / ratio
    step = int(scale * num_table)
    nwin, n_orig = len(win), len(x)
    t_reg = 0.0
    for t in range(n_out):
        nn = int(t_reg)
        frac = scale * (t_reg - nn)
        idx = frac * num_table
        off = int(idx)
        eta = idx - off
        acc = np.float32(0)
        for i in range(min(nn + 1, (nwin - off) // step)):
            acc = np.float32(acc + (win[off + i * step] + eta * delta[off + i * step]) * x[nn - i])
        frac = scale - frac
        idx = frac * num_table
        off = int(idx)
        eta = idx - off
        for k in range(min(n_orig - nn - 1, (nwin - off) // step)):
            acc = np.float32(acc + (win[off + k * step] + eta * delta[off + k * step]) * x[nn + k + 1])
        y[t] = acc
        t_reg += inc
    want = int(math.ceil(len(x) * ratio))
    return np.pad(y, (0, want - n_out)) if want > n_out else y[:want]
