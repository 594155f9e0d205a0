"""Audio ingest for the MI355X path: decode on the host, resample to the training rate on the GPU.

The reference reads every file with `librosa.core.load(path, sr=16000)` (`src/dataset/upstream_dataset.py:55`): soundfile /
audioread decode to float32, channel mean, then `resampy.resample(y, sr_orig, 16000, filter='kaiser_best')` and
`librosa.util.fix_length` to ceil(n * ratio) samples.  librosa 0.8.1 and resampy 0.2.2 are not in the reference tree nor in
this image: the resampler is restated from resampy's published algorithm and filter parameters (parity unpinned) -
`kaiser_best` = 64 zero crossings x 512 table entries, Kaiser beta 14.769656459379492, roll-off 0.9475937167399596.
Host part: the filter table (float64, built once) and, per (input length, rate pair), the positions of every output sample from
resampy's sequentially accumulated float64 time register; device part: `audiossl_resample_sinc`, one thread per output sample.
Decoding stays `scipy.io.wavfile` (PCM WAV) / `.npy`."""
import math

import numpy as np
import torch

from src import _native as N

KAISER_BEST = dict(num_zeros=64, precision=9, beta=14.769656459379492, rolloff=0.9475937167399596)
_FILTERS, _PLANS = {}, {}


def sinc_window(num_zeros, precision, beta, rolloff):
    """resampy.filters.sinc_window with a Kaiser taper: right half of the interpolation filter, (table, entries per zero)."""
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def positions(n_orig, sample_ratio, nwin, num_table):
    """Per output sample: n, (offset, eta) of the left and right wing - `resampy.interpn.resample_f`, time register included."""
    n_out = int(n_orig * sample_ratio)
    scale = min(1.0, sample_ratio)
    inc = np.full(n_out, 1.0 / sample_ratio)
    inc[0] = 0.0
    time_register = np.add.accumulate(inc)                  # 0, inc, inc + inc, ... (sequential float64 additions)
    n = time_register.astype(np.int64)
    frac = scale * (time_register - n)
    index_frac = frac * num_table
    off_l = index_frac.astype(np.int64)
    eta_l = index_frac - off_l
    frac = scale - frac
    index_frac = frac * num_table
    off_r = index_frac.astype(np.int64)
    eta_r = index_frac - off_r
    return n_out, n.astype(np.int32), off_l.astype(np.int32), off_r.astype(np.int32), eta_l, eta_r, int(scale * num_table)


def _filter(name, sample_ratio, device):
    key = (name, round(sample_ratio, 12) if sample_ratio < 1 else 1.0, str(device))
    if key not in _FILTERS:
        if name != "kaiser_best":
            raise NotImplementedError("only librosa's default res_type 'kaiser_best' is built")
        win, num_table = sinc_window(**KAISER_BEST)
        if sample_ratio < 1:
            win = win * sample_ratio
        delta = np.zeros_like(win)
        delta[:-1] = np.diff(win)
        _FILTERS[key] = (torch.from_numpy(win).to(device), torch.from_numpy(delta).to(device), num_table)
    return _FILTERS[key]


@torch.no_grad()
def resample(wave, sr_orig, sr_new, res_type="kaiser_best", fix=True):
    """wave [n] or [clips, n] float32 -> resampled on the GPU, [..., ceil(n * sr_new / sr_orig)] like librosa.resample (fix=True
    pads / trims resampy's int(n * ratio) samples to that length)."""
    wave = torch.as_tensor(wave, dtype=torch.float32)
    squeeze = wave.dim() == 1
    x = (wave if wave.is_cuda else wave.cuda()).reshape(-1, wave.shape[-1]).contiguous()
    if sr_orig == sr_new:
        return x[0] if squeeze else x
    ratio = float(sr_new) / sr_orig
    clips, n_orig = x.shape
    win, delta, num_table = _filter(res_type, ratio, x.device)
    key = (n_orig, sr_orig, sr_new, res_type, str(x.device))
    if key not in _PLANS:
        n_out, n, ol, orr, el, er, step = positions(n_orig, ratio, win.numel(), num_table)
        _PLANS.clear() if len(_PLANS) > 64 else None
        _PLANS[key] = (n_out, step) + tuple(torch.from_numpy(a).to(x.device) for a in (n, ol, orr, el, er))
    n_out, step, n_d, ol_d, or_d, el_d, er_d = _PLANS[key]
    if n_out < 1:
        raise ValueError("Input signal length is too small to resample")
    y = torch.empty(clips, n_out, dtype=torch.float32, device=x.device)
    N.call("resample_sinc", x, y, clips, n_orig, n_out, n_d, ol_d, or_d, el_d, er_d, win, delta, win.numel(), step)
    if fix:
        want = int(math.ceil(n_orig * ratio))
        if want != n_out:
            y = torch.nn.functional.pad(y, (0, want - n_out)) if want > n_out else y[:, :want]
    return y[0] if squeeze else y


def decode(path):
    """-> (float32 mono numpy array, native sample rate); PCM WAV through scipy, `.npy` arrays are taken as 16 kHz."""
    if str(path).endswith(".npy"):
        return np.load(path, allow_pickle=False).astype(np.float32), 16000
    from scipy.io import wavfile
    rate, data = wavfile.read(path)
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    elif data.dtype.kind == "u":
        data = (data.astype(np.float32) - 128.0) / 128.0
    data = data.astype(np.float32)
    if data.ndim > 1:
        data = data.mean(axis=1)                           # librosa.to_mono
    return data, int(rate)
