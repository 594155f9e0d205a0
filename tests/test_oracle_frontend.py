"""CPU: log-mel oracle sanity (parity UNPINNED vs librosa 0.8.1 - library absent; cross-checks only)."""
import numpy as np
import torch

from oracle import fill
from oracle import frontend as FE


def test_mel_filterbank_structure():
    m = FE.mel_filterbank()
    assert m.shape == (64, 513) and m.dtype == np.float32
    assert int((m != 0).sum()) == 966
    assert int((m != 0).sum(1).max()) == 45
    cols = np.nonzero(m.any(0))[0]
    assert cols[0] == 4 and cols[-1] == 499
    # Slaney area normalisation: each triangle integrates to ~1 in Hz
    assert np.all(m >= 0)


def test_stft_matches_torch_stft():
    y = fill.uniform((16000,), 11, -0.1, 0.1)
    y += (0.3 * np.sin(2 * np.pi * 440 * np.arange(16000) / 16000)).astype(np.float32)
    X = FE.stft_c64(y)
    Xt = torch.stft(torch.tensor(y), 1024, 160, window=torch.hann_window(1024, periodic=True), center=True,
                    pad_mode="reflect", return_complex=True).numpy()
    assert X.shape == (513, 101)
    assert np.abs(X - Xt).max() <= 1e-5 * np.abs(Xt).max()
    assert FE.stft_c64(np.zeros(15200, np.float32)).shape == (513, 96)


def test_log_mel_edge_cases():
    lm0 = FE.log_mel(torch.zeros(16000))
    assert lm0.shape == (64, 101) and lm0.dtype == torch.float32
    assert torch.allclose(lm0, torch.full_like(lm0, float(np.log(np.float32(FE.F32_EPS)))), atol=1e-4)
    full = FE.log_mel(torch.ones(16000))
    assert torch.isfinite(full).all()
    b = FE.log_mel_batch(torch.from_numpy(fill.uniform((3, 16000), 5, -0.5, 0.5)))
    assert b.shape == (3, 64, 101)


def test_resample_oracle_sanity():
    """oracle.frontend.resample_kaiser_best (resampy 0.2.2 restated, parity unpinned): length = ceil(n * ratio) as
    librosa.resample fixes it, a tone keeps its frequency and (roll-off 0.95, pass band) its amplitude, energy above the new
    Nyquist frequency is removed."""
    import numpy as np
    from oracle import frontend as FE
    sr, n = 44100, 4410
    t = np.arange(n) / sr
    y = FE.resample_kaiser_best((0.5 * np.sin(2 * np.pi * 1000 * t)).astype(np.float32), sr, 16000)
    assert y.shape == (1600,)
    mid = y[200:-200]
    assert abs(np.abs(mid).max() - 0.5) < 5e-3
    hi = FE.resample_kaiser_best((0.5 * np.sin(2 * np.pi * 12000 * t)).astype(np.float32), sr, 16000)
    assert np.abs(hi[200:-200]).max() < 5e-3
