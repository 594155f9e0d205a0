"""Upstream data path: CSV of audio files -> waveform batches -> (on the GPU) log-mel + two augmented views.

API mirror of `src/dataset/upstream_dataset.py:36-124` (BaseDataset, BaselineDataModule), re-shaped for the GPU:
the reference decodes, crops, STFTs and augments ONE clip at a time inside DataLoader workers; here the loader
only decodes and crops (the window start is drawn with the reference's generator, at the reference's position in the
python `random` stream), and `UpstreamFrontEnd` turns a whole [B, L] waveform batch into the two [B,1,64,T] views
with three fused launches (log-mel, normalise, views).
Decoding: the reference uses librosa.core.load (audioread/soundfile + resampy); neither is in the image, so WAV
files are read with scipy.io.wavfile (and .npy arrays directly) and resampled with scipy.signal.resample_poly -
audio ingest is a "next" row of SURVEY 8f, not part of the parity-checked hot path.
"""
import numpy as np
import pandas as pd
import torch

from src import _native as N
import torch.nn.functional as F
from torch.utils.data import DataLoader, Dataset

from src.utils import MelSpectrogramLibrosa, extract_log_mel_spectrogram, extract_window

AUDIO_SR = 16000


def load_audio(path, sr=AUDIO_SR):
    """-> float32 mono numpy array at `sr`: host decode (`ingest.decode`), then the windowed-sinc resampler on the GPU
    (`ingest.resample`, resampy kaiser_best as librosa.core.load uses it).  Inside a DataLoader worker process (no GPU context
    there) the clip is returned at its native rate wrapped in `NativeRate`; the main-process collate resamples it."""
    from src.dataset import ingest
    data, rate = ingest.decode(path)
    if rate == sr:
        return data
    import torch.utils.data as tud
    if tud.get_worker_info() is not None:
        return NativeRate(data, rate)
    return ingest.resample(torch.from_numpy(data), rate, sr).cpu().numpy()


class NativeRate:
    """A decoded clip that still has its file's sample rate (produced in loader workers, consumed by `WindowCollate`)."""

    def __init__(self, data, rate):
        self.data, self.rate = data, rate


class UpstreamFrontEnd:
    """waveforms [B, L] (device) -> (img_1, img_2), each [B, 1, n_mels, T] float32, entirely on the GPU."""

    def __init__(self, config, tfms):
        self.config = config
        self.tfms = tfms
        self.to_mel_spec = MelSpectrogramLibrosa()
        self.l2 = config["pretrain"]["normalization"] == "l2"

    @torch.no_grad()
    def __call__(self, waves, plan=None):
        if not waves.is_cuda:
            waves = waves.cuda(non_blocking=True)
        return self.tfms.augment_batch(self.log_mel(waves), plan=plan)

    @torch.no_grad()
    def log_mel(self, waves):
        """[B, L] device waveforms -> [B, n_mels, T] log-mel; `normalization: l2` first scales every clip to unit L2 norm
        (`F.normalize(waveform, dim=-1, p=2)`, src/dataset/upstream_dataset.py:61-62 of the reference) - one wave per clip."""
        if self.l2:
            waves = waves.float().contiguous()
            out = torch.empty_like(waves)
            inv = torch.empty(waves.shape[0], dtype=torch.float32, device=waves.device)
            N.call("l2norm_fwd", N.F32, waves, waves.shape[0], waves.shape[1], out, out, inv)
            waves = out
        return extract_log_mel_spectrogram(waves, self.to_mel_spec)

    # ---- one-batch-ahead pipelining: the front end of batch i+1 (log-mel, running norm, mixup, crops: ~0.4 ms of small
    #      launches, the running-norm scan being a one-workgroup recurrence) runs on its own stream underneath the training
    #      step of batch i.  All front-end state (RunningNorm, mixup ring) is only ever touched on that stream, in order.
    _stream = None

    def submit(self, waves, plan=None, after=None):
        """Start the front end for `waves` on the front-end stream; -> ticket for `collect`.  Call it BEFORE launching the
        training step it should overlap with (the front-end stream waits for what is already queued on the current one).
        after: event of an upload of `waves` issued on a copy stream of the caller (the front-end stream alone waits for it)."""
        if not waves.is_cuda:
            waves = waves.cuda(non_blocking=True)
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=waves.device)
        self._stream.wait_stream(torch.cuda.current_stream(waves.device))
        if after is not None:
            self._stream.wait_event(after)
        with torch.cuda.stream(self._stream):
            views = self(waves, plan)
            ev = torch.cuda.Event()
            ev.record()
        waves.record_stream(self._stream)
        return views, ev

    def collect(self, ticket):
        """-> (img_1, img_2) of a submitted batch, ordered before whatever is queued next on the current stream."""
        views, ev = ticket
        cur = torch.cuda.current_stream(views[0].device)
        cur.wait_event(ev)
        for v in views:
            v.record_stream(cur)
        return views


class BaseDataset(Dataset):
    def __init__(self, config, args, data_csv, tfms, per_sample=False):
        self.config = config
        self.tfms = tfms
        self.length = self.config["pretrain"]["input"]["length_wave"]
        self.norm_status = self.config["pretrain"]["normalization"]
        self.sampling_rate = self.config["pretrain"]["input"]["sampling_rate"]
        self.upstream = getattr(args, "upstream", None)
        self.data = pd.read_csv(data_csv)
        self.per_sample = per_sample
        self.to_mel_spec = MelSpectrogramLibrosa() if per_sample else None
        if self.config["pretrain"]["input"]["type"] != "raw_wav":
            raise NotImplementedError("only input.type == raw_wav is supported (SURVEY 2.4)")

    @property
    def unit_length(self):
        return int(self.length * 16000)

    def __getitem__(self, idx):
        wave = load_audio(self.data["files"][idx], self.sampling_rate)
        if isinstance(wave, NativeRate):
            return wave                      # resampled on the GPU by the main-process collate
        wave = torch.from_numpy(wave)
        if not self.per_sample:
            return wave                      # cropped at collate time, with the draws in the reference's order
        # reference-shaped per-sample path (one clip through the same kernels; slow, for API parity)
        waveform = extract_window(wave, data_size=self.length)
        if self.norm_status == "l2":
            waveform = F.normalize(waveform, dim=-1, p=2)
        lms = extract_log_mel_spectrogram(waveform.cuda(), self.to_mel_spec).unsqueeze(0)
        return self.tfms(lms) if self.tfms else lms

    def __len__(self):
        return len(self.data)


class WindowCollate:
    """Crop every clip of the batch to the training window and plan the augmentations.

    The python `random` stream is consumed exactly as the reference's sequential loader would (per clip: the window
    start, then that clip's view parameters), so crops and augmentation indices are bit-identical to a reference run
    with `num_workers=0`."""

    def __init__(self, tfms, unit_length, n_mels, hop=160):
        self.tfms, self.unit, self.n_mels, self.hop = tfms, unit_length, n_mels, hop

    def __call__(self, waves):
        from src.dataset import ingest
        waves = [ingest.resample(torch.from_numpy(w.data), w.rate, AUDIO_SR).cpu() if isinstance(w, NativeRate) else w for w in waves]
        B = len(waves)
        lens = np.array([len(w) for w in waves], np.int32)
        T = 1 + self.unit // self.hop
        self.tfms._ensure_ring(B)
        plan = self.tfms.plan(B, self.n_mels, T, lens=lens, unit=self.unit)
        starts = self.tfms.last_starts
        out = torch.zeros(B, self.unit, dtype=torch.float32)
        for b, w in enumerate(waves):
            n = len(w)
            if n >= self.unit:
                out[b] = w[starts[b]:starts[b] + self.unit]
            else:
                left = (self.unit - n) // 2
                out[b, left:left + n] = w
        return out, plan


class BaselineDataModule:
    def __init__(self, config, args, tfms, data_csv='./', batch_size=8, num_workers=8):
        self.config = config
        self.args = args
        self.data_dir_train = data_csv
        self.batch_size = batch_size
        self.num_workers = num_workers
        self.transformation = tfms
        self.dataset_sizes = {}
        self.name = "audio"
        self.front_end = UpstreamFrontEnd(config, tfms)

    def setup(self, stage=None):
        if stage == 'fit' or stage is None:
            self.train_dataset = BaseDataset(self.config, self.args, self.data_dir_train, self.transformation)
            self.dataset_sizes['train'] = len(self.train_dataset)

    def train_dataloader(self, sampler=None):
        unit = self.train_dataset.unit_length
        collate = WindowCollate(self.transformation, unit, self.config["pretrain"]["input"]["n_mels"])
        # decoding / resampling runs in `run.num_dataloader_workers` worker processes (they return the raw clips of a batch);
        # cropping + augmentation planning is stateful (ONE sequential random stream, as a reference run with num_workers=0
        # consumes it) and stays in the main process, applied to each batch as it arrives
        loader = DataLoader(self.train_dataset, shuffle=sampler is None, sampler=sampler, batch_size=self.batch_size,
                            num_workers=int(self.num_workers or 0), drop_last=True, collate_fn=_identity_collate,
                            persistent_workers=bool(self.num_workers))
        return _CollatedLoader(loader, collate)


def _identity_collate(items):
    return items


class _CollatedLoader:
    """Iterates a DataLoader of raw clip lists and applies the (stateful, main-process) window / plan collate to each batch."""

    def __init__(self, loader, collate):
        self.loader, self.collate = loader, collate

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for items in self.loader:
            waves, plan = self.collate(items)
            yield waves.pin_memory() if torch.cuda.is_available() else waves, plan
