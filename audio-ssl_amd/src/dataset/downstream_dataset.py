"""Downstream (labelled) data path - API mirror of `src/dataset/downstream_dataset.py:64-107` of the reference
(`DownstreamDataset(args, config, split, tfms=None, labels_dict=None)`, a CSV with `wav` and `label` columns).

The reference decodes, crops and log-mels ONE clip per `__getitem__` on the CPU; here the dataset hands out the cropped
waveform and `DownstreamFrontEnd` turns a whole [B, L] batch into [B, 1, n_mels, T] log-mels on the GPU (`logmel_fwd`).
`per_sample=True` keeps the reference's item shape (log-mel computed through the same kernel, one clip per call).
The HuggingFace variant (`DownstreamDatasetHF`) downloads its data and is not available offline: it raises.
Label ids: the reference enumerates a python `set` of strings (order changes from run to run); here the labels are sorted."""
import pandas as pd
import torch
from torch.utils.data import Dataset

from src.dataset.upstream_dataset import load_audio
from src.utils import MelSpectrogramLibrosa, extract_log_mel_spectrogram, extract_window


class DownstreamDataset(Dataset):
    def __init__(self, args, config, split, tfms=None, labels_dict=None, per_sample=False):
        self.config, self.task, self.split = config, getattr(args, "task", None), split
        path = {"train": getattr(args, "train_csv", None), "valid": getattr(args, "valid_csv", None),
                "validation": getattr(args, "valid_csv", None), "test": getattr(args, "test_csv", None)}[split]
        self.dataset = pd.read_csv(path)
        self.sample_rate = config["downstream"]["input"]["sampling_rate"]
        self.duration = config["run"]["duration"]
        self.labels_dict = self.get_label2id() if labels_dict is None else labels_dict
        self.no_of_classes = len(self.labels_dict)
        self.per_sample = per_sample
        self.to_mel_spec = MelSpectrogramLibrosa() if per_sample else None
        self.tfms = tfms

    def get_label2id(self):
        return {k: i for i, k in enumerate(sorted(set(self.dataset["label"]), key=str))}

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        row = self.dataset.iloc[idx, :]
        wave = torch.from_numpy(load_audio(row["wav"], self.sample_rate))
        wave = extract_window(wave, data_size=self.duration)
        label = self.labels_dict[row["label"]]
        if not self.per_sample:
            return wave, label
        mel = extract_log_mel_spectrogram(wave.cuda(), self.to_mel_spec).unsqueeze(0)
        if self.tfms:
            mel = self.tfms(mel)
        return mel, label


class DownstreamDatasetHF(Dataset):
    def __init__(self, *a, **k):
        raise NotImplementedError("DownstreamDatasetHF downloads its dataset from the HuggingFace hub; provide CSVs "
                                  "(--train_csv / --test_csv) and use DownstreamDataset instead")


class DownstreamFrontEnd:
    """[B, L] waveforms -> [B, 1, n_mels, T] log-mel on the GPU (one launch per batch)."""

    def __init__(self):
        self.to_mel_spec = MelSpectrogramLibrosa()

    @torch.no_grad()
    def __call__(self, waves):
        return extract_log_mel_spectrogram(waves.cuda(non_blocking=True).float().contiguous(), self.to_mel_spec).unsqueeze(1)
