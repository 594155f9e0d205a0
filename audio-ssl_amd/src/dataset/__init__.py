from src.dataset.upstream_dataset import BaseDataset, BaselineDataModule, UpstreamFrontEnd  # noqa: F401
from src.dataset.downstream_dataset import DownstreamDataset, DownstreamDatasetHF, DownstreamFrontEnd  # noqa: F401
