from src.dataset.upstream_dataset import BaseDataset, BaselineDataModule, UpstreamFrontEnd  # noqa: F401
