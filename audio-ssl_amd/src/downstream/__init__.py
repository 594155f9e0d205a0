from src.downstream.downstream_encoder import DownstreamEncoder  # noqa: F401
