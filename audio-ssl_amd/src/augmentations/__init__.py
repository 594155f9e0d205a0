"""`src.augmentations` of the reference, backed by the batched HIP kernels (see augmentations.py)."""
from src.augmentations.augmentations import *  # noqa: F401,F403
from src.augmentations.augmentations import AugmentationModule  # noqa: F401
