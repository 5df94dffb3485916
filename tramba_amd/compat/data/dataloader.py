"""Shim for the reference's data/dataloader.py (train.py:9, test_TSOD.py:11, test_SOD.py:10 import RGB_Dataset)."""
from tramba_amd.data import ImageLoader, RGB_Dataset, get_transform, natural_sorted as sort  # noqa: F401
