"""Shim for utils/loss.py (only iou_loss is used by train.py:80-85)."""
from tramba_amd.train import iou_loss  # noqa: F401
