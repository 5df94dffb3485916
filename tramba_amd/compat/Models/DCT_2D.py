"""Shim for Models/DCT_2D.py."""
from tramba_amd.modules import DCT2D, DCT2DSpatialTransformLayer_x, DCT2DSpatialTransformLayer_y  # noqa: F401
