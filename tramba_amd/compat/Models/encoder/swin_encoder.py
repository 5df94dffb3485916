"""Shim for the reference's Models/encoder/swin_encoder.py."""
from tramba_amd.encoders import BasicLayer, PatchMerging, SwinTransformer, SwinTransformerBlock  # noqa: F401
