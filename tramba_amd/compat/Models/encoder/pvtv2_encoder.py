"""Shim for the reference's Models/encoder/pvtv2_encoder.py."""
from tramba_amd.encoders import PyramidVisionTransformerImpr, pvt_v2_b4  # noqa: F401
