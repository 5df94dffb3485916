"""Shim for the reference's Models/encoder/resnet_encoder.py."""
from tramba_amd.models import Bottleneck, ResNet  # noqa: F401
