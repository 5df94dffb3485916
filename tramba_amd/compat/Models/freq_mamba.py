"""Shim for Models/freq_mamba.py."""
from tramba_amd.modules import FreqBlockv6, FreqSS2Dv6  # noqa: F401
