"""Shim for the reference's Trambav6.py (Tramba-V)."""
from tramba_amd.models import BaseUMamba, VSSMDecoder, bulid_model  # noqa: F401
