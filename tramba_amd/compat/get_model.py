"""Shim for the reference's get_model.py:2-31."""
from tramba_amd.models import build  # noqa: F401
