"""Shim for Models/mamba_init.py."""
from tramba_amd.modules import A_log_init, D_init, Dt_init  # noqa: F401
