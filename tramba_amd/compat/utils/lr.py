"""Shim for utils/lr.py."""
from tramba_amd.train import adjust_learning_rate  # noqa: F401
