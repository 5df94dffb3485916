"""Shim for the reference's Trambav6_enc.py (Tramba-R; Tramba-S/P encoders raise NotImplementedError)."""
from tramba_amd.models import BaseUMambaEnc as BaseUMamba, ResNet, VSSMDecoder  # noqa: F401
from tramba_amd.models import bulid_model_enc as bulid_model  # noqa: F401
