"""Shim for the reference's Trambav6_enc.py (Tramba-R / Tramba-S / Tramba-P)."""
from tramba_amd.encoders import SwinTransformer, pvt_v2_b4  # noqa: F401
from tramba_amd.models import BaseUMambaEnc as BaseUMamba, ResNet, VSSMDecoder  # noqa: F401
from tramba_amd.models import bulid_model_enc as bulid_model  # noqa: F401
