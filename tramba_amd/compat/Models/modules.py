"""Shim for the hot subset of Models/modules.py."""
from tramba_amd.modules import (FinalPatchExpand_X4, FreqExpand2D, LayerNorm2d, Linear2d, Mlp, PatchExpand,  # noqa: F401
                                Permute)
