"""Shim for Models/vmamba.py."""
from tramba_amd.modules import (SS2D, DWConv, DWMSMlp, LayerNorm2d, Linear2d, Mlp, MultiScaleDecoderBlock,  # noqa: F401
                                Permute, VSSBlock, VSSMEncoder, load_pretrained_Base)
from tramba_amd.ops import CrossMerge, CrossMerge_Line, CrossScan, CrossScan_Line, SelectiveScanOflex  # noqa: F401
