"""tramba_amd -- MI355X-native (gfx950) implementation of the Tramba hot path.

Drop-in for the reference's module surface (get_model.build / Trambav6.bulid_model /
Trambav6_enc.bulid_model, SS2D scan/merge plugin API, selective_scan_cuda_oflex fwd/bwd) with
every hot op a hand-written HIP kernel behind the C ABI of include/tramba_hip.h.
"""
from .graph import GraphedForward, GraphedTrainStep  # noqa: F401
from . import data, evaluate, hip  # noqa: F401  (hip: ctypes binding, loads lazily)
from .models import (BaseUMamba, BaseUMambaEnc, VSSMDecoder, build, bulid_model, bulid_model_enc,  # noqa: F401
                     prepare_inference)
from .modules import (DCT2D, SS2D, DropPath, DWConv, DWMSMlp, FinalPatchExpand_X4, FreqBlockv6,  # noqa: F401
                      FreqExpand2D, FreqSS2Dv6, LayerNorm2d, Linear2d, Mlp, MultiScaleDecoderBlock, PatchExpand,
                      VSSBlock, VSSMEncoder, load_pretrained_Base)
from .ops import (CrossMerge, CrossMerge_Dilation, CrossMerge_Line, CrossMerge_Window, CrossScan,  # noqa: F401
                  CrossScan_Dilation, CrossScan_Line, CrossScan_Window, SelectiveScanOflex,
                  selective_scan_cuda_oflex)

__version__ = "0.1.0"
