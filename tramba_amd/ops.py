"""Drop-in operator surface of the reference's L0/L1 layers, backed by libtramba_hip.

* ``selective_scan_cuda_oflex``-compatible ``fwd`` / ``bwd`` (call sites
  Models/SS2D/csms6s.py:910, :920-922) and the ``SelectiveScanOflex`` autograd.Function
  (csms6s.py:904-923) with the reference's exact argument list.
* ``CrossScan*/CrossMerge*`` autograd.Functions (csms6s.py:13-216): the scan-order *plugin API*
  ``SS2D(..., scan=Cls, merge=Cls, k_group=K)``; ``Cls.apply(x:(B,C,H,W)) -> (B,K,C,H*W)``,
  ``merge.apply(ys:(B,K,D,H,W)) -> (B,D,H*W)``, each the adjoint (= backward) of the other.

Unlike the reference nothing happens at import time: tables are generated on demand for ANY
square size (the reference ships 12/24/48/96 only, csms6s.py:58-61,107-110,157) and cached
per device.
"""
import torch

from . import hip


# ----------------------------------------------------------------------------- L0
class _OflexExtension:
    """Object with the two entry points of the absent CUDA extension."""

    @staticmethod
    def fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows, oflex):
        out, ckpt = hip.selective_scan_fwd(_c(u), _c(delta), A, _c(B), _c(C), D, delta_bias,
                                           bool(delta_softplus), bool(oflex), want_ckpt=True)
        return out, ckpt

    @staticmethod
    def bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows):
        du, ddelta, dA, dB, dC, dD, dbias = hip.selective_scan_bwd(
            _c(u), _c(delta), A, _c(B), _c(C), D, delta_bias, _c(dout), x, bool(delta_softplus))
        dA = dA.to(A.dtype)
        dB, dC = dB.to(B.dtype), dC.to(C.dtype)
        dD = None if dD is None else dD.to(D.dtype)
        dbias = None if dbias is None else dbias.to(delta_bias.dtype)
        return du, ddelta, dA, dB, dC, dD, dbias


selective_scan_cuda_oflex = _OflexExtension()


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class SelectiveScanOflex(torch.autograd.Function):
    """Same call signature as csms6s.py:904-923."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1, backnrows=1,
                oflex=True):
        ctx.delta_softplus = delta_softplus
        out, x = selective_scan_cuda_oflex.fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, 1, oflex)
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias, x)
        return out

    @staticmethod
    def backward(ctx, dout, *args):
        u, delta, A, B, C, D, delta_bias, x = ctx.saved_tensors
        du, ddelta, dA, dB, dC, dD, ddelta_bias = selective_scan_cuda_oflex.bwd(
            u, delta, A, B, C, D, delta_bias, dout, x, ctx.delta_softplus, 1)
        return du, ddelta, dA, dB, dC, dD, ddelta_bias, None, None, None, None


# ----------------------------------------------------------------------------- L1
def _make_scan_pair(family: str, k: int, doc_scan: str, doc_merge: str):
    class Scan(torch.autograd.Function):
        _tramba_family = family
        _tramba_k = k

        @staticmethod
        def forward(ctx, x):
            b, c, h, w = x.shape
            ctx.shape = (b, c, h, w)
            order = hip.scan_order(family, h, w, x.device)
            return hip.cross_scan(_c(x), order)

        @staticmethod
        def backward(ctx, ys):
            b, c, h, w = ctx.shape
            order = hip.scan_order(family, h, w, ys.device)
            return hip.cross_merge(_c(ys), order).view(b, c, h, w)

    class Merge(torch.autograd.Function):
        _tramba_family = family
        _tramba_k = k

        @staticmethod
        def forward(ctx, ys):
            b, kk, d, h, w = ys.shape
            ctx.shape = (h, w)
            order = hip.scan_order(family, h, w, ys.device)
            return hip.cross_merge(_c(ys).view(b, kk, d, h * w), order)

        @staticmethod
        def backward(ctx, x):
            h, w = ctx.shape
            b, c, l = x.shape
            order = hip.scan_order(family, h, w, x.device)
            return hip.cross_scan(_c(x), order).view(b, order.k, c, h, w)

    Scan.__doc__, Merge.__doc__ = doc_scan, doc_merge
    return Scan, Merge


CrossScan, CrossMerge = _make_scan_pair(
    "raster", 4, "csms6s.py:13-31 (row-major, column-major and their flips)", "csms6s.py:34-55")
CrossScan_Line, CrossMerge_Line = _make_scan_pair(
    "helix", 8, "csms6s.py:161-185 Helix: 4 raster + 4 Bresenham-line directions", "csms6s.py:188-216")
CrossScan_Window, CrossMerge_Window = _make_scan_pair(
    "window", 4, "csms6s.py:113-129 window-partition scan (Hi-Fi branch)", "csms6s.py:132-152")
CrossScan_Dilation, CrossMerge_Dilation = _make_scan_pair(
    "dilation", 4, "csms6s.py:64-80 dilated scan, rate 4 (Lo-Fi branch)", "csms6s.py:83-103")
for _cls, _name in ((CrossScan, "CrossScan"), (CrossMerge, "CrossMerge"),
                    (CrossScan_Line, "CrossScan_Line"), (CrossMerge_Line, "CrossMerge_Line"),
                    (CrossScan_Window, "CrossScan_Window"), (CrossMerge_Window, "CrossMerge_Window"),
                    (CrossScan_Dilation, "CrossScan_Dilation"), (CrossMerge_Dilation, "CrossMerge_Dilation")):
    _cls.__name__ = _cls.__qualname__ = _name


def flops_selective_scan_fn(B=1, L=256, D=768, N=16, with_D=True, with_Z=False, with_complex=False):
    """FLOP model of the scan, same formula as csms6s.py:772-793 (9*B*L*D*N + B*D*L)."""
    assert not with_complex
    flops = 9 * B * L * D * N
    if with_D:
        flops += B * D * L
    if with_Z:
        flops += B * D * L
    return flops
