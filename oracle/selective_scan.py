"""Oracle: selective scan (wrapper over selective_scan_ref.c + a tiny pure-numpy form).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED for this operator (third-party
``selective_scan_cuda_oflex`` is absent from the reference tree; see the C file header).
Boundary restated: Models/SS2D/csms6s.py:904-923 (SelectiveScanOflex.forward/backward).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_scan.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "selective_scan_ref.c")
    if force or not os.path.exists(_SO) or (
        os.path.exists(src) and os.path.getmtime(_SO) < os.path.getmtime(src)
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a, ct=ctypes.c_double):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ct))


def _np64(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().double().numpy()
    return np.ascontiguousarray(t, dtype=np.float64)


def selective_scan_fwd(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True):
    """fp64 forward.  u, delta: (B, KD, L); A: (KD, N); B, C: (B, K, N, L); D, bias: (KD).
    Returns a float64 torch tensor (B, KD, L)."""
    u_, d_, A_, B_, C_, D_, b_ = map(_np64, (u, delta, A, B, C, D, delta_bias))
    nb, kd, L = u_.shape
    K, N = B_.shape[1], B_.shape[2]
    assert kd % K == 0 and A_.shape == (kd, N) and B_.shape == C_.shape == (nb, K, N, L)
    out = np.empty_like(u_)
    lib().oracle_selective_scan_fwd(
        _p(u_), _p(d_), _p(A_), _p(B_), _p(C_), _p(D_), _p(b_), int(delta_softplus),
        nb, kd, K, N, L, _p(out))
    return torch.from_numpy(out)


def selective_scan_bwd(u, delta, A, B, C, D, delta_bias, dout, delta_softplus=True):
    """fp64 backward -> (du, ddelta, dA, dB, dC, dD, ddelta_bias) as float64 tensors."""
    u_, d_, A_, B_, C_, D_, b_, g_ = map(_np64, (u, delta, A, B, C, D, delta_bias, dout))
    nb, kd, L = u_.shape
    K, N = B_.shape[1], B_.shape[2]
    du, dd = np.empty_like(u_), np.empty_like(u_)
    dA, dB, dC = np.zeros_like(A_), np.zeros_like(B_), np.zeros_like(C_)
    dD = np.zeros(kd) if D_ is not None else None
    db = np.zeros(kd) if b_ is not None else None
    lib().oracle_selective_scan_bwd(
        _p(u_), _p(d_), _p(A_), _p(B_), _p(C_), _p(D_), _p(b_), _p(g_), int(delta_softplus),
        nb, kd, K, N, L, _p(du), _p(dd), _p(dA), _p(dB), _p(dC), _p(dD), _p(db))
    return tuple(None if x is None else torch.from_numpy(x) for x in (du, dd, dA, dB, dC, dD, db))


def selective_scan_fwd_f32(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True):
    """fp32-arithmetic forward; only used to size tolerances."""
    def f32(t):
        return None if t is None else np.ascontiguousarray(t.detach().cpu().float().numpy())
    u_, d_, A_, B_, C_, D_, b_ = map(f32, (u, delta, A, B, C, D, delta_bias))
    nb, kd, L = u_.shape
    K, N = B_.shape[1], B_.shape[2]
    out = np.empty_like(u_)
    f = ctypes.c_float
    lib().oracle_selective_scan_fwd_f32(
        _p(u_, f), _p(d_, f), _p(A_, f), _p(B_, f), _p(C_, f), _p(D_, f), _p(b_, f),
        int(delta_softplus), nb, kd, K, N, L, _p(out, f))
    return torch.from_numpy(out)


def selective_scan_numpy(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True):
    """Independent pure-numpy restatement (vectorised over rows, loop over l); small cases."""
    u_, d_, A_, B_, C_, D_, b_ = map(_np64, (u, delta, A, B, C, D, delta_bias))
    nb, kd, L = u_.shape
    K, N = B_.shape[1], B_.shape[2]
    dt = d_ + (0.0 if b_ is None else b_[None, :, None])
    if delta_softplus:
        dt = np.where(dt > 20.0, dt, np.log1p(np.exp(np.minimum(dt, 20.0))))
    rep = kd // K
    Bx = np.repeat(B_, rep, axis=1)  # (B, KD, N, L)
    Cx = np.repeat(C_, rep, axis=1)
    h = np.zeros((nb, kd, N))
    out = np.empty_like(u_)
    for l in range(L):
        a = np.exp(dt[:, :, l, None] * A_[None])
        h = a * h + dt[:, :, l, None] * Bx[:, :, :, l] * u_[:, :, l, None]
        out[:, :, l] = (Cx[:, :, :, l] * h).sum(-1)
    if D_ is not None:
        out += D_[None, :, None] * u_
    return torch.from_numpy(out)


class SelectiveScanOracleFn(torch.autograd.Function):
    """Autograd wrapper so the functional oracle model can be differentiated on CPU."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D, delta_bias, delta_softplus):
        ctx.softplus = delta_softplus
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias)
        return selective_scan_fwd(u, delta, A, B, C, D, delta_bias, delta_softplus).to(u.dtype)

    @staticmethod
    def backward(ctx, dout):
        u, delta, A, B, C, D, bias = ctx.saved_tensors
        g = selective_scan_bwd(u, delta, A, B, C, D, bias, dout, ctx.softplus)
        cast = lambda t, ref: None if t is None else t.to(ref.dtype)
        return (cast(g[0], u), cast(g[1], delta), cast(g[2], A), cast(g[3], B), cast(g[4], C),
                cast(g[5], D), cast(g[6], bias), None)


def selective_scan(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True):
    return SelectiveScanOracleFn.apply(u, delta, A, B, C, D, delta_bias, delta_softplus)
