"""ctypes binding of libtramba_hip.so (the C ABI in include/tramba_hip.h).

PyTorch is plumbing here: it owns device memory and streams; every kernel on the Tramba hot
path is launched through this module onto torch's CURRENT stream.  There is no fallback: if
the library is missing or the tensors are not on a HIP device the call raises.
"""
import ctypes
import os
import threading
import weakref

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libtramba_hip.so")

F32, F16, BF16 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GELU, ACT_SIGMOID_GATE, ACT_GELU_GRAD_MUL = 0, 1, 2, 3, 4
SCAN_RASTER, SCAN_LINE, SCAN_HELIX, SCAN_WINDOW, SCAN_DILATION = range(5)
FAMILY = {"raster": SCAN_RASTER, "line": SCAN_LINE, "helix": SCAN_HELIX, "window": SCAN_WINDOW,
          "dilation": SCAN_DILATION}
PROF_SCAN_BOUNDARY, PROF_SCAN_FUSED, PROF_GEMM, PROF_MERGE, PROF_SCAN_BWD, PROF_WGRAD, PROF_LAYERNORM, PROF_DW = 0, 1, 2, 3, 4, 5, 6, 7

_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}

c_int, c_i64, c_f, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/tramba_hip.h one to one
SIGNATURES = {
    "tramba_last_error": (ctypes.c_char_p, []),
    "tramba_abi_version": (c_int, []),
    "tramba_device_error": (c_int, []),
    "tramba_tune_set": (c_int, [c_int, c_int]),
    "tramba_tune_get": (c_int, [c_int]),
    "tramba_profile_enable": (c_int, [c_int, c_int]),
    "tramba_profile_read": (c_int, [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "tramba_profile_min_units": (c_int, [c_int, ctypes.c_double]),
    "tramba_scan_family_k": (c_int, [c_int]),
    "tramba_default_window": (c_int, [c_int]),
    "tramba_scan_table": (c_int, [c_int, c_int, c_int, c_int, c_vp]),
    "tramba_scan_table_inverse": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "tramba_selective_scan_nchunk": (c_int, [c_int, c_int]),
    "tramba_selective_scan_fwd": (c_int, [c_vp] * 9 + [c_int] * 8 + [c_vp]),
    "tramba_selective_scan_bwd": (c_int, [c_vp] * 16 + [c_int] * 8 + [c_vp]),
    "tramba_cross_scan": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "tramba_cross_merge": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "tramba_ss2d_group_stride": (c_int, [c_int]),
    "tramba_ss2d_scan_workspace": (ctypes.c_size_t, [c_int] * 4),
    "tramba_ss2d_scan_cl": (c_int, [c_vp] * 9 + [ctypes.c_size_t] + [c_int] * 7 + [c_vp, c_int, c_vp]),
    "tramba_ss2d_scan_bwd_workspace": (ctypes.c_size_t, [c_int] * 4),
    "tramba_ss2d_scan_bwd_cl": (c_int, [c_vp] * 12 + [c_int, c_vp, c_vp, ctypes.c_size_t] + [c_int] * 9 + [c_vp]),
    "tramba_ss2d_bwd_prep_cl": (c_int, [c_vp] * 6 + [c_int] * 6 + [c_vp]),
    "tramba_ss2d_bwd_assemble_cl": (c_int, [c_vp] * 4 + [c_int] * 5 + [c_vp]),
    "tramba_ss2d_merge_grad_cl": (c_int, [c_vp] * 8 + [c_int] * 4 + [c_f, c_int, c_int, c_int, c_vp]),
    "tramba_ss2d_merge_norm_cl": (c_int, [c_vp] * 6 + [c_int] * 4 + [c_f, c_int, c_int, c_int, c_vp]),
    "tramba_layernorm_cl": (c_int, [c_vp] * 4 + [c_i64, c_int, c_f, c_int, c_int, c_vp]),
    "tramba_layernorm_bwd_parts": (c_i64, [c_i64, c_int, c_int]),
    "tramba_layernorm_bwd_cl": (c_int, [c_vp] * 5 + [c_i64, c_int, c_f, c_int, c_vp]),
    "tramba_layernorm_bwd_res_cl": (c_int, [c_vp] * 7 + [c_i64, c_vp, c_i64, c_int, c_f, c_int, c_vp]),
    "tramba_layernorm_bwd_any_cl": (c_int, [c_vp] * 7 + [c_i64, c_vp, c_i64, c_int, c_f, c_int, c_int, c_int, c_int, c_vp]),
    "tramba_shuffle_norm_bwd_cl": (c_int, [c_vp] * 5 + [c_int] * 5 + [c_f, c_int, c_vp]),
    "tramba_add_layernorm_cl": (c_int, [c_vp] * 3 + [c_i64] + [c_vp] * 5 + [c_i64, c_int, c_f, c_int, c_int, c_vp]),
    "tramba_shuffle_norm_cl": (c_int, [c_vp] * 4 + [c_int] * 5 + [c_f, c_int, c_vp]),
    "tramba_shuffle_norm_head_cl": (c_int, [c_vp] * 4 + [c_f, c_vp] + [c_int] * 5 + [c_f, c_int, c_vp]),
    "tramba_shuffle_norm_head_bwd_parts": (c_i64, [c_int] * 6),
    "tramba_shuffle_norm_head_bwd_cl": (c_int, [c_vp] * 6 + [c_int] * 5 + [c_f, c_int, c_vp]),
    "tramba_rowdot_cl": (c_int, [c_vp, c_vp, c_f, c_vp, c_i64, c_int, c_int, c_vp]),
    "tramba_rowdot_bwd_parts": (c_i64, [c_i64, c_int, c_int]),
    "tramba_rowdot_bwd_cl": (c_int, [c_vp] * 5 + [c_i64, c_int, c_int, c_vp]),
    "tramba_saliency_stats": (c_int, [c_vp] * 4 + [c_int] * 3 + [c_vp]),
    "tramba_dw_pack": (c_int, [c_vp] * 8 + [c_int] * 2 + [c_vp]),
    "tramba_dwconv_cl": (c_int, [c_vp] * 4 + [c_int] * 7 + [c_vp]),
    "tramba_dwconv_dual_cl": (c_int, [c_vp] * 5 + [c_int] * 8 + [c_vp]),
    "tramba_dw_unpack_grad": (c_int, [c_vp] * 5 + [c_int] * 3 + [c_vp]),
    "tramba_dw_pack_multi": (c_int, [c_vp] * 10 + [c_int, c_vp]),
    "tramba_dw_unpack_grad_multi": (c_int, [c_vp] * 8 + [c_int, c_vp]),
    "tramba_im2col3x3_cl": (c_int, [c_vp] * 2 + [c_int] * 8 + [c_vp]),
    "tramba_upsample_bilinear_bwd": (c_int, [c_vp] * 2 + [c_int] * 5 + [c_vp]),
    "tramba_col2im3x3_cl": (c_int, [c_vp] * 2 + [c_int] * 8 + [c_vp]),
    "tramba_dwconv_wgrad_parts": (c_i64, [c_int, c_int, c_int, c_int, c_int]),
    "tramba_dwconv_wgrad_cl": (c_int, [c_vp] * 3 + [c_int] * 6 + [c_vp]),
    "tramba_dct_split_cl": (c_int, [c_vp] * 6 + [c_int] * 4 + [c_vp]),
    "tramba_linear_cl": (c_int, [c_vp] * 5 + [c_i64, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "tramba_linear_dual_cl": (c_int, [c_vp] * 5 + [c_i64, c_int, c_int, c_int, c_int, c_vp]),
    "tramba_linear_ln_cl": (c_int, [c_vp] * 6 + [c_i64, c_int, c_int, c_f, c_int, c_int, c_int, c_vp]),
    "tramba_linear2_cl": (c_int, [c_vp, c_vp, c_int] + [c_vp] * 4 + [c_i64, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "tramba_wgrad_workspace": (ctypes.c_size_t, [c_i64, c_int, c_int, c_int, c_int]),
    "tramba_wgrad_cl": (c_int, [c_vp] * 4 + [ctypes.c_size_t, c_i64, c_int, c_int, c_int, c_int, c_i64, c_i64, c_int, c_i64,
                                             c_i64, c_int, c_int, c_int, c_vp]),
    "tramba_rows_gemm_cl": (c_int, [c_vp] * 3 + [c_int, c_i64, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "tramba_shadow_cast_multi": (c_int, [c_vp, c_int, c_i64, c_int, c_vp]),
    "tramba_slab_sum": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp]),
    "tramba_multi_sum": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "tramba_multi_sum_strided": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "tramba_wgrad_parts_cl": (c_int, [c_vp] * 4 + [ctypes.c_size_t, c_i64, c_int, c_int, c_int, c_int, c_i64, c_i64, c_int, c_i64,
                                                   c_i64, c_int, c_int, c_int, c_vp, c_vp]),
    "tramba_expand_norm_head_cl": (c_int, [c_vp] * 5 + [c_f, c_vp] + [c_int] * 5 + [c_f, c_int, c_vp]),
    "tramba_conv3x3s2_cl": (c_int, [c_vp] * 4 + [c_int] * 6 + [c_vp]),
    "tramba_stem_conv_ln_gelu": (c_int, [c_vp] * 6 + [c_int] * 3 + [c_f, c_int, c_int, c_vp]),
    "tramba_sod_loss_sums": (c_int, [c_vp] * 3 + [c_int] * 6 + [c_vp]),
    "tramba_sod_loss_finish": (c_int, [c_vp] * 4 + [c_int, c_int, c_i64, c_vp, c_vp]),
    "tramba_sod_loss_grad_workspace": (ctypes.c_size_t, [c_int] * 5),
    "tramba_sod_loss_grad": (c_int, [c_vp] * 6 + [ctypes.c_size_t] + [c_int] * 5 + [c_vp]),
    "tramba_adam_step": (c_int, [c_vp] * 6 + [c_int] + [ctypes.c_double] * 5 + [c_vp]),
}

_lib = None
_lock = threading.Lock()


def lib():
    """Load the shared library (raises if it has not been built -- never falls back)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: build it with `python -m tramba_amd.buildlib` "
                        "(hipcc, gfx950).  tramba_amd has no CPU or PyTorch fallback.")
                l = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(l, name)
                    fn.restype, fn.argtypes = res, args
                _lib = l
    return _lib


class TrambaHipError(RuntimeError):
    pass


def _check(rc, what):
    if rc < 0:
        raise TrambaHipError(f"{what}: {lib().tramba_last_error().decode()} (code {rc})")
    return rc


def dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TrambaHipError(f"unsupported dtype {t.dtype}") from None


def _dev(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise TrambaHipError("tramba_amd kernels need tensors on a HIP device (no CPU fallback)")
        if not t.is_contiguous():
            raise TrambaHipError("tramba_amd kernels need contiguous tensors")


def _ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """the current HIP stream of the current device as an integer handle.  torch.cuda.current_stream() builds a Stream
    object per call (~9 us): with ~850 library launches per training step that alone was 7 ms of the launch threads' time."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


_f32_cache = {}


def _f32(t):
    """fp32 contiguous view of a parameter / buffer.  16-bit parameters (prepare_inference) are converted ONCE per
    (object, version, storage): a cast kernel per bias per call was ~10 launches of every forward."""
    if t is None:
        return None
    if t.dtype == torch.float32 and t.is_contiguous():
        return t.detach()
    key = id(t)
    ver = (t._version, t.data_ptr(), t.dtype)
    hit = _f32_cache.get(key)
    if hit is not None and hit[0] == ver and hit[2]() is t:
        return hit[1]
    conv = t.detach().float().contiguous()
    try:
        ref = weakref.ref(t, lambda _r, k=key: _f32_cache.pop(k, None))
    except TypeError:
        return conv
    _f32_cache[key] = (ver, conv, ref)
    return conv


# ----------------------------------------------------------------------------- tables
_table_cache = {}


def default_window(h: int) -> int:
    return lib().tramba_default_window(h)


def scan_table_host(family: str, h: int, w: int = None, param: int = 0) -> np.ndarray:
    """(K, L) int32 flat pixel index per sequence position (host)."""
    w = h if w is None else w
    fam = FAMILY[family]
    k = _check(lib().tramba_scan_family_k(fam), "scan_family_k")
    out = np.empty((k, h * w), dtype=np.int32)
    _check(lib().tramba_scan_table(fam, h, w, param, out.ctypes.data), f"scan_table({family},{h})")
    return out


def scan_table_inverse_host(table: np.ndarray):
    k, l = table.shape
    table = np.ascontiguousarray(table, dtype=np.int32)
    ptr = np.empty(l + 1, dtype=np.int32)
    idx = np.empty(k * l, dtype=np.int32)
    _check(lib().tramba_scan_table_inverse(table.ctypes.data, k, l, ptr.ctypes.data, idx.ctypes.data),
           "scan_table_inverse")
    return ptr, idx


class ScanOrder:
    """Device-resident scan table + its inverse (CSR) for one (family, size, device)."""

    def __init__(self, family, h, w, param, device):
        host = scan_table_host(family, h, w, param)
        ptr, idx = scan_table_inverse_host(host)
        self.family, self.h, self.w, self.k, self.l = family, h, w, host.shape[0], host.shape[1]
        self.table = torch.from_numpy(host).to(device)
        self.inv_ptr = torch.from_numpy(ptr).to(device)
        self.inv_idx = torch.from_numpy(idx).to(device)
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(device).synchronize()   # tables are shared by every stream from here on


def scan_order(family: str, h: int, w: int, device, param: int = 0) -> ScanOrder:
    key = (family, h, w, param, str(device))
    so = _table_cache.get(key)
    if so is None:
        so = _table_cache[key] = ScanOrder(family, h, w, param, device)
    return so


# ----------------------------------------------------------------------------- profiling / tuning
TUNE_MERGE_FORM, TUNE_SCAN_FORM, TUNE_SCAN_W, TUNE_GEMM_TILE, TUNE_MAILBOX_SKIP, TUNE_WGRAD_FORM, TUNE_DW_FORM, TUNE_DW_ROWS = 0, 1, 2, 3, 4, 5, 6, 7


def device_error():
    """raise TrambaHipError if a kernel launched earlier raised the library's device error word (tramba_device_error: a fused
    scan whose carry mailbox timed out leaves NaN in its output); call after a synchronisation to be certain"""
    _check(lib().tramba_device_error(), "device_error")


def tune_set(knob: int, value: int):
    """kernel-variant selection for A/B timing scripts (0 = the library's own choice)"""
    _check(lib().tramba_tune_set(knob, value), "tune_set")


def profile_enable(which: int, on: bool):
    _check(lib().tramba_profile_enable(which, int(on)), "profile_enable")


def profile_min_units(which: int, min_units: float):
    """time only launches accounting for at least `min_units` algorithmic bytes / flops (0 = all)"""
    _check(lib().tramba_profile_min_units(which, float(min_units)), "profile_min_units")


def profile_read(which: int):
    ms, units = ctypes.c_double(), ctypes.c_double()
    n = _check(lib().tramba_profile_read(which, ctypes.byref(ms), ctypes.byref(units)), "profile_read")
    return n, ms.value, units.value


# ----------------------------------------------------------------------------- L0 / L1 boundary ops
def selective_scan_nchunk(l: int, dtype: torch.dtype) -> int:
    return lib().tramba_selective_scan_nchunk(l, _DT[dtype])


def selective_scan_fwd(u, delta, A, B, C, D, delta_bias, delta_softplus=True, oflex=True, want_ckpt=True):
    _dev(u, delta, A, B, C, D, delta_bias)
    nb, kd, l = u.shape
    k, n = B.shape[1], B.shape[2]
    if delta.shape != u.shape or A.shape != (kd, n) or B.shape != (nb, k, n, l) or C.shape != B.shape:
        raise TrambaHipError(f"selective_scan_fwd: inconsistent shapes u{tuple(u.shape)} delta{tuple(delta.shape)} "
                             f"A{tuple(A.shape)} B{tuple(B.shape)} C{tuple(C.shape)}")
    if not (delta.dtype == B.dtype == C.dtype == u.dtype):
        raise TrambaHipError("selective_scan_fwd: u, delta, B, C must share a dtype")
    out = torch.empty((nb, kd, l), dtype=torch.float32 if oflex else u.dtype, device=u.device)
    ckpt = None
    if want_ckpt:
        ckpt = torch.empty((nb, kd, selective_scan_nchunk(l, u.dtype), n), dtype=torch.float32, device=u.device)
    A, D, delta_bias = _f32(A), _f32(D), _f32(delta_bias)
    _check(lib().tramba_selective_scan_fwd(
        _ptr(u), _ptr(delta), _ptr(A), _ptr(B), _ptr(C), _ptr(D), _ptr(delta_bias), _ptr(out), _ptr(ckpt),
        nb, kd, k, n, l, dt(u), dt(out), int(delta_softplus), _stream()), "selective_scan_fwd")
    return out, ckpt


def selective_scan_bwd(u, delta, A, B, C, D, delta_bias, dout, ckpt, delta_softplus=True):
    _dev(u, delta, A, B, C, D, delta_bias, dout, ckpt)
    nb, kd, l = u.shape
    k, n = B.shape[1], B.shape[2]
    A32, D32, b32 = _f32(A), _f32(D), _f32(delta_bias)
    dout = dout.float().contiguous()
    du, ddelta = torch.empty_like(u), torch.empty_like(delta)
    dA = torch.zeros((kd, n), dtype=torch.float32, device=u.device)
    ncopy = max(1, min(16, (kd // k) // 8))  # private dB/dC copies: <= 1/8 of a group's rows per address
    dB = torch.zeros((ncopy, nb, k, n, l), dtype=torch.float32, device=u.device)
    dC = torch.zeros_like(dB)
    dD = torch.zeros(kd, dtype=torch.float32, device=u.device) if D is not None else None
    dbias = torch.zeros(kd, dtype=torch.float32, device=u.device) if delta_bias is not None else None
    _check(lib().tramba_selective_scan_bwd(
        _ptr(u), _ptr(delta), _ptr(A32), _ptr(B), _ptr(C), _ptr(D32), _ptr(b32), _ptr(dout), _ptr(ckpt),
        _ptr(du), _ptr(ddelta), _ptr(dA), _ptr(dB), _ptr(dC), _ptr(dD), _ptr(dbias),
        nb, kd, k, n, l, dt(u), int(delta_softplus), ncopy, _stream()), "selective_scan_bwd")
    return du, ddelta, dA, dB.sum(0), dC.sum(0), dD, dbias


def cross_scan(x, order: ScanOrder):
    """x: (B, C, H, W) or (B, C, L) contiguous NCHW -> (B, K, C, L)."""
    _dev(x)
    b, c = x.shape[0], x.shape[1]
    l = order.l
    xs = torch.empty((b, order.k, c, l), dtype=x.dtype, device=x.device)
    _check(lib().tramba_cross_scan(_ptr(x), _ptr(order.table), _ptr(xs), b, c, l, order.k, dt(x), _stream()),
           "cross_scan")
    return xs


def cross_merge(ys, order: ScanOrder):
    """ys: (B, K, C, L) -> (B, C, L)."""
    _dev(ys)
    b, k, c, l = ys.shape
    if k != order.k or l != order.l:
        raise TrambaHipError(f"cross_merge: ys {tuple(ys.shape)} does not match table K={order.k} L={order.l}")
    y = torch.empty((b, c, l), dtype=ys.dtype, device=ys.device)
    _check(lib().tramba_cross_merge(_ptr(ys), _ptr(order.inv_ptr), _ptr(order.inv_idx), _ptr(y), b, c, l, k,
                                    dt(ys), _stream()), "cross_merge")
    return y


# ----------------------------------------------------------------------------- channels-last kernels
def ss2d_group_stride(r: int) -> int:
    """Floats per direction group of the x_proj output: [dt_0..dt_{R-1}, 0-pad to R8 = 8*ceil(R/8), B, C, 0, 0]."""
    return lib().tramba_ss2d_group_stride(r)


def pad_x_proj_weight(x_proj_weight: torch.Tensor) -> torch.Tensor:
    """(K, R+2, D) -> (K*RG, D) with zero rows as padding, the layout ss2d_scan_cl consumes."""
    k, r2, d = x_proj_weight.shape
    rg = ss2d_group_stride(r2 - 2)
    w = x_proj_weight.new_zeros((k, rg, d))
    w[:, :r2 - 2] = x_proj_weight[:, :r2 - 2]
    w[:, rg - 4:rg - 2] = x_proj_weight[:, r2 - 2:]      # B, C after the 8-padded ranks
    return w.reshape(k * rg, d)


_scan_ws = {}
_scan_ws_retired = []


def _scan_workspace(device, nbytes):
    """Per-(device, stream) scratch for the wave-segment scan (grown on demand, reused by every launch: launches
    on one stream are ordered, and graph capture sees a stable address; concurrent streams get their own)."""
    key = (str(device), _stream())
    ws = _scan_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _scan_ws_retired.append(ws)      # a captured hipGraph may still replay launches that point at the old buffer
        ws = _scan_ws[key] = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8, device=device)
    return ws


def ss2d_scan_states(x, order: ScanOrder):
    """the buffer ss2d_scan_cl(states=...) fills for ss2d_scan_bwd_cl(states=...): the recurrence state entering every
    32-position tile, (B, K, ceil(L/32) + 8, D) f32"""
    b, l, d = x.shape
    return torch.empty(lib().tramba_ss2d_scan_bwd_workspace(b, l, d, order.k), dtype=torch.uint8, device=x.device)


def ss2d_scan_cl(x, xdbl, order: ScanOrder, dt_w, dt_bias, A, Ds, ys_dtype=torch.float32, segmented=True, states=None,
                 a_log=False):
    """x: (B, L, D); xdbl: (B, L, K*RG) f32 -> ys (B, K, L, D).  With `segmented` a workspace is passed and
    the library picks the wave-segment or a chained form per shape (tune_set(TUNE_SCAN_FORM, ...) forces one).
    states (training): a buffer from ss2d_scan_states(); the launch saves the per-tile entering states in it (chained forms
    only, ys in the input dtype), and the backward launch given the same buffer skips its first sweep.
    a_log (training): `A` is the parameter A_logs, the kernel forms -exp(A_logs) itself."""
    _dev(x, xdbl, dt_w, dt_bias, A, Ds)
    b, l, d = x.shape
    k, r = order.k, dt_w.shape[-1]
    if xdbl.dtype != torch.float32 or xdbl.shape != (b, l, k * ss2d_group_stride(r)):
        raise TrambaHipError(f"ss2d_scan_cl: xdbl must be f32 (B,L,K*RG), got {xdbl.dtype} {tuple(xdbl.shape)}")
    if l != order.l or dt_w.shape != (k, d, r) or A.numel() != k * d or Ds.numel() != k * d or dt_bias.numel() != k * d:
        raise TrambaHipError("ss2d_scan_cl: parameter shapes do not match (K, D, R)")
    ys = torch.empty((b, k, l, d), dtype=ys_dtype, device=x.device)
    ws, ws_bytes = None, 0
    if segmented:
        ws_bytes = lib().tramba_ss2d_scan_workspace(b, l, d, k)
        ws = _scan_workspace(x.device, ws_bytes)
    _check(lib().tramba_ss2d_scan_cl(_ptr(x), _ptr(xdbl), _ptr(order.table), _ptr(dt_w), _ptr(dt_bias), _ptr(A),
                                     _ptr(Ds), _ptr(ys), _ptr(ws), ws_bytes, b, l, d, k, r, dt(x), dt(ys), _ptr(states),
                                     int(a_log), _stream()), "ss2d_scan_cl")
    return ys


def ss2d_merge_sum_cl(ys, order: ScanOrder, out_dtype):
    """ys (B, K, L, D) in sequence order -> (B, L, D): CrossMerge alone (no norm), fixed summation order."""
    _dev(ys)
    b, k, l, d = ys.shape
    y = torch.empty((b, l, d), dtype=out_dtype, device=ys.device)
    _check(lib().tramba_ss2d_merge_norm_cl(_ptr(ys), _ptr(order.inv_ptr), _ptr(order.inv_idx), None, None, _ptr(y), b, l,
                                           d, k, -1.0, ACT_NONE, dt(ys), dt(y), _stream()), "ss2d_merge_sum_cl")
    return y


def ss2d_scan_bwd_cl(x, xdbl, order: ScanOrder, dt_w, dt_bias, A, Ds, gym, g_seq=None, states=None, a_log=False,
                     bc_partials=False):
    """Backward of ss2d_scan_cl + merge.  gym (B, L, D) f32 = gradient of the merged map.
    Returns gu, graw (B,K,L,D) x.dtype, gB, gC (B,K,L) f32, gpar (B,3,K,D) f32 (dA, dD, dbias planes).
    g_seq: a ZEROED (B,K,L,RG) f32 x_dbl-gradient table in sequence order -- gB / gC are then accumulated straight into its
    B / C columns (RG - 4, RG - 3) and returned as views of it.
    states: the buffer the forward launch filled (ss2d_scan_cl(states=...)): the sweep that recomputes them is skipped.
    a_log: `A` is the parameter A_logs; gpar's first plane is then dL/dA_logs.
    bc_partials: gB / gC come back as (B, K, ceil(D/32), L) per-channel-tile partial sums written without atomics (no zero
    fill; ss2d_bwd_prep adds them in a fixed order)."""
    _dev(x, xdbl, dt_w, dt_bias, A, Ds, gym, g_seq)
    b, l, d = x.shape
    k, r = order.k, dt_w.shape[-1]
    if gym.dtype not in (torch.float32, x.dtype) or gym.shape != x.shape:
        raise TrambaHipError("ss2d_scan_bwd_cl: gym must be (B, L, D) in f32 or the activation dtype")
    gu = torch.empty((b, k, l, d), dtype=x.dtype, device=x.device)
    graw = torch.empty_like(gu)
    if bc_partials:
        gB = torch.empty((b, k, (d + 31) // 32, l), dtype=torch.float32, device=x.device)
        gC = torch.empty_like(gB)
        bcs = 1
    elif g_seq is None:
        gB = torch.zeros((b, k, l), dtype=torch.float32, device=x.device)
        gC = torch.zeros_like(gB)
        bcs = 1
    else:
        rg = ss2d_group_stride(r)
        if g_seq.shape != (b, k, l, rg) or g_seq.dtype != torch.float32 or not g_seq.is_contiguous():
            raise TrambaHipError(f"ss2d_scan_bwd_cl: g_seq must be a contiguous float32 (B, K, L, {rg}) tensor")
        gB, gC, bcs = g_seq[..., rg - 4], g_seq[..., rg - 3], rg
    gpar = torch.empty((b, 3, k, d), dtype=torch.float32, device=x.device)
    ws_bytes = lib().tramba_ss2d_scan_bwd_workspace(b, l, d, k)
    if states is not None:
        _dev(states)
        if states.numel() * states.element_size() < ws_bytes:
            raise TrambaHipError("ss2d_scan_bwd_cl: the states buffer is smaller than ss2d_scan_states() makes it")
        ws = states
    else:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    _check(lib().tramba_ss2d_scan_bwd_cl(_ptr(x), _ptr(xdbl), _ptr(order.table), _ptr(dt_w), _ptr(dt_bias), _ptr(A),
                                         _ptr(Ds), _ptr(gym), _ptr(gu), _ptr(graw), _ptr(gB), _ptr(gC), bcs, _ptr(gpar),
                                         _ptr(ws), ws_bytes, 0 if states is None else 1, b, l, d, k, r, dt(x), dt(gym),
                                         (1 if a_log else 0) | (2 if bc_partials else 0), _stream()), "ss2d_scan_bwd_cl")
    return gu, graw, gB, gC, gpar


def ss2d_bwd_prep(xdbl, order: ScanOrder, bpart, cpart, r, dtype):
    """after ss2d_scan_bwd_cl(bc_partials=True): (ranks (B,K,L,R8) `dtype` = the dt-rank rows of xdbl in sequence order,
    g_seq (B,K,L,RG) f32 with the summed dB / dC in columns R8, R8 + 1; the rank columns are for rows_gemm_cl to fill)"""
    _dev(xdbl, bpart, cpart)
    b, l, pc = xdbl.shape
    k = order.k
    rg = ss2d_group_stride(r)
    if pc != k * rg or bpart.shape != cpart.shape or bpart.shape[:2] != (b, k) or bpart.shape[3] != l:
        raise TrambaHipError("ss2d_bwd_prep: operand shapes do not match")
    ranks = torch.empty((b, k, l, rg - 4), dtype=dtype, device=xdbl.device)
    gseq = torch.empty((b, k, l, rg), dtype=torch.float32, device=xdbl.device)
    _check(lib().tramba_ss2d_bwd_prep_cl(_ptr(xdbl), _ptr(order.table), _ptr(bpart), _ptr(cpart), _ptr(ranks), _ptr(gseq),
                                         b, l, k, r, bpart.shape[2], _DT[dtype], _stream()), "ss2d_bwd_prep_cl")
    return ranks, gseq


def ss2d_bwd_assemble(gseq, order: ScanOrder, r, dtype):
    """gseq (B,K,L,RG) f32 in sequence order -> (B, L, K*RG) `dtype` in spatial order (gather-sum through the inverse table)"""
    _dev(gseq)
    b, k, l, rg = gseq.shape
    if k != order.k or l != order.l or rg != ss2d_group_stride(r):
        raise TrambaHipError("ss2d_bwd_assemble: gseq does not match the scan order / dt_rank")
    out = torch.empty((b, l, k * rg), dtype=dtype, device=gseq.device)
    _check(lib().tramba_ss2d_bwd_assemble_cl(_ptr(gseq), _ptr(order.inv_ptr), _ptr(order.inv_idx), _ptr(out), b, l, k, r,
                                             _DT[dtype], _stream()), "ss2d_bwd_assemble_cl")
    return out


def ss2d_merge_grad_cl(gu, order: ScanOrder, addend, zpre):
    """(merge(gu) + addend) * silu'(zpre): gu (B,K,L,D) sequence order, addend / zpre (B,L,D) -> (B,L,D), zpre's dtype"""
    _dev(gu, addend, zpre)
    b, k, l, d = gu.shape
    if zpre.shape != (b, l, d) or (addend is not None and (addend.shape != zpre.shape or addend.dtype != zpre.dtype)):
        raise TrambaHipError("ss2d_merge_grad_cl: operand shapes / dtypes do not match")
    y = torch.empty((b, l, d), dtype=zpre.dtype, device=gu.device)
    _check(lib().tramba_ss2d_merge_grad_cl(_ptr(gu), _ptr(order.inv_ptr), _ptr(order.inv_idx), None, None, _ptr(y),
                                           _ptr(addend), _ptr(zpre), b, l, d, k, -1.0, ACT_NONE, dt(gu), dt(y), _stream()),
           "ss2d_merge_grad_cl")
    return y


def ss2d_merge_norm_cl(ys, order: ScanOrder, ln_w, ln_b, eps, act, out_dtype):
    _dev(ys, ln_w, ln_b)
    b, k, l, d = ys.shape
    y = torch.empty((b, l, d), dtype=out_dtype, device=ys.device)
    _check(lib().tramba_ss2d_merge_norm_cl(_ptr(ys), _ptr(order.inv_ptr), _ptr(order.inv_idx), _ptr(ln_w),
                                           _ptr(ln_b), _ptr(y), b, l, d, k, eps, act, dt(ys), dt(y), _stream()),
           "ss2d_merge_norm_cl")
    return y


def layernorm_cl(x, w, b, eps=1e-5, act=ACT_NONE):
    """x: (..., C) contiguous."""
    _dev(x, w, b)
    c = x.shape[-1]
    y = torch.empty_like(x)
    _check(lib().tramba_layernorm_cl(_ptr(x), _ptr(w), _ptr(b), _ptr(y), x.numel() // c, c, eps, act, dt(x),
                                     _stream()), "layernorm_cl")
    return y


def layernorm_bwd_cl(x, dy, w, eps=1e-5, defer=False):
    """x, dy: (..., C) contiguous -> (dx like x, dw (C) f32, db (C) f32).  defer: dw / db go to autograd as the gradients of
    fp32 leaves as they are (see _SumQueue); anything that reads or casts them must pass False."""
    _dev(x, dy, w)
    c = x.shape[-1]
    rows = x.numel() // c
    dx = torch.empty_like(x)
    part = torch.empty((lib().tramba_layernorm_bwd_parts(rows, c, dt(x)), 2, c), dtype=torch.float32, device=x.device)
    _check(lib().tramba_layernorm_bwd_cl(_ptr(x), _ptr(dy), _ptr(w), _ptr(dx), _ptr(part), rows, c, eps, dt(x), _stream()),
           "layernorm_bwd_cl")
    s = slab_sum(part, defer=defer)
    return dx, s[0], s[1]


def add_layernorm_cl(x, y, mask, w, b, eps=1e-5, act=ACT_NONE, dual=False):
    """(xsum, n, n_act): xsum = x + y * mask[sample] (None when y is None), n = LayerNorm(xsum or x), n_act = act(n) when
    `dual`.  x, y (B, ..., C) contiguous, mask (B) f32 or None."""
    _dev(x, y, mask, w, b)
    c = x.shape[-1]
    rows = x.numel() // c
    if y is not None and (y.shape != x.shape or y.dtype != x.dtype):
        raise TrambaHipError("add_layernorm_cl: x / y mismatch")
    if mask is not None and (mask.dtype != torch.float32 or mask.numel() != x.shape[0]):
        raise TrambaHipError("add_layernorm_cl: mask must be float32 with one entry per sample")
    xsum = torch.empty_like(x) if y is not None else None
    n = torch.empty_like(x)
    na = torch.empty_like(x) if dual else None
    _check(lib().tramba_add_layernorm_cl(_ptr(x), _ptr(y), _ptr(mask), rows // x.shape[0], _ptr(w), _ptr(b), _ptr(xsum),
                                         _ptr(n), _ptr(na), rows, c, eps, act, dt(x), _stream()), "add_layernorm_cl")
    return xsum, n, na


def layernorm_bwd_res_cl(x, dy, w, eps=1e-5, gres=None, mask=None, want_masked=False, defer=False):
    """LayerNorm backward on the residual stream: (dx + gres, that * mask[sample] or None, dw, db)"""
    _dev(x, dy, w, gres, mask)
    c = x.shape[-1]
    rows = x.numel() // c
    if gres is not None and (gres.shape != x.shape or gres.dtype != x.dtype):
        raise TrambaHipError("layernorm_bwd_res_cl: gres must match x")
    dx = torch.empty_like(x)
    dxm = torch.empty_like(x) if want_masked else None
    part = torch.empty((lib().tramba_layernorm_bwd_parts(rows, c, dt(x)), 2, c), dtype=torch.float32, device=x.device)
    _check(lib().tramba_layernorm_bwd_res_cl(_ptr(x), _ptr(dy), _ptr(w), _ptr(dx), _ptr(part), _ptr(gres), _ptr(mask),
                                             rows // x.shape[0], _ptr(dxm), rows, c, eps, dt(x), _stream()),
           "layernorm_bwd_res_cl")
    s = slab_sum(part, defer=defer)
    return dx, dxm, s[0], s[1]


def shuffle_norm_cl(x, w, b, p, eps=1e-5):
    """x: (B, H, W, P*P*C) -> (B, H*P, W*P, C), pixel-shuffle + LayerNorm over C."""
    _dev(x, w, b)
    bb, h, wd, cc = x.shape
    c = cc // (p * p)
    y = torch.empty((bb, h * p, wd * p, c), dtype=x.dtype, device=x.device)
    _check(lib().tramba_shuffle_norm_cl(_ptr(x), _ptr(w), _ptr(b), _ptr(y), bb, h, wd, c, p, eps, dt(x), _stream()),
           "shuffle_norm_cl")
    return y


def shuffle_norm_bwd_cl(x, dy, w, p, eps=1e-5, defer=False):
    """backward of shuffle_norm_cl: x (B,H,W,P*P*C), dy (B,H*P,W*P,C) -> (dx like x, dw (C), db (C))"""
    _dev(x, dy, w)
    bb, h, wd, cc = x.shape
    c = cc // (p * p)
    if dy.shape != (bb, h * p, wd * p, c) or dy.dtype != x.dtype:
        raise TrambaHipError("shuffle_norm_bwd_cl: dy does not match the shuffled map")
    rows = bb * h * wd * p * p
    dx = torch.empty_like(x)
    part = torch.empty((lib().tramba_layernorm_bwd_parts(rows, c, dt(x)), 2, c), dtype=torch.float32, device=x.device)
    _check(lib().tramba_shuffle_norm_bwd_cl(_ptr(x), _ptr(dy), _ptr(w), _ptr(dx), _ptr(part), bb, h, wd, c, p, eps, dt(x),
                                            _stream()), "shuffle_norm_bwd_cl")
    s = slab_sum(part, defer=defer)
    return dx, s[0], s[1]


def shuffle_norm_head_cl(x, w, b, head_w, head_b: float, p, eps=1e-5):
    """x: (B, H, W, P*P*C) -> (B, H*P, W*P) f32: pixel-shuffle + LayerNorm over C + dot with head_w (C) + head_b."""
    _dev(x, w, b, head_w)
    bb, h, wd, cc = x.shape
    c = cc // (p * p)
    y = torch.empty((bb, h * p, wd * p), dtype=torch.float32, device=x.device)
    _check(lib().tramba_shuffle_norm_head_cl(_ptr(x), _ptr(w), _ptr(b), _ptr(head_w), float(head_b), _ptr(y), bb, h, wd,
                                             c, p, eps, dt(x), _stream()), "shuffle_norm_head_cl")
    return y


def shuffle_norm_head_bwd_cl(x, g, ln_w, head_w, p, eps=1e-5):
    """Backward of shuffle_norm_head_cl in one pass over x (B, H, W, P*P*C): g (B, H*P, W*P) f32 -> (dx like x, part (S, C + 4)
    f32): the rows of `part` sum to A_c = sum g xhat_c (slots 0..C) and G = sum g (slot C), from which every parameter gradient
    follows (tramba_shuffle_norm_head_bwd_cl)."""
    _dev(x, g, ln_w, head_w)
    bb, h, wd, cc = x.shape
    c = cc // (p * p)
    if g.dtype != torch.float32 or g.numel() != bb * h * wd * p * p or ln_w.dtype != torch.float32 or head_w.dtype != torch.float32:
        raise TrambaHipError("shuffle_norm_head_bwd_cl: fp32 logit gradients / parameters of the forward's shapes")
    if ln_w.numel() != c or head_w.numel() != c:
        raise TrambaHipError("shuffle_norm_head_bwd_cl: parameter vectors of another length than C")
    nparts = lib().tramba_shuffle_norm_head_bwd_parts(bb, h, wd, c, p, dt(x))
    if nparts <= 0:
        raise TrambaHipError(f"shuffle_norm_head_bwd_cl: C = {c} is not served by the rows kernel")
    dx = torch.empty_like(x)
    part = torch.empty((nparts, c + 4), dtype=torch.float32, device=x.device)
    _check(lib().tramba_shuffle_norm_head_bwd_cl(_ptr(x), _ptr(g), _ptr(ln_w), _ptr(head_w), _ptr(dx), _ptr(part), bb, h, wd, c, p,
                                                 eps, dt(x), _stream()), "shuffle_norm_head_bwd_cl")
    return dx, part


def shuffle_norm_head_ok(x, c):
    """shapes the fused norm + head pair serves: C a multiple of 8 (4 for f32) on at most 64 lanes"""
    vm = 4 if x.dtype == torch.float32 else 8
    return x.is_cuda and c % vm == 0 and c // vm <= 64


def expand_norm_head_cl(x, w, ln_w, ln_b, head_w, head_b: float, p, eps=1e-5):
    """x (B,H,W,Cin) 16-bit, w (P*P*128, Cin) -> (B, H*P, W*P) f32: expand GEMM + pixel shuffle + LayerNorm(128) + head."""
    _dev(x, w, ln_w, ln_b, head_w)
    bb, h, wd, cin = x.shape
    if w.shape != (p * p * 128, cin) or w.dtype != x.dtype or ln_w.numel() != 128 or head_w.numel() != 128:
        raise TrambaHipError("expand_norm_head_cl: needs 128-channel groups and a (P*P*128, Cin) weight")
    y = torch.empty((bb, h * p, wd * p), dtype=torch.float32, device=x.device)
    _check(lib().tramba_expand_norm_head_cl(_ptr(x), _ptr(w), _ptr(ln_w), _ptr(ln_b), _ptr(head_w), float(head_b), _ptr(y),
                                            bb, h, wd, cin, p, eps, dt(x), _stream()), "expand_norm_head_cl")
    return y


EVAL_NINT, EVAL_NDBL = 520, 48


def saliency_stats(pred, gt):
    """pred (B, H, W) f32 = sigmoid(logits), gt (B, H, W) bool / u8 -> (ints (B, 520) i64, dbl (B, 48) f64) on the
    device: the per-image statistics behind MAE / F / E / S-measure (layout: include/tramba_hip.h)."""
    if pred.dtype != torch.float32 or pred.dim() != 3 or gt.shape != pred.shape:
        raise TrambaHipError("saliency_stats: pred must be (B, H, W) float32 and gt the same shape")
    pred = pred.contiguous()
    g8 = (gt if gt.dtype == torch.bool else gt != 0).contiguous().view(torch.uint8)
    _dev(pred, g8)
    b, h, w = pred.shape
    ints = torch.empty((b, EVAL_NINT), dtype=torch.int64, device=pred.device)
    dbl = torch.empty((b, EVAL_NDBL), dtype=torch.float64, device=pred.device)
    _check(lib().tramba_saliency_stats(_ptr(pred), _ptr(g8), _ptr(ints), _ptr(dbl), b, h, w, _stream()), "saliency_stats")
    return ints, dbl


def rowdot_cl(x, w, bias: float):
    """x: (..., C); w: (C) f32 -> (...) f32 = <x, w> + bias."""
    _dev(x, w)
    c = x.shape[-1]
    y = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    _check(lib().tramba_rowdot_cl(_ptr(x), _ptr(w), float(bias), _ptr(y), x.numel() // c, c, dt(x), _stream()), "rowdot_cl")
    return y


def rowdot_bwd_ok(x):
    """does the one-pass backward kernel of rowdot_cl serve this map? (rows of at most 64 x 16 bytes)"""
    c = x.shape[-1]
    return x.is_cuda and lib().tramba_rowdot_bwd_parts(x.numel() // c, c, dt(x)) > 0


def rowdot_bwd_cl(x, gy, w):
    """backward of rowdot_cl: x (..., C), gy (...) f32, w (C) f32 -> (gx like x, gw (C) f32, gb () f32)"""
    _dev(x, gy, w)
    c = x.shape[-1]
    rows = x.numel() // c
    if gy.dtype != torch.float32 or gy.numel() != rows:
        raise TrambaHipError("rowdot_bwd_cl: gy must be float32 with one entry per row")
    nparts = lib().tramba_rowdot_bwd_parts(rows, c, dt(x))
    if nparts <= 0:
        raise TrambaHipError(f"rowdot_bwd_cl: C={c} is not served")
    gx = torch.empty_like(x)
    part = torch.empty((nparts, c + 4), dtype=torch.float32, device=x.device)
    _check(lib().tramba_rowdot_bwd_cl(_ptr(x), _ptr(gy), _ptr(w), _ptr(gx), _ptr(part), rows, c, dt(x), _stream()),
           "rowdot_bwd_cl")
    s_ = slab_sum(part)
    return gx, s_[:c], s_[c]


def dw_pack(w, bias=None, w3=None, b3=None, w5=None, b5=None):
    """Reference-layout depth-wise weights (C,1,ks,ks) -> tap-major (ks*ks, C) f32 + bias (C) f32.
    With w3/b3/w5/b5 given (ks=7): the folded multi-scale stencil of DWMSMlp."""
    ts = [_f32(t) for t in (w, bias, w3, b3, w5, b5)]
    _dev(*ts)
    c, ks = w.shape[0], w.shape[-1]
    wt = torch.empty((ks * ks, c), dtype=torch.float32, device=w.device)
    bt = torch.empty((c,), dtype=torch.float32, device=w.device)
    _check(lib().tramba_dw_pack(*[_ptr(t) for t in ts], _ptr(wt), _ptr(bt), c, ks, _stream()), "dw_pack")
    return wt, bt


def dw_pack_multi_check(items):
    for it in items:
        _dev(*it)
        if any(t is not None and t.dtype != torch.float32 for t in it):
            raise TrambaHipError("dw_pack_multi: fp32 tensors only")
        c, ks = it[0].shape[0], it[0].shape[-1]
        if tuple(it[6].shape) != (ks * ks, c) or it[7].numel() != c:
            raise TrambaHipError("dw_pack_multi: output shapes do not match the weights")


def dw_pack_multi(items):
    """items: [(w, bias, w3, b3, w5, b5, wt, bt)] -- `dw_pack` of every item into its (preallocated) wt (ks*ks, C) / bt (C) f32,
    all in one launch per 40 items (tramba_dw_pack_multi); fp32 device tensors, None where dw_pack takes none."""
    if len(items) == 0:
        return
    dw_pack_multi_check(items)
    dw_pack_multi_run(dw_pack_multi_prepare(items, _checked=True))


def dw_pack_multi_prepare(items, _checked=False):
    """the host arrays of one dw_pack_multi call, built once for a fixed set of tensors (`dw_pack_multi_run` launches them)"""
    if not _checked:
        dw_pack_multi_check(items)
    n = len(items)
    cols = [(ctypes.c_void_p * n)(*[_ptr(it[k]) for it in items]) for k in range(8)]
    cs = (ctypes.c_int * n)(*[it[0].shape[0] for it in items])
    kss = (ctypes.c_int * n)(*[it[0].shape[-1] for it in items])
    return (cols, cs, kss, n, tuple(items))        # (the tensors stay referenced)


def dw_pack_multi_run(prepared):
    cols, cs, kss, n, _ = prepared
    _check(lib().tramba_dw_pack_multi(*cols, cs, kss, n, _stream()), "dw_pack_multi")


def dwconv_cl(x, wt, bt, act=ACT_NONE):
    """x: (B, H, W, C); wt: (ks*ks, C) f32 tap-major, bt: (C) f32 (see dw_pack)."""
    _dev(x, wt, bt)
    bb, h, wd, c = x.shape
    ks = int(round(wt.shape[0] ** 0.5))
    if wt.shape != (ks * ks, c) or bt.shape != (c,):
        raise TrambaHipError(f"dwconv_cl: packed weight {tuple(wt.shape)} does not match C={c}")
    y = torch.empty_like(x)
    _check(lib().tramba_dwconv_cl(_ptr(x), _ptr(wt), _ptr(bt), _ptr(y), bb, h, wd, c, ks, act, dt(x), _stream()),
           "dwconv_cl")
    return y


def dwconv_dual_cl(x, wt, bt, act=ACT_NONE, want_pre=True, flip=False):
    """(pre, act(pre)): the stencil's output before and after its activation from one launch; flip: mirrored taps (the
    stencil's adjoint: its input gradient)."""
    _dev(x, wt, bt)
    bb, h, wd, c = x.shape
    ks = int(round(wt.shape[0] ** 0.5))
    if wt.shape != (ks * ks, c) or bt.shape != (c,):
        raise TrambaHipError(f"dwconv_dual_cl: packed weight {tuple(wt.shape)} does not match C={c}")
    pre = torch.empty_like(x) if want_pre else None
    y = torch.empty_like(x)
    _check(lib().tramba_dwconv_dual_cl(_ptr(x), _ptr(wt), _ptr(bt), _ptr(pre), _ptr(y), bb, h, wd, c, ks, act, int(flip),
                                       dt(x), _stream()), "dwconv_dual_cl")
    return pre, y


def dw_unpack_grad(gwt, ks, multiscale=False, nbias=0, defer=False):
    """gwt (ks*ks + 1, C) f32 tap-major taps + bias row -> (g7 (C,1,ks,ks), g5, g3 (multi-scale fold, else None), gb (nbias, C)).
    defer: inside `deferred_sums()` the unpack is recorded and runs at `flush_sums()`, after the recorded sums (gwt may be
    the output of one), all unpacks of the pass in one launch -- for results that go straight to autograd as leaf gradients."""
    _dev(gwt)
    c = gwt.shape[1]
    g7 = torch.empty((c, 1, ks, ks), dtype=torch.float32, device=gwt.device)
    g5 = torch.empty((c, 1, 5, 5), dtype=torch.float32, device=gwt.device) if multiscale else None
    g3 = torch.empty((c, 1, 3, 3), dtype=torch.float32, device=gwt.device) if multiscale else None
    gb = torch.empty((nbias, c), dtype=torch.float32, device=gwt.device) if nbias else None
    if defer and _sumq.enabled:
        if gwt.dtype != torch.float32 or tuple(gwt.shape) != (ks * ks + 1, c):
            raise TrambaHipError("dw_unpack_grad: gradient table of another shape")
        if _sumq.poison:
            for t in (g7, g5, g3, gb):
                if t is not None:
                    t.fill_(float("nan"))
        with _sumq.lock:
            _sumq.unpacks.append((_ptr(gwt), _ptr(g7), _ptr(g5), _ptr(g3), _ptr(gb), int(nbias), int(c), int(ks), _stream(),
                                  (gwt, g7, g5, g3, gb)))
        # VIEWS go out, the queue keeps the bases: autograd hands a leaf the incoming gradient as it is (no copy) only when
        # nobody else references that tensor object -- a copy would read the buffer before the flush has filled it
        # (tests/test_gpu_grad.py NaN-fills deferred outputs to catch exactly that)
        return tuple(None if t is None else t.view(t.shape) for t in (g7, g5, g3, gb))
    _check(lib().tramba_dw_unpack_grad(_ptr(gwt), _ptr(g7), _ptr(g5), _ptr(g3), _ptr(gb), nbias, c, ks, _stream()),
           "dw_unpack_grad")
    return g7, g5, g3, gb


def upsample_bilinear_bwd(gout, h, w):
    """gout (..., H, W) f32 = gradient of F.interpolate(x (..., h, w), (H, W), mode="bilinear") -> gradient of x."""
    _dev(gout)
    gout = gout.contiguous()
    hh, ww = gout.shape[-2:]
    planes = gout.numel() // (hh * ww)
    gin = torch.empty(gout.shape[:-2] + (h, w), dtype=torch.float32, device=gout.device)
    if gout.dtype != torch.float32:
        raise TrambaHipError("upsample_bilinear_bwd: fp32 gradients only")
    _check(lib().tramba_upsample_bilinear_bwd(_ptr(gout), _ptr(gin), planes, h, w, hh, ww, _stream()), "upsample_bilinear_bwd")
    return gin


def _loss_nblk(npix):
    """workgroups per plane of tramba_sod_loss_sums: ~1024 pixels each"""
    return max(1, min(256, (npix + 1023) // 1024))


def sod_loss(outputs, label, weights=None):
    """The deep-supervision loss of train.py:76-85 (every output resized to the label, BCE-with-logits + IoU, summed with
    `weights`): outputs = fp32 (B, C, h_i, w_i) logit maps, label (B, C, H, W) f32.  Returns (loss 0-dim f32, coefs): coefs[i]
    (B*C, 4) feeds `sod_loss_grad`.  One launch per output + one finishing block."""
    _dev(label, *outputs)
    if label.dtype != torch.float32 or any(o.dtype != torch.float32 for o in outputs):
        raise TrambaHipError("sod_loss: fp32 logits and labels only")
    hh, ww = label.shape[-2:]
    planes = label.numel() // (hh * ww)
    nout = len(outputs)
    if not 0 < nout <= 8:
        raise TrambaHipError(f"sod_loss: 1..8 outputs, got {nout}")
    nblk = _loss_nblk(hh * ww)
    stream = _stream()
    parts = torch.empty((nout, planes, nblk, 3), dtype=torch.float32, device=label.device)
    coefs = torch.empty((nout, planes, 4), dtype=torch.float32, device=label.device)
    for i, o in enumerate(outputs):
        h, w = o.shape[-2:]
        if o.numel() != planes * h * w or h > hh or w > ww:
            raise TrambaHipError(f"sod_loss: output {i} {tuple(o.shape)} does not match the label {tuple(label.shape)}")
        _check(lib().tramba_sod_loss_sums(_ptr(o), _ptr(label), parts[i].data_ptr(), planes, h, w, hh, ww, nblk, stream),
               "sod_loss_sums")
    loss = torch.empty((), dtype=torch.float32, device=label.device)
    pp = (ctypes.c_void_p * nout)(*[parts[i].data_ptr() for i in range(nout)])
    cc = (ctypes.c_void_p * nout)(*[coefs[i].data_ptr() for i in range(nout)])
    nb = (ctypes.c_int * nout)(*([nblk] * nout))
    wt = None if weights is None else (ctypes.c_float * nout)(*[float(w) for w in weights])
    _check(lib().tramba_sod_loss_finish(pp, nb, wt, cc, nout, planes, hh * ww, _ptr(loss), stream), "sod_loss_finish")
    return loss, coefs


def sod_loss_grad(output, label, coef, gscale=None):
    """d loss / d output for one output of `sod_loss` (coef = its row of the coefficient table), times the device scalar
    `gscale` (the gradient arriving at the loss)."""
    _dev(output, label, coef, gscale)
    if gscale is not None and (gscale.dtype != torch.float32 or gscale.numel() != 1):
        raise TrambaHipError("sod_loss_grad: the incoming gradient must be one fp32 scalar")
    hh, ww = label.shape[-2:]
    h, w = output.shape[-2:]
    planes = label.numel() // (hh * ww)
    g = torch.empty_like(output)
    nbytes = lib().tramba_sod_loss_grad_workspace(planes, h, w, hh, ww)
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=output.device) if nbytes else None
    _check(lib().tramba_sod_loss_grad(_ptr(output), _ptr(label), _ptr(coef), _ptr(gscale), _ptr(g), _ptr(ws), nbytes, planes, h, w,
                                      hh, ww, _stream()), "sod_loss_grad")
    return g


def adam_step(params, grads, exp_avgs, exp_avg_sqs, steps, lr, beta1, beta2, eps, weight_decay=0.0):
    """One Adam step on lists of fp32 device tensors (tramba_adam_step); `steps` are the 0-dim fp32 device counters of
    torch.optim.Adam's capturable state."""
    n = len(params)
    if n == 0:
        return
    if not (len(grads) == len(exp_avgs) == len(exp_avg_sqs) == len(steps) == n):
        raise TrambaHipError("adam_step: the five tensor lists differ in length")
    for group in (params, grads, exp_avgs, exp_avg_sqs, steps):
        for t in group:
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                raise TrambaHipError("adam_step: contiguous fp32 tensors on a HIP device only (a sparse or strided gradient?)")
    for p, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs):
        if not (g.numel() == m.numel() == v.numel() == p.numel()):
            raise TrambaHipError("adam_step: gradient / state of another size than the parameter")
    adam_step_raw(pointer_array(params), pointer_array(grads), pointer_array(exp_avgs), pointer_array(exp_avg_sqs),
                  pointer_array(steps), (ctypes.c_int64 * n)(*[t.numel() for t in params]), n, lr, beta1, beta2, eps,
                  weight_decay)


def pointer_array(tensors):
    """host array of the tensors' device addresses, as the by-value multi-tensor entries take them"""
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def adam_step_raw(params, grads, exp_avgs, exp_avg_sqs, steps, numel, n, lr, beta1, beta2, eps, weight_decay=0.0):
    """`adam_step` on prepared pointer arrays (tramba_amd.train.Adam keeps those of the parameters and of the state between
    steps and rebuilds only the gradients'); the caller vouches for fp32, contiguous, same-size device tensors."""
    _check(lib().tramba_adam_step(params, grads, exp_avgs, exp_avg_sqs, steps, numel, n, float(lr), float(beta1), float(beta2),
                                  float(eps), float(weight_decay), _stream()), "adam_step")


def im2col3x3_cl(x, stride, pad, ckp):
    """x (B,H,W,C) -> cols (B*Ho*Wo, ckp), column (ky*3 + kx)*C + ci, zeros for padding taps / alignment columns."""
    _dev(x)
    bb, h, wd, c = x.shape
    ho, wo = (h + 2 * pad - 3) // stride + 1, (wd + 2 * pad - 3) // stride + 1
    cols = torch.empty((bb * ho * wo, ckp), dtype=x.dtype, device=x.device)
    _check(lib().tramba_im2col3x3_cl(_ptr(x), _ptr(cols), bb, h, wd, c, stride, pad, ckp, dt(x), _stream()), "im2col3x3_cl")
    return cols


def col2im3x3_cl(gcols, shape, stride, pad):
    """gcols (B*Ho*Wo, CKp) -> gx of `shape` = (B,H,W,C): the adjoint of im2col3x3_cl."""
    _dev(gcols)
    bb, h, wd, c = shape
    gx = torch.empty(shape, dtype=gcols.dtype, device=gcols.device)
    _check(lib().tramba_col2im3x3_cl(_ptr(gcols), _ptr(gx), bb, h, wd, c, stride, pad, gcols.shape[1], dt(gcols), _stream()),
           "col2im3x3_cl")
    return gx


def dwconv_wgrad_cl(x, gy, ks):
    """x, gy: (B, H, W, C) -> (gw (ks*ks, C) f32 tap-major, gb (C) f32)."""
    _dev(x, gy)
    bb, h, wd, c = x.shape
    if gy.shape != x.shape or gy.dtype != x.dtype:
        raise TrambaHipError("dwconv_wgrad_cl: x / gy mismatch")
    part = torch.empty((lib().tramba_dwconv_wgrad_parts(bb, h, wd, c, ks), ks * ks + 1, c), dtype=torch.float32, device=x.device)
    _check(lib().tramba_dwconv_wgrad_cl(_ptr(x), _ptr(gy), _ptr(part), bb, h, wd, c, ks, dt(x), _stream()),
           "dwconv_wgrad_cl")
    s = slab_sum(part)
    return s[:ks * ks], s[ks * ks]


def dwconv_wgrad_table(x, gy, ks, defer=False):
    """the same as one (ks*ks + 1, C) f32 table (taps, then the bias row): the input of dw_unpack_grad (defer: the slab sum is
    recorded inside `deferred_sums()` -- only for a table that nothing but a deferred unpack reads)"""
    _dev(x, gy)
    bb, h, wd, c = x.shape
    if gy.shape != x.shape or gy.dtype != x.dtype:
        raise TrambaHipError("dwconv_wgrad_cl: x / gy mismatch")
    part = torch.empty((lib().tramba_dwconv_wgrad_parts(bb, h, wd, c, ks), ks * ks + 1, c), dtype=torch.float32, device=x.device)
    _check(lib().tramba_dwconv_wgrad_cl(_ptr(x), _ptr(gy), _ptr(part), bb, h, wd, c, ks, dt(x), _stream()),
           "dwconv_wgrad_cl")
    return slab_sum(part, defer=defer)


def dct_split_cl(x, wx, wy):
    """x: (B, n, n, C) -> (high, low), each (B, n/2, n/2, C)."""
    _dev(x, wx, wy)
    bb, n, n2, c = x.shape
    if n != n2 or n % 2:
        raise TrambaHipError(f"dct_split_cl: need an even square map, got {n}x{n2}")
    tmp = torch.empty((bb, n, n, c), dtype=torch.float32, device=x.device)
    high = torch.empty((bb, n // 2, n // 2, c), dtype=x.dtype, device=x.device)
    low = torch.empty_like(high)
    _check(lib().tramba_dct_split_cl(_ptr(x), _ptr(wx), _ptr(wy), _ptr(tmp), _ptr(high), _ptr(low), bb, n, c,
                                     dt(x), _stream()), "dct_split_cl")
    return high, low


def _check_epilogue(what, bias, residual, dtype, m, n):
    """the kernels read `bias` as (N) fp32 and `residual` as (M, N) in the INPUT dtype: anything else would be misread"""
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != n):
        raise TrambaHipError(f"{what}: bias must be float32 with {n} elements, got {bias.dtype} x {bias.numel()}")
    if residual is not None and (residual.dtype != dtype or residual.numel() != m * n):
        raise TrambaHipError(f"{what}: residual must be {dtype} with {m} x {n} elements, got {residual.dtype} x "
                             f"{residual.numel()}")


def linear_cl(x, w, bias=None, residual=None, act=ACT_NONE, out_dtype=None):
    """x: (..., K); w: (N, K) same dtype; -> (..., N)."""
    _dev(x, w, bias, residual)
    k = x.shape[-1]
    n = w.shape[0]
    m = x.numel() // k
    out_dtype = x.dtype if out_dtype is None else out_dtype
    y = torch.empty(x.shape[:-1] + (n,), dtype=out_dtype, device=x.device)
    if w.dtype != x.dtype or w.shape[1] != k:
        raise TrambaHipError("linear_cl: weight dtype/shape mismatch")
    _check_epilogue("linear_cl", bias, residual, x.dtype, m, n)
    _check(lib().tramba_linear_cl(_ptr(x), _ptr(w), _ptr(bias), _ptr(residual), _ptr(y), m, n, k, act, dt(x),
                                  dt(y), _stream()), "linear_cl")
    return y


def linear_dual_ok(x, w):
    """can linear_dual_cl take this GEMM? (16-bit operands, whole 64-deep K steps, 16-byte rows)"""
    return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and w.dtype == x.dtype and x.shape[-1] % 64 == 0
            and w.shape[0] % 8 == 0 and x.is_contiguous() and w.is_contiguous())


def linear_dual_cl(x, w, bias, act):
    """x: (..., K); w: (N, K) -> (pre, act(pre)), both (..., N) in x's dtype, from one launch."""
    _dev(x, w, bias)
    k, n = x.shape[-1], w.shape[0]
    m = x.numel() // k
    if not linear_dual_ok(x, w) or w.shape[1] != k:
        raise TrambaHipError("linear_dual_cl: needs 16-bit contiguous operands with K % 64 == 0 and N % 8 == 0")
    _check_epilogue("linear_dual_cl", bias, None, x.dtype, m, n)
    y_pre = torch.empty(x.shape[:-1] + (n,), dtype=x.dtype, device=x.device)
    y_act = torch.empty_like(y_pre)
    _check(lib().tramba_linear_dual_cl(_ptr(x), _ptr(w), _ptr(bias), _ptr(y_pre), _ptr(y_act), m, n, k, act, dt(x), _stream()),
           "linear_dual_cl")
    return y_pre, y_act


def linear_ln_cl(x, w_folded, colsum, bias, eps, residual=None, act=ACT_NONE, out_dtype=None):
    """act(LayerNorm(x) @ W^T + b) in one launch: w_folded = W * gamma (N, K) in x's dtype, colsum (N) f32 its row sums,
    bias (N) f32 = W @ beta + b (see include/tramba_hip.h)."""
    _dev(x, w_folded, colsum, bias, residual)
    k = x.shape[-1]
    n = w_folded.shape[0]
    m = x.numel() // k
    if w_folded.dtype != x.dtype or w_folded.shape[1] != k or colsum.dtype != torch.float32 or colsum.numel() != n:
        raise TrambaHipError("linear_ln_cl: operand shapes / dtypes do not match")
    _check_epilogue("linear_ln_cl", bias, residual, x.dtype, m, n)
    out_dtype = x.dtype if out_dtype is None else out_dtype
    y = torch.empty(x.shape[:-1] + (n,), dtype=out_dtype, device=x.device)
    _check(lib().tramba_linear_ln_cl(_ptr(x), _ptr(w_folded), _ptr(colsum), _ptr(bias), _ptr(residual), _ptr(y), m, n, k,
                                     float(eps), act, dt(x), dt(y), _stream()), "linear_ln_cl")
    return y


def linear2_cl(x1, x2, w, bias=None, residual=None, act=ACT_NONE, out_dtype=None):
    """Linear2d on torch.cat((x1, x2), dim=-1) without the concatenation: x1 (..., K1), x2 (..., K2), w (N, K1+K2)."""
    _dev(x1, x2, w, bias, residual)
    k1, k2 = x1.shape[-1], x2.shape[-1]
    n = w.shape[0]
    m = x1.numel() // k1
    if x2.shape[:-1] != x1.shape[:-1] or w.shape[1] != k1 + k2 or w.dtype != x1.dtype or x2.dtype != x1.dtype:
        raise TrambaHipError("linear2_cl: operand shapes / dtypes do not match")
    _check_epilogue("linear2_cl", bias, residual, x1.dtype, m, n)
    out_dtype = x1.dtype if out_dtype is None else out_dtype
    y = torch.empty(x1.shape[:-1] + (n,), dtype=out_dtype, device=x1.device)
    _check(lib().tramba_linear2_cl(_ptr(x1), _ptr(x2), k1, _ptr(w), _ptr(bias), _ptr(residual), _ptr(y), m, n, k1 + k2,
                                   act, dt(x1), dt(y), _stream()), "linear2_cl")
    return y


def wgrad_cl(gy, x, want_bias=False, defer=False):
    """Weight gradient of y = x @ W^T: gy (..., N), x (..., K) 16-bit, same leading shape -> (gw (N, K) f32, gb (N) f32 or
    None).  Rows may be strided (a column slice of a wider tensor) as long as the last dim is contiguous.  defer: inside
    `deferred_sums()` the slab sum of the launch is recorded for `flush_sums()` (see _SumQueue)."""
    n, k = gy.shape[-1], x.shape[-1]
    gy2, x2 = gy.reshape(-1, n), x.reshape(-1, k)
    if gy2.stride(-1) != 1 or x2.stride(-1) != 1:
        gy2, x2 = gy2.contiguous(), x2.contiguous()
    m = gy2.shape[0]
    if x2.shape[0] != m or gy2.dtype != x2.dtype:
        raise TrambaHipError("wgrad_cl: gy / x mismatch")
    return _wgrad(gy2, x2, m, n, k, 1, 1, 0, 0, gy2.stride(0), 0, 0, x2.stride(0), want_bias, defer)


def wgrad_rows_cl(gy, x, segments, defer=False):
    """Rows of the weight gradient gy^T x (gy (M, N), x (M, K) 16-bit) in another row order: segments = [(first row, rows)],
    the result holds those row ranges one after the other ((sum of rows, K) f32).  Inside `deferred_sums()` (defer=True) every
    range is summed from the GEMM's partial slabs straight into its place at `flush_sums()` -- no full-size sum, no
    concatenation; otherwise the ranges are cut out of the finished gradient."""
    _dev(gy, x)
    m, n = gy.shape
    k = x.shape[-1]
    if x.shape[0] != m or gy.dtype != x.dtype:
        raise TrambaHipError("wgrad_rows_cl: gy / x mismatch")
    if any(a < 0 or c <= 0 or a + c > n for a, c in segments):
        raise TrambaHipError("wgrad_rows_cl: a row range outside the gradient")
    total = sum(c for _, c in segments)
    if defer and _sumq.enabled:
        ws_bytes = lib().tramba_wgrad_workspace(m, n, k, 1, 1)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=gy.device)
        full = torch.empty((1, n * k + n), dtype=torch.float32, device=gy.device)
        nslab = ctypes.c_int(0)
        _check(lib().tramba_wgrad_parts_cl(_ptr(gy), _ptr(x), _ptr(full), _ptr(ws), ws_bytes, m, n, k, 1, 1, 0, 0, n, 0, 0, k, 0,
                                           dt(gy), _stream(), ctypes.byref(nslab)), "wgrad_parts_cl")
        if nslab.value > 0:
            out = torch.empty((total, k), dtype=torch.float32, device=gy.device)
            if _sumq.poison:
                out.fill_(float("nan"))
            row = 0
            for a, c in segments:
                _enqueue_sum(_ptr(ws) + a * k * 4, _ptr(out) + row * k * 4, c * k, nslab.value, (ws, out), stride=n * k + n)
                row += c
            return out.view(total, k)       # (a view goes out, the queue keeps the base: see dw_unpack_grad)
        gw = full[0, :n * k].view(n, k)     # a single slab: the gradient is already there
    else:
        gw = wgrad_cl(gy, x)[0]
    return torch.cat([gw[a:a + c] for a, c in segments], dim=0)


def wgrad_grouped_cl(gy, x, defer=False):
    """gy (B, G, L, N), x (B, G, L, K) 16-bit contiguous -> (G, N, K) f32: per group g, sum over b and l of gy^T x.
    defer: see wgrad_cl (inside `deferred_sums()` the slab sums are recorded; for results handed to autograd as they are)."""
    _dev(gy, x)
    b, g, l, n = gy.shape
    k = x.shape[-1]
    if x.shape[:3] != gy.shape[:3] or gy.dtype != x.dtype:
        raise TrambaHipError("wgrad_grouped_cl: gy / x mismatch")
    out, _ = _wgrad(gy, x, l, n, k, g, b, g * l * n, l * n, n, g * l * k, l * k, k, False, defer)
    return out


def tn_shared_cl(a, x):
    """out[g] = a^T @ x[g]: a (T, N) 16-bit shared by every group, x (G, T, K) same dtype -> (G, N, K) f32 -- a separable
    transform along a leading axis of a channels-last tensor (the DCT backward: T = coefficients, N = pixels)."""
    _dev(a, x)
    nb = a.shape[0] if a.dim() == 3 else 1        # a (P, T, N): out[g] = sum over p of a[p]^T @ x[g] (a table kept in pieces)
    t, n = a.shape[-2:]
    g, t2, k = x.shape
    if t2 != t or a.dtype != x.dtype:
        raise TrambaHipError("tn_shared_cl: operand shapes / dtypes do not match")
    out, _ = _wgrad(a, x, t, n, k, g, nb, t * n, 0, n, 0, t * k, k, False)
    return out if g > 1 else out.view(1, n, k)


def _wgrad(gy, x, m, n, k, groups, nbatch, gy_bs, gy_gs, gy_ld, x_bs, x_gs, x_ld, want_bias, defer=False):
    for t in (gy, x):
        if not t.is_cuda:
            raise TrambaHipError("tramba_amd kernels need tensors on a HIP device (no CPU fallback)")
    ws_bytes = lib().tramba_wgrad_workspace(m, n, k, groups, nbatch)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=gy.device)
    out = torch.empty((groups, n * k + n), dtype=torch.float32, device=gy.device)
    if defer and _sumq.enabled:
        nslab = ctypes.c_int(0)
        _check(lib().tramba_wgrad_parts_cl(_ptr(gy), _ptr(x), _ptr(out), _ptr(ws), ws_bytes, m, n, k, groups, nbatch, gy_bs,
                                           gy_gs, gy_ld, x_bs, x_gs, x_ld, int(want_bias), dt(gy), _stream(),
                                           ctypes.byref(nslab)), "wgrad_parts_cl")
        if nslab.value > 0:          # (0: a single slab, already in `out`)
            slab = n * k + n
            if groups > 1 and not want_bias:
                # per-group gradients without the bias slot between them: summed into a DENSE (groups, N, K) tensor -- a leaf
                # takes an incoming gradient as it is only when its strides are the parameter's own (a copy would read the
                # buffer before the flush has filled it)
                dense = torch.empty((groups, n, k), dtype=torch.float32, device=gy.device)
                if _sumq.poison:
                    dense.fill_(float("nan"))
                for g in range(groups):
                    _enqueue_sum(_ptr(ws) + g * nslab.value * slab * 4, _ptr(dense) + g * n * k * 4, n * k, nslab.value,
                                 (ws, dense), stride=slab)
                return dense.view(groups, n, k), None
            if _sumq.poison:
                out.fill_(float("nan"))
            for g in range(groups):
                _enqueue_sum(_ptr(ws) + g * nslab.value * slab * 4, _ptr(out) + g * slab * 4, slab, nslab.value, (ws, out))
    else:
        _check(lib().tramba_wgrad_cl(_ptr(gy), _ptr(x), _ptr(out), _ptr(ws), ws_bytes, m, n, k, groups, nbatch, gy_bs, gy_gs,
                                     gy_ld, x_bs, x_gs, x_ld, int(want_bias), dt(gy), _stream()), "wgrad_cl")
    gw = out[:, :n * k].view(groups, n, k)
    if groups == 1:
        return gw[0], (out[0, n * k:] if want_bias else None)
    return gw, None


def rows_gemm_cl(x, w, y, n):
    """y[z, t, :n] = x[z, t, :] @ w[z % G]^T: x (Z, M, K) 16-bit, w (G, n, K) same dtype, y (Z, M, ldy) f32 (in place)."""
    _dev(x, w, y)
    z, m, k = x.shape
    if w.shape[1:] != (n, k) or w.dtype != x.dtype or y.dtype != torch.float32 or y.shape[:2] != (z, m):
        raise TrambaHipError("rows_gemm_cl: operand shapes / dtypes do not match")
    _check(lib().tramba_rows_gemm_cl(_ptr(x), _ptr(w), _ptr(y), z, m, n, k, w.shape[0], y.shape[-1], dt(x), _stream()),
           "rows_gemm_cl")
    return y


class _SumQueue:
    """Pending partial-sum reductions of a training step (tramba_multi_sum).  The parameter gradients of LayerNorm and of
    every Linear2d leave their kernels as partial tables (one row per workgroup / token split) that a small kernel adds up in
    a fixed order.  Nothing reads those gradients before the optimizer (reference train.py:86-89), so inside
    `deferred_sums()` a reduction is only RECORDED -- the caller gets the output tensor, to be filled later -- and all of them
    run as a handful of launches at `flush_sums()`: ~300 launches of 4-7 us per step, each on its launch floor, become ~10.
    Off by default: outside the context every sum runs where it is issued.  Only call sites whose result goes straight to
    autograd as a leaf's gradient pass defer=True (anything that READS the sum must not)."""

    def __init__(self):
        self.enabled = False
        self.poison = False         # tests: NaN-fill a deferred output, so that a premature reader cannot go unnoticed
        self.items = []
        self.unpacks = []           # depth-wise gradient unpacks (tramba_dw_unpack_grad_multi), run after the sums
        self.lock = threading.Lock()


_sumq = _SumQueue()


class deferred_sums:
    """with hip.deferred_sums(): loss.backward()   -- record the deferrable partial-sum reductions, run them at the exit"""

    def __enter__(self):
        self._was = _sumq.enabled
        _sumq.enabled = True
        return self

    def __exit__(self, *exc):
        _sumq.enabled = self._was
        if not self._was:
            flush_sums()
        return False


def _enqueue_sum(part_ptr, out_ptr, n, nslab, keep, stride=None):
    with _sumq.lock:
        _sumq.items.append((part_ptr, out_ptr, int(n), int(nslab), _stream(), keep, int(n if stride is None else stride)))


def flush_sums():
    """Run every recorded reduction (one tramba_multi_sum call per stream they were recorded on)."""
    with _sumq.lock:
        items, _sumq.items = _sumq.items, []
        unpacks, _sumq.unpacks = _sumq.unpacks, []
    by_stream = {}
    for it in items:
        by_stream.setdefault(it[4], []).append(it)
    for stream, its in by_stream.items():
        cnt = len(its)
        parts = (ctypes.c_void_p * cnt)(*[it[0] for it in its])
        outs = (ctypes.c_void_p * cnt)(*[it[1] for it in its])
        ns = (ctypes.c_int64 * cnt)(*[it[2] for it in its])
        nsl = (ctypes.c_int * cnt)(*[it[3] for it in its])
        strides = (ctypes.c_int64 * cnt)(*[it[6] for it in its])
        _check(lib().tramba_multi_sum_strided(parts, outs, ns, strides, nsl, cnt, stream), "multi_sum")
    streams = set(by_stream)
    by_stream = {}
    for it in unpacks:          # (a table and its unpack are recorded on the same stream: the sums above come first)
        by_stream.setdefault(it[8], []).append(it)
    for stream, its in by_stream.items():
        cnt = len(its)
        cols = [(ctypes.c_void_p * cnt)(*[it[k] for it in its]) for k in range(5)]
        ints = [(ctypes.c_int * cnt)(*[it[k] for it in its]) for k in (5, 6, 7)]
        _check(lib().tramba_dw_unpack_grad_multi(*cols, *ints, cnt, stream), "dw_unpack_grad_multi")
    streams |= set(by_stream)
    # sums recorded on ANOTHER stream (the guide branches' backward runs on the side stream of models._forward_overlapped) were
    # launched there: whatever the caller's stream does next -- the optimizer reads these gradients -- must come after them
    if not streams:     # (nothing was recorded: e.g. a host-tensor model, which has no stream to ask about)
        return
    cur = _stream()
    others = [st for st in streams if st != cur]
    if others:
        here = torch.cuda.current_stream()
        for st in others:
            ext = torch.cuda.ExternalStream(st)
            ev = torch.cuda.Event()
            ev.record(ext)
            here.wait_event(ev)


def pending_sums():
    return len(_sumq.items) + len(_sumq.unpacks)


def slab_sum(part, defer=False):
    """part (S, ...) f32 -> sum over the first axis, slabs added in a fixed order (tramba_slab_sum); the small partial-sum
    tables of the backward kernels.  defer: inside `deferred_sums()` the sum is recorded and runs at `flush_sums()` (the
    returned tensor is filled then) -- for results that go straight to autograd as a parameter's gradient."""
    _dev(part)
    nslab = part.shape[0]
    n = part.numel() // max(nslab, 1)
    if part.dtype != torch.float32 or not part.is_contiguous() or n % 4 or nslab == 0:
        return part.sum(dim=0)
    out = torch.empty(part.shape[1:], dtype=torch.float32, device=part.device)
    if defer and _sumq.enabled:
        if _sumq.poison:
            out.fill_(float("nan"))
        _enqueue_sum(_ptr(part), _ptr(out), n, nslab, (part, out))   # (both stay allocated until the sum has run)
        return out
    _check(lib().tramba_slab_sum(_ptr(part), _ptr(out), n, nslab, _stream()), "slab_sum")
    return out


def shadow_cast_multi(table, ntensors, total_tiles, dtype):
    """One launch over a device-resident table (ntensors, 8) int64 {src f32, dst, dst_t, rows, cols, first_tile, dst_ld,
    dst_t_ld}: every dst = src cast to `dtype`, every dst_t = its transpose, with leading dimensions (blocks of padded
    layouts); tramba_amd.modules.refresh_lowp_shadows builds the table."""
    _dev(table)
    if table.dtype != torch.int64 or table.shape != (ntensors, 8) or not table.is_contiguous():
        raise TrambaHipError("shadow_cast_multi: table must be a contiguous (ntensors, 8) int64 tensor")
    _check(lib().tramba_shadow_cast_multi(_ptr(table), ntensors, total_tiles, _DT[dtype], _stream()), "shadow_cast_multi")


def conv3x3s2_cl(x, w_kmajor, bias):
    """x: (B, H, W, Cin) bf16/f16; w_kmajor: (Cout, 9*Cin) = weight.permute(0,2,3,1), same dtype."""
    _dev(x, w_kmajor, bias)
    bb, h, wd, cin = x.shape
    cout = w_kmajor.shape[0]
    if w_kmajor.dtype != x.dtype or w_kmajor.shape[1] != 9 * cin:
        raise TrambaHipError("conv3x3s2_cl: weight dtype/shape mismatch")
    y = torch.empty((bb, (h + 1) // 2, (wd + 1) // 2, cout), dtype=x.dtype, device=x.device)
    _check(lib().tramba_conv3x3s2_cl(_ptr(x), _ptr(w_kmajor), _ptr(bias), _ptr(y), bb, h, wd, cin, cout, dt(x),
                                     _stream()), "conv3x3s2_cl")
    return y


def stem_conv_ln_gelu(img, w, bias, ln_w, ln_b, eps, out_dtype):
    """img: (B, 3, H, W) NCHW f32 or out_dtype -> (B, H/2, W/2, 64) out_dtype."""
    _dev(img, w, bias, ln_w, ln_b)
    bb, c3, h, wd = img.shape
    if c3 != 3 or tuple(w.shape) != (64, 3, 3, 3):
        raise TrambaHipError("stem_conv_ln_gelu: expects a 3 -> 64 channel 3x3 stem")
    y = torch.empty((bb, (h + 1) // 2, (wd + 1) // 2, 64), dtype=out_dtype, device=img.device)
    _check(lib().tramba_stem_conv_ln_gelu(_ptr(img), _ptr(w), _ptr(bias), _ptr(ln_w), _ptr(ln_b), _ptr(y), bb, h, wd,
                                          eps, _DT[img.dtype], _DT[out_dtype], _stream()), "stem_conv_ln_gelu")
    return y
