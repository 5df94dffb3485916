"""Host-side mirror of the reference's nn.Module surface for the Tramba hot path.

Same class names, constructor arguments, parameter/buffer names and shapes as the reference
(state_dict-compatible: 679 entries for Tramba-V), same ``forward(x:(B,C,H,W)) -> (B,C,H,W)``
contracts -- but a different machine underneath:

* activations live CHANNELS-LAST ((B, H, W, C) contiguous) between modules; a module's public
  ``forward`` accepts/returns NCHW-shaped tensors whose memory is channels-last, so chaining
  modules never copies.  ``_forward_cl`` is the internal entry working on (B,H,W,C).
* with autograd off (inference) every op on the path is a hand-written HIP kernel from
  libtramba_hip (tramba_amd/hip.py); the SS2D core runs fused (no (B,K,D,L) gather, no `delta`
  tensor, merge + out_norm + GELU in one kernel).
* with autograd on (training) the SS2D core runs through the reference's own plugin path --
  ``scan.apply -> grouped projections -> SelectiveScanOflex.apply -> merge.apply`` -- whose
  gather/merge and selective-scan forward AND backward are HIP kernels; the surrounding
  LayerNorm / depth-wise conv / GELU use stock PyTorch-ROCm ops for their autograd.
* there is no CPU path: a CPU tensor raises.

Reference files mirrored: Models/modules.py:10-27,134-153,183-274,678-696; Models/vmamba.py:18-323
(SS2D), :327-396 (VSSBlock), :399-518 (VSSMEncoder), :595-704 (DWConv, DWMSMlp,
MultiScaleDecoderBlock); Models/DCT_2D.py; Models/freq_mamba.py; Models/mamba_init.py.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip
from .ops import (CrossMerge, CrossMerge_Dilation, CrossMerge_Line, CrossMerge_Window, CrossScan,
                  CrossScan_Dilation, CrossScan_Line, CrossScan_Window, SelectiveScanOflex)

# One implementation per op: every kernel below is libtramba_hip; there are no environment switches and no alternative
# backends.  The reference's scan / merge PLUGIN API (SS2D(scan=, merge=)) is the only other route through SS2D, taken when
# a caller supplies scan classes of its own (scripts/ab_forward.py monkey-patches module attributes for A/B timing).


# LayerNorm2d -> Linear2d pairs as one launch in 16-bit inference (a scheduling choice of the same arithmetic: A/B timing
# scripts flip it, scripts/ab_forward.py tramba_amd.modules.FOLD_LAYERNORM)
FOLD_LAYERNORM = True


def to_cl(x: torch.Tensor) -> torch.Tensor:
    """NCHW-shaped -> (B,H,W,C) contiguous (free when x is already channels-last in memory)."""
    return x.permute(0, 2, 3, 1).contiguous()


def from_cl(x: torch.Tensor) -> torch.Tensor:
    """(B,H,W,C) contiguous -> NCHW-shaped view with channels-last strides."""
    return x.permute(0, 3, 1, 2)


def _need_device(x):
    if not x.is_cuda:
        raise RuntimeError("tramba_amd runs on a HIP device only (MI355X); got a CPU tensor. "
                           "There is deliberately no CPU fallback.")


def _infer(*tensors) -> bool:
    """True when no autograd graph is needed -> all-HIP inference kernels."""
    if not torch.is_grad_enabled():
        return True
    return not any(t is not None and t.requires_grad for t in tensors)


def _f32(p):
    return hip._f32(p)


class _PackCache:
    """Derived, kernel-ready copies of parameters (packed stencils, padded projections, -exp(A)),
    rebuilt only when a source parameter changes (version counter / storage / dtype)."""

    def __init__(self):
        self._store = {}

    def get(self, name, params, builder):
        key = tuple((p._version, p.data_ptr(), p.dtype, str(p.device)) for p in params if p is not None)
        hit = self._store.get(name)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, builder())
            self._store[name] = hit
        return hit[1]


def _cache(module) -> _PackCache:
    c = module.__dict__.get("_tramba_cache")
    if c is None:
        c = module.__dict__["_tramba_cache"] = _PackCache()
    return c


def _act_torch(x, act):
    if act == hip.ACT_GELU:
        return F.gelu(x)
    if act == hip.ACT_SILU:
        return F.silu(x)
    return x


def _split16(t):
    """fp32 -> (hi, lo) bf16 with hi + lo == t to 2^-17 relative: the 16-bit matrix-core kernels serve the fp32 validation
    mode by three passes (hi.hi + hi.lo + lo.hi)."""
    hi = t.to(torch.bfloat16)
    return hi, (t - hi.float()).to(torch.bfloat16)


def _wgrad(gy2, x2, want_bias=False, defer=False):
    """Weight (and bias) gradient of y = x @ W^T: gy2 (M, N), x2 (M, K) -> gw (N, K) f32 [, gb (N) f32], on
    tramba_wgrad_cl (token-split TN GEMM, transposed LDS reads).  defer: the result goes straight to autograd as the
    parameter's gradient -- inside hip.deferred_sums() its slab sum joins the step's batched reduction (hip._SumQueue)."""
    if gy2.dtype != torch.float32:
        return hip.wgrad_cl(gy2, x2, want_bias, defer)
    gh, gl = _split16(gy2)
    xh, xl = _split16(x2)
    gw = hip.wgrad_cl(gh, xh)[0] + hip.wgrad_cl(gh, xl)[0] + hip.wgrad_cl(gl, xh)[0]
    return gw, (gy2.sum(0) if want_bias else None)


def _dgrad(gy2, wa, wa_t=None):
    """Input gradient of y = x @ W^T: gy2 (M, N) @ wa (N, K) on the forward GEMM kernel, which wants the weight transposed
    (wa_t (K, N): the shadow kept for that purpose, or a transpose made here)."""
    return hip.linear_cl(gy2, wa.t().contiguous() if wa_t is None else wa_t)


class _ConvIm2colCL(torch.autograd.Function):
    """A dense convolution of the TRAINING path on a channels-last map with autograd (the 3x3 / stride-2 convs of the
    VMamba stem and downsample layers, vmamba.py:454,481,486), all on the library: im2col / col2im
    (tramba_im2col3x3_cl / tramba_col2im3x3_cl: one pass each over the column matrix, in the k-major column order
    (ky, kx, ci), no transposed or padded copies) around the GEMMs -- forward and input gradient on tramba_linear_cl, weight /
    bias gradient on tramba_wgrad_cl.  x (B,H,W,Cin), w (Cout,Cin,3,3), b (Cout) or None -> (B,Ho,Wo,Cout)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding):
        bsz, h, wd, cin = x.shape
        cout, _, kh, kw = w.shape
        if (kh, kw) != (3, 3) or stride[0] != stride[1] or padding[0] != padding[1]:
            raise NotImplementedError("the training path of a dense convolution is built for 3x3 kernels with one stride / padding")
        st, pd = int(stride[0]), int(padding[0])
        ck = cin * 9
        ckp = (ck + 7) // 8 * 8                                                                # 16-byte rows for the kernels
        ct = hip.im2col3x3_cl(x.contiguous(), st, pd, ckp)                                     # (B*L, CKp), columns (ky, kx, ci)
        w2 = w.detach().permute(0, 2, 3, 1).reshape(cout, ck).to(x.dtype)                      # the same column order
        if ckp != ck:
            w2 = F.pad(w2, (0, ckp - ck))
        ho, wo = (h + 2 * pd - 3) // st + 1, (wd + 2 * pd - 3) // st + 1
        y = hip.linear_cl(ct, w2.contiguous(), None if b is None else b.detach().float().contiguous())
        ctx.save_for_backward(ct, w2)
        ctx.cfg = (st, pd, tuple(x.shape), cin, ck, b is not None, w.dtype)
        return y.view(bsz, ho, wo, cout)

    @staticmethod
    def backward(ctx, gy):
        ct, w2 = ctx.saved_tensors
        st, pd, xshape, cin, ck, has_bias, wdtype = ctx.cfg
        gy2 = gy.reshape(-1, gy.shape[-1]).contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = hip.col2im3x3_cl(_dgrad(gy2, w2), xshape, st, pd)
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            gw, gb = _wgrad(gy2, ct, has_bias)
            gw = gw[:, :ck].reshape(-1, 3, 3, cin).permute(0, 3, 1, 2).contiguous().to(wdtype)   # back to (Cout, Cin, 3, 3)
        return gx, gw, gb, None, None


class _DwConvCL(torch.autograd.Function):
    """Depth-wise stencil on a channels-last map with autograd, all HIP (training path): forward
    tramba_dwconv_cl; input gradient = the same kernel with the taps flipped; weight / bias gradient
    tramba_dwconv_wgrad_cl.  wt (ks*ks, C) f32 tap-major, bt (C) f32."""

    @staticmethod
    def forward(ctx, x, wt, bt):
        wt = wt.contiguous()
        ctx.save_for_backward(x, wt)
        return hip.dwconv_cl(x, wt, bt.contiguous(), hip.ACT_NONE)

    @staticmethod
    def backward(ctx, gy):
        x, wt = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = hip.dwconv_cl(gy, wt.flip(0).contiguous(), _zeros_const(wt[0].shape, wt.dtype, wt.device), hip.ACT_NONE)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            ks = int(round(wt.shape[0] ** 0.5))
            gw, gb = hip.dwconv_wgrad_cl(x, gy, ks)
        return gx, gw, gb


def _tap_major(conv: nn.Conv2d):
    """(C,1,ks,ks) / (C) parameters -> tap-major (ks*ks, C) + bias (C), fp32, differentiable."""
    w = conv.weight.float()
    c = w.shape[0]
    wt = w.reshape(c, -1).t()
    bt = conv.bias.float() if conv.bias is not None else _zeros_const((c,), torch.float32, w.device)
    return wt, bt


def _dwconv_train_cl(x_cl, conv: nn.Conv2d):
    """(B,H,W,C) -> (B,H,W,C), training path of a depth-wise conv."""
    wt, bt = _tap_major(conv)
    return _DwConvCL.apply(x_cl.contiguous(), wt, bt)


_centre_tap_cache = {}


def _centre_tap7(device, dtype):
    """The identity tap of the folded 7x7 stencil: a constant built once per device (writing one element from the host is
    a synchronous copy -- two launches per call, and not allowed while a stream is being captured into a hipGraph)."""
    key = (device, dtype)
    t = _centre_tap_cache.get(key)
    if t is None:
        host = torch.zeros(1, 1, 7, 7, dtype=dtype)
        host[0, 0, 3, 3] = 1.0
        t = _centre_tap_cache[key] = host.to(device)
    return t


def _dwms_train_cl(h, c3: nn.Conv2d, c5: nn.Conv2d, c7: nn.Conv2d):
    """h + dw3(h) + dw5(h) + dw7(h) (vmamba.py:622) as ONE 7x7 stencil: the fold is written in differentiable
    torch ops on the small weight tensors, so autograd hands each parameter its share of the stencil gradient."""
    w7 = c7.weight.float()
    c = w7.shape[0]
    wf = w7 + F.pad(c5.weight.float(), (1, 1, 1, 1)) + F.pad(c3.weight.float(), (2, 2, 2, 2))
    wf = wf + _centre_tap7(wf.device, wf.dtype)
    bt = c7.bias.float() + c5.bias.float() + c3.bias.float()
    return _DwConvCL.apply(h.contiguous(), wf.reshape(c, 49).t(), bt)


class _MaskPool:
    """Stochastic-depth masks of a whole training step from ONE bernoulli draw: the k-th use of a DropPath instance within
    a step owns a row of a (slots, batch) table (a VSSBlock uses its DropPath twice per forward, vmamba.py:384-396), and
    the table is redrawn at the first use after begin_step() -- called by the models' forward in training mode and by
    train.train_step.  ~50 x (bernoulli_ + div_) launches per step become three.  Code that never calls begin_step()
    still gets independent masks: uses simply keep taking new rows, and the 16th use of an instance starts a new table."""

    def __init__(self):
        self.slots, self.keep, self.count, self.buf, self.probs, self.fresh = {}, [], {}, None, None, False
        self.buf32 = None

    def begin_step(self):
        self.count = {}
        self.fresh = False

    def forget_draw(self):
        """the next take() draws a fresh table whatever the bookkeeping says (around a hipGraph capture)"""
        self.fresh = False

    def take(self, mod, keep, batch, dtype, device, want_f32=False):
        """the mask row of this use: (batch) in `dtype`, or the fp32 row the fused residual kernels read (want_f32)"""
        import weakref
        k = self.count.get(id(mod), 0)
        if k >= 16:
            self.begin_step()
            k = 0
        self.count[id(mod)] = k + 1
        ent = self.slots.get((id(mod), k))
        if ent is None or ent[1]() is not mod or self.keep[ent[0]] != keep:
            if len(self.keep) > 8192:                     # ids of modules long gone: start over
                self.slots, self.keep = {}, []
            ent = self.slots[(id(mod), k)] = (len(self.keep), weakref.ref(mod))
            self.keep.append(keep)
            self.fresh = False
        if (not self.fresh or self.buf is None or self.buf.shape != (len(self.keep), batch) or self.buf.dtype != dtype
                or self.buf.device != device):
            n = len(self.keep)
            if self.probs is None or self.probs.shape != (n, 1) or self.probs.device != device:
                self.probs = torch.tensor(self.keep, dtype=torch.float32).view(n, 1).to(device)
            self.buf32 = torch.bernoulli(self.probs.expand(n, batch)) / self.probs
            self.buf = self.buf32.to(dtype)
            self.fresh = True
        return self.buf32[ent[0]] if want_f32 else self.buf[ent[0]]


_mask_pool = _MaskPool()        # DropPath instances used outside a model that owns a pool (bare blocks in tests / scripts)


def model_mask_pool(model):
    """The stochastic-depth table of ONE model: created at its first training forward and handed to every DropPath below it,
    so that two training models in a process neither share draws nor reallocate the table a captured hipGraph of the other
    one reads (GraphedTrainStep keeps the tensors of its capture alive on top of that)."""
    pool = model.__dict__.get("_tramba_mask_pool")
    if pool is None:
        pool = model.__dict__["_tramba_mask_pool"] = _MaskPool()
        for m in model.modules():
            if isinstance(m, DropPath):
                m.__dict__["_pool"] = pool
    return pool


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.models.layers.DropPath semantics)."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def mask_for(self, x):
        """per-sample keep mask scaled by 1 / keep, broadcastable to x"""
        keep = 1.0 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        if x.is_cuda:
            pool = self.__dict__.get("_pool") or _mask_pool
            return pool.take(self, keep, x.shape[0], x.dtype, x.device).view(shape)
        return x.new_empty(shape).bernoulli_(keep).div_(keep)

    def mask_f32(self, x):
        """(B) fp32 keep / keep_prob per sample for the fused residual kernels, or None when this use drops nothing"""
        if self.drop_prob == 0.0 or not self.training:
            return None
        pool = self.__dict__.get("_pool") or _mask_pool
        return pool.take(self, 1.0 - self.drop_prob, x.shape[0], x.dtype, x.device, want_f32=True)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        return x * self.mask_for(x)

    def extra_repr(self):
        return f"drop_prob={self.drop_prob}"


# ----------------------------------------------------------------------------- basic layers
# Low-precision shadows of the fp32 master weights for the training step: the 16-bit W (N,K) the forward GEMM reads and the
# 16-bit W^T (K,N) the input-gradient GEMM reads (dx = gy @ W runs on the forward kernel with the transposed weight).
# ~230 per-layer `weight.to(bf16)` and as many `w.t().contiguous()` launches per step become ONE launch after the
# optimizer step (train.train_step -> refresh_lowp_shadows -> tramba_shadow_cast_multi, driven by a device-resident table
# of the model's weights).  A shadow is used only while the parameter's version counter still equals the one it was cast
# at; otherwise the layer casts / transposes as before.
_lowp_shadow = {}        # id(p) -> [shadow, version, weakref(p), shadow_t or None]


def _shadow_params(model):
    """[(parameter, dt_rank or None)]: the Linear2d weights, and the x_proj weights of the SS2D cores that can take the fused
    training path (their shadow is the padded (K*RG, D) layout the kernels read, see _XProjCL)"""
    out = [(m.weight, None) for m in model.modules() if isinstance(m, Linear2d)
           if m.weight is not None and m.weight.is_cuda and m.weight.dtype == torch.float32 and m.weight.dim() == 2]
    for m in model.modules():
        if isinstance(m, SS2D):
            p = m.x_proj_weight
            if p.is_cuda and p.dtype == torch.float32 and p.dim() == 3 and p.shape[1] == m.dt_rank + 2:
                out.append((p, m.dt_rank))
            q = m.dt_projs_weight          # (K, D, R): the scan backward's rank projection reads its (K, R, D) transpose
            if q.is_cuda and q.dtype == torch.float32 and q.dim() == 3:
                out.append((q, "dtw"))
    return out


@torch.no_grad()
def refresh_lowp_shadows(model, dtype):
    if dtype is None or dtype == torch.float32:
        return 0
    import weakref
    plan = model.__dict__.get("_tramba_shadow_plan")     # (signature, table, ntensors, total_tiles, dtype, params)
    params = plan[5] if plan is not None else _shadow_params(model)
    if not params:
        return 0
    sig = (dtype,) + tuple((p.data_ptr(), tuple(p.shape)) for p, _ in params)
    if plan is None or plan[0] != sig:
        params = _shadow_params(model)
        sig = (dtype,) + tuple((p.data_ptr(), tuple(p.shape)) for p, _ in params)
        for key in [k for k, ent in _lowp_shadow.items() if ent[2]() is None]:   # parameters of models that no longer exist
            del _lowp_shadow[key]
        rows, first = [], 0
        es = torch.empty((), dtype=dtype).element_size()

        def entry(src, dst, dst_t, n, k, ld, ld_t):
            nonlocal first
            rows.append([src, dst, dst_t, n, k, first, ld, ld_t])
            first += ((n + 63) // 64) * ((k + 63) // 64)

        for p, r in params:
            ent = _lowp_shadow.get(id(p))
            if r == "dtw":     # dt_projs_weight (K, D, R): only the per-direction transposes (K, R, D)
                kk, d, rr = p.shape
                if (ent is None or ent[3] is None or ent[3].dtype != dtype or ent[3].shape != (kk, rr, d)
                        or ent[3].device != p.device or ent[2]() is not p):
                    tt = torch.empty((kk, rr, d), dtype=dtype, device=p.device)
                    ent = [tt, -1, weakref.ref(p), tt]
                    _lowp_shadow[id(p)] = ent
                for g in range(kk):
                    entry(p.data_ptr() + g * d * rr * 4, 0, ent[3].data_ptr() + g * rr * d * es, d, rr, rr, d)
            elif r is None:      # Linear2d weight (N, K): W and W^T
                n, k = p.shape
                if (ent is None or ent[0].dtype != dtype or ent[0].shape != p.shape or ent[0].device != p.device
                        or ent[2]() is not p or ent[3] is None):
                    ent = [torch.empty_like(p, dtype=dtype), -1, weakref.ref(p),
                           torch.empty((k, n), dtype=dtype, device=p.device)]
                    _lowp_shadow[id(p)] = ent
                entry(p.data_ptr(), ent[0].data_ptr(), ent[3].data_ptr(), n, k, k, n)
            else:              # x_proj weight (K, R + 2, D) -> padded (K*RG, D) [R ranks, 0-pad, B, C, 2 pad] and its transpose
                kk, _, d = p.shape
                rg = hip.ss2d_group_stride(r)
                r8 = rg - 4
                if (ent is None or ent[0].dtype != dtype or ent[0].shape != (kk * rg, d) or ent[0].device != p.device
                        or ent[2]() is not p or ent[3] is None):
                    ent = [torch.zeros((kk * rg, d), dtype=dtype, device=p.device), -1, weakref.ref(p),
                           torch.zeros((d, kk * rg), dtype=dtype, device=p.device)]      # the pad rows stay zero for good
                    _lowp_shadow[id(p)] = ent
                for g in range(kk):
                    src = p.data_ptr() + g * (r + 2) * d * 4
                    entry(src, ent[0].data_ptr() + g * rg * d * es, ent[3].data_ptr() + g * rg * es, r, d, d, kk * rg)
                    entry(src + r * d * 4, ent[0].data_ptr() + (g * rg + r8) * d * es, ent[3].data_ptr() + (g * rg + r8) * es,
                          2, d, d, kk * rg)
        table = torch.tensor(rows, dtype=torch.int64).to(params[0][0].device)
        plan = (sig, table, len(rows), first, dtype, params)
        model.__dict__["_tramba_shadow_plan"] = plan
    hip.shadow_cast_multi(plan[1], plan[2], plan[3], dtype)
    for p, _ in params:
        _lowp_shadow[id(p)][1] = p._version
    return len(params)


def restamp_lowp_shadows(model):
    """Declare the shadows of `model`'s parameters current (after a hipGraph replay that refreshed them itself and a
    version bump of the parameters: tramba_amd.graph.GraphedTrainStep)."""
    for m in model.modules():
        ps = (m.weight,) if isinstance(m, Linear2d) else ((m.x_proj_weight, m.dt_projs_weight) if isinstance(m, SS2D) else ())
        for p in ps:
            ent = _lowp_shadow.get(id(p))
            if ent is not None and ent[2]() is p:
                ent[1] = p._version
    restamp_dw_packs(model)


def _lowp(p, dtype):
    """p cast to dtype: the shadow when it is current, a fresh cast otherwise"""
    if p.dtype == dtype:
        return p
    ent = _lowp_shadow.get(id(p))
    if ent is not None and ent[1] == p._version and ent[0].dtype == dtype and ent[2]() is p:
        return ent[0]
    return p.to(dtype)


def _lowp_t(p, dtype):
    """p^T cast to dtype when a current shadow holds it, else None (the caller transposes)"""
    ent = _lowp_shadow.get(id(p))
    if ent is not None and ent[1] == p._version and ent[0].dtype == dtype and ent[2]() is p:
        return ent[3]
    return None


class _LinearTrainCL(torch.autograd.Function):
    """y = x @ w^T + b on channels-last activations with autograd (training path), all on the library's matrix-core
    kernels: forward and input gradient tramba_linear_cl (bias in the epilogue), weight / bias gradient tramba_wgrad_cl
    (fp32 out, no cast kernel afterwards)."""

    @staticmethod
    def forward(ctx, x, w, b):
        wa = _lowp(w, x.dtype).detach().contiguous()
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        ctx.save_for_backward(x2, wa)
        # the transposed shadow is rewritten only by the refresh that follows this step's optimizer update
        ctx.wa_t = _lowp_t(w, x.dtype)
        ctx.wdtype, ctx.has_bias, ctx.xshape = w.dtype, b is not None, x.shape
        y = hip.linear_cl(x2, wa, None if b is None else b.detach().float().contiguous())
        return y.view(x.shape[:-1] + (wa.shape[0],))

    @staticmethod
    def backward(ctx, gy):
        x2, wa = ctx.saved_tensors
        gx = gw = gb = None
        gy2 = gy.reshape(-1, gy.shape[-1])
        gy2 = gy2 if gy2.is_contiguous() and gy2.dtype == x2.dtype else gy2.to(x2.dtype).contiguous()
        if ctx.needs_input_grad[0]:
            gx = _dgrad(gy2, wa, ctx.wa_t).view(ctx.xshape)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = _wgrad(gy2, x2, ctx.has_bias, defer=ctx.wdtype == torch.float32)   # (a cast below would READ the sum)
            gw = gw.to(ctx.wdtype)
        return gx, gw, gb


class _GeluLinearTrainCL(torch.autograd.Function):
    """y = gelu(h) @ w^T + b (training path): the activation belongs to the Linear that consumes it, so that its gradient
    rides in the epilogue of that Linear's input-gradient GEMM -- gh = (gy @ W) * gelu'(h) is ONE launch
    (TRAMBA_ACT_GELU_GRAD_MUL) instead of a GEMM and an elementwise kernel over the widest maps of the model (the 4x-wide
    Mlp hidden layer, the SS2D inner width)."""

    @staticmethod
    def forward(ctx, h, w, b, a=None):
        """a: gelu(h) when the producer has already written it (the dual-output GEMM of _LinearGeluPairTrainCL)"""
        wa = _lowp(w, h.dtype).detach().contiguous()
        h2 = h.reshape(-1, h.shape[-1])
        h2 = h2 if h2.is_contiguous() else h2.contiguous()
        a2 = F.gelu(h2) if a is None else a.detach().reshape(h2.shape)
        ctx.save_for_backward(h2, a2, wa)
        ctx.wa_t = _lowp_t(w, h.dtype)
        ctx.wdtype, ctx.has_bias, ctx.hshape = w.dtype, b is not None, h.shape
        y = hip.linear_cl(a2, wa, None if b is None else b.detach().float().contiguous())
        return y.view(h.shape[:-1] + (wa.shape[0],))

    @staticmethod
    def backward(ctx, gy):
        h2, a2, wa = ctx.saved_tensors
        gh = gw = gb = None
        gy2 = gy.reshape(-1, gy.shape[-1])
        gy2 = gy2 if gy2.is_contiguous() and gy2.dtype == h2.dtype else gy2.to(h2.dtype).contiguous()
        if ctx.needs_input_grad[0]:
            wt = wa.t().contiguous() if ctx.wa_t is None else ctx.wa_t
            gh = hip.linear_cl(gy2, wt, None, h2, hip.ACT_GELU_GRAD_MUL).view(ctx.hshape)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = _wgrad(gy2, a2, ctx.has_bias, defer=ctx.wdtype == torch.float32)
            gw = gw.to(ctx.wdtype)
        return gh, gw, gb, None


class _LinearGeluPairTrainCL(torch.autograd.Function):
    """(h, gelu(h)) with h = x @ w^T + b from ONE launch (tramba_linear_dual_cl): the Mlp's fc1 on the training path.  h
    carries the gradient; gelu(h) is handed to the next Linear (_GeluLinearTrainCL), which owns the activation's backward."""

    @staticmethod
    def forward(ctx, x, w, b):
        wa = _lowp(w, x.dtype).detach().contiguous()
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        ctx.save_for_backward(x2, wa)
        ctx.wa_t = _lowp_t(w, x.dtype)
        ctx.wdtype, ctx.has_bias, ctx.xshape = w.dtype, b is not None, x.shape
        h, a = hip.linear_dual_cl(x2, wa, None if b is None else b.detach().float().contiguous(), hip.ACT_GELU)
        shape = x.shape[:-1] + (wa.shape[0],)
        h, a = h.view(shape), a.view(shape)
        ctx.mark_non_differentiable(a)          # on the tensor that is RETURNED ...
        ctx.set_materialize_grads(False)        # ... and no (M, 4C) zero gradient is materialised for it in backward
        return h, a

    @staticmethod
    def backward(ctx, gh, _ga):
        if gh is None:
            return None, None, None
        x2, wa = ctx.saved_tensors
        gx = gw = gb = None
        gy2 = gh.reshape(-1, gh.shape[-1])
        gy2 = gy2 if gy2.is_contiguous() and gy2.dtype == x2.dtype else gy2.to(x2.dtype).contiguous()
        if ctx.needs_input_grad[0]:
            gx = _dgrad(gy2, wa, ctx.wa_t).view(ctx.xshape)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = _wgrad(gy2, x2, ctx.has_bias, defer=ctx.wdtype == torch.float32)   # (a cast below would READ the sum)
            gw = gw.to(ctx.wdtype)
        return gx, gw, gb


class _Linear2TrainCL(torch.autograd.Function):
    """y = [x1 | x2] @ w^T + b under autograd without the concatenation (decoder concat_back_dim, Trambav6.py:122-125;
    FreqSS2Dv6, freq_mamba.py:52-56): forward tramba_linear2_cl (the K loop reads the two tensors in turn); the two input
    gradients are GEMMs against the two row blocks of W^T, the weight gradient two TN GEMMs side by side."""

    @staticmethod
    def forward(ctx, x1, x2, w, b):
        wa = _lowp(w, x1.dtype).detach().contiguous()
        k1, k2 = x1.shape[-1], x2.shape[-1]
        a1, a2 = x1.reshape(-1, k1).contiguous(), x2.reshape(-1, k2).contiguous()
        ctx.save_for_backward(a1, a2, wa)
        ctx.wa_t = _lowp_t(w, x1.dtype)
        ctx.wdtype, ctx.has_bias, ctx.shapes = w.dtype, b is not None, (x1.shape, x2.shape)
        y = hip.linear2_cl(a1, a2, wa, None if b is None else b.detach().float().contiguous())
        return y.view(x1.shape[:-1] + (wa.shape[0],))

    @staticmethod
    def backward(ctx, gy):
        a1, a2, wa = ctx.saved_tensors
        k1 = a1.shape[1]
        gy2 = gy.reshape(-1, gy.shape[-1])
        gy2 = gy2 if gy2.is_contiguous() and gy2.dtype == a1.dtype else gy2.to(a1.dtype).contiguous()
        wt = wa.t().contiguous() if ctx.wa_t is None else ctx.wa_t             # (K1 + K2, N)
        g1 = hip.linear_cl(gy2, wt[:k1]).view(ctx.shapes[0]) if ctx.needs_input_grad[0] else None
        g2 = hip.linear_cl(gy2, wt[k1:]).view(ctx.shapes[1]) if ctx.needs_input_grad[1] else None
        gw = gb = None
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            gw1, gb = _wgrad(gy2, a1, ctx.has_bias)
            gw = torch.cat((gw1, _wgrad(gy2, a2)[0]), dim=1).to(ctx.wdtype)
        return g1, g2, gw, gb


class _ShuffleNormHeadCL(torch.autograd.Function):
    """logits (B, H*P, W*P) f32 = head(LayerNorm_C(pixel_shuffle_P(x))) without its bias, x (B, H, W, P*P*C): the last decoder
    stage of the training path (FinalPatchExpand_X4's norm + seg_layers[-1], Trambav6.py:132-137).  Forward: the inference
    kernel (tramba_shuffle_norm_head_cl); backward: one pass over x (tramba_shuffle_norm_head_bwd_cl) -- the normalised
    (B, 4H, 4W, C) map, 302 MB at batch 8, exists neither way.  Parameter gradients from the two row sums A, G."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, head_w, p, eps):
        x = x.contiguous()
        y = hip.shuffle_norm_head_cl(x, hip._f32(ln_w), hip._f32(ln_b), hip._f32(head_w).reshape(-1), 0.0, p, eps)
        ctx.save_for_backward(x, ln_w, ln_b, head_w)
        ctx.p, ctx.eps = p, eps
        return y

    @staticmethod
    def backward(ctx, g):
        x, ln_w, ln_b, head_w = ctx.saved_tensors
        gam, bet, hw = ln_w.detach().float(), ln_b.detach().float(), head_w.detach().float().reshape(-1)
        dx, part = hip.shuffle_norm_head_bwd_cl(x, g.contiguous().float(), gam, hw, ctx.p, ctx.eps)
        s_ = hip.slab_sum(part)
        c = gam.numel()
        a, gs = s_[:c], s_[c]
        return (dx if ctx.needs_input_grad[0] else None, (hw * a).to(ln_w.dtype), (hw * gs).to(ln_b.dtype),
                (gam * a + bet * gs).reshape(head_w.shape).to(head_w.dtype), None, None)


class _RowDotCL(torch.autograd.Function):
    """A C -> 1 segmentation head (nn.Conv2d(C, 1, 1), Trambav6.py:82,130) on a channels-last map with autograd:
    forward tramba_rowdot_cl (fp32 logits), weight gradient on tramba_wgrad_cl (the one output channel padded to the
    kernel's 8-channel granule)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        wf = w.detach().float().reshape(-1).contiguous()
        ctx.save_for_backward(x, wf)
        ctx.wshape, ctx.wdtype = w.shape, w.dtype
        return hip.rowdot_cl(x, wf, 0.0) + b.detach().float()

    @staticmethod
    def backward(ctx, gy):
        x, wf = ctx.saved_tensors
        gx = gw = gb = None
        if hip.rowdot_bwd_ok(x):   # one pass over x: input, weight and bias gradient (tramba_rowdot_bwd_cl)
            gx, gw, gb = hip.rowdot_bwd_cl(x, gy.float().contiguous(), wf)
            return gx, gw.reshape(ctx.wshape).to(ctx.wdtype), gb.reshape(1)
        if ctx.needs_input_grad[0]:
            gx = (gy.unsqueeze(-1) * wf).to(x.dtype)
        if ctx.needs_input_grad[1]:
            g8 = torch.zeros((gy.numel(), 8), dtype=x.dtype, device=x.device)
            g8[:, 0] = gy.reshape(-1)
            gw = _wgrad(g8, x.view(-1, x.shape[-1]))[0][0].reshape(ctx.wshape).to(ctx.wdtype)
        if ctx.needs_input_grad[2]:
            gb = gy.sum().reshape(1)
        return gx, gw, gb


class Linear2d(nn.Linear):
    """1x1 convolution stored as an (out,in) matrix (modules.py:10-19)."""

    def _forward_cl(self, x, act=hip.ACT_NONE, residual=None, out_dtype=None, gelu_in=False, gelu_ready=None):
        """gelu_in: x is a pre-activation whose GELU this layer applies first (training path: the producer leaves the
        activation to its consumer, see _GeluLinearTrainCL); gelu_ready: that GELU, when the producer wrote it as well"""
        if _infer(x, self.weight):
            w = self.weight if self.weight.dtype == x.dtype else self.weight.to(x.dtype)
            return hip.linear_cl(F.gelu(x) if gelu_in else x, w.detach(), _f32(self.bias), residual, act, out_dtype)
        if gelu_in:
            y = _act_torch(_GeluLinearTrainCL.apply(x, self.weight, self.bias, gelu_ready), act)
        else:
            y = _act_torch(_LinearTrainCL.apply(x, self.weight, self.bias), act)
        if residual is not None:
            y = y + residual
        return y if out_dtype is None else y.to(out_dtype)

    def _forward_gelu_pair_cl(self, x):
        """training path: (h, gelu(h) or None) with h = self(x) -- both from one launch where the GEMM kernel takes the shape"""
        if (not _infer(x, self.weight) and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.shape[-1] % 64 == 0
                and self.weight.shape[0] % 8 == 0):
            return _LinearGeluPairTrainCL.apply(x, self.weight, self.bias)
        return self._forward_cl(x), None

    def _forward_norm_cl(self, x, norm, act=hip.ACT_NONE, residual=None, out_dtype=None):
        """self(norm(x)) for a LayerNorm2d `norm` over the input channels.  16-bit inference folds the LayerNorm into the
        GEMM (tramba_linear_ln_cl: W * gamma cached per weight version, mean / rstd derived by the GEMM block from its own
        rows), so the normalised map and the LayerNorm launch do not exist; elsewhere the two ops run one after the other."""
        k = x.shape[-1]
        if (FOLD_LAYERNORM and _infer(x, self.weight, norm.weight) and x.dtype != torch.float32 and k % 64 == 0 and k <= 2048
                and self.weight.shape[0] % 8 == 0):
            wf, cs, tb = _cache(self).get(("lnfold", x.dtype, id(norm)), (self.weight, self.bias, norm.weight, norm.bias),
                                          lambda: _fold_layernorm(self, norm, x.dtype))
            return hip.linear_ln_cl(x, wf, cs, tb, norm.eps, residual, act, out_dtype)
        return self._forward_cl(norm._forward_cl(x), act=act, residual=residual, out_dtype=out_dtype)

    def _forward_cat_cl(self, x1, x2, act=hip.ACT_NONE, residual=None, out_dtype=None):
        """Linear2d(torch.cat((x1, x2), dim=-1)) -- in 16-bit inference the K loop reads the two tensors in turn
        (tramba_linear2_cl), so the concatenation never exists.  act = ACT_SIGMOID_GATE multiplies by `residual`."""
        if (_infer(x1, x2, self.weight) and x1.dtype != torch.float32
                and x1.dtype == x2.dtype
                and x1.shape[-1] % 64 == 0 and x2.shape[-1] % 64 == 0):
            w = self.weight if self.weight.dtype == x1.dtype else self.weight.to(x1.dtype)
            return hip.linear2_cl(x1.contiguous(), x2.contiguous(), w.detach(), _f32(self.bias), residual, act, out_dtype)
        gate = act == hip.ACT_SIGMOID_GATE
        if (x1.is_cuda and x1.dtype in (torch.bfloat16, torch.float16) and x2.dtype == x1.dtype and out_dtype is None
                and x1.shape[-1] % 64 == 0 and x2.shape[-1] % 64 == 0 and self.weight.shape[0] % 8 == 0):
            y = _act_torch(_Linear2TrainCL.apply(x1, x2, self.weight, self.bias), hip.ACT_NONE if gate else act)
            if gate:
                return torch.sigmoid(y) * residual
            return y if residual is None else y + residual
        y = self._forward_cl(torch.cat((x1, x2), dim=-1), act=hip.ACT_NONE if gate else act,
                             residual=None if gate else residual, out_dtype=out_dtype)
        return torch.sigmoid(y) * residual if gate else y

    def forward(self, x):
        _need_device(x)
        return from_cl(self._forward_cl(to_cl(x)))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        key = prefix + "weight"
        if key in state_dict:  # accept (out,in,1,1) conv checkpoints like the reference
            state_dict[key] = state_dict[key].view(self.weight.shape)
        return super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                             unexpected_keys, error_msgs)


def _fold_layernorm(lin, norm, dtype):
    """(W * gamma in `dtype`, its row sums as rounded, W beta + b) for the LayerNorm-folded GEMM; elementwise ops and
    reductions only (no BLAS call even at cache-build time)."""
    w = lin.weight.detach().float()
    wf = (w * norm.weight.detach().float()[None, :]).to(dtype).contiguous()
    colsum = wf.float().sum(dim=1).contiguous()
    t = (w * norm.bias.detach().float()[None, :]).sum(dim=1)
    if lin.bias is not None:
        t = t + lin.bias.detach().float()
    return wf, colsum, t.contiguous()


class _LayerNormCL(torch.autograd.Function):
    """LayerNorm over the last dim with autograd, both directions HIP (training path): no fp32 round trip of the
    activation, dgamma / dbeta by in-kernel accumulation + atomics."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = x.contiguous()
        wf = w.detach().float().contiguous()
        ctx.save_for_backward(x, wf)
        ctx.eps = eps
        ctx.defer = w.dtype == torch.float32 and b.dtype == torch.float32   # (an fp32 leaf takes the f32 sums as they are)
        return hip.layernorm_cl(x, wf, b.detach().float().contiguous(), eps, hip.ACT_NONE)

    @staticmethod
    def backward(ctx, gy):
        x, wf = ctx.saved_tensors
        dx, dw, db = hip.layernorm_bwd_cl(x, gy.contiguous().to(x.dtype), wf, ctx.eps, defer=ctx.defer)
        return dx, dw, db, None


class LayerNorm2d(nn.LayerNorm):
    """LayerNorm over C of an NCHW tensor (modules.py:22-27) -- no permutes needed in channels-last."""

    def _forward_cl(self, x, act=hip.ACT_NONE):
        if _infer(x, self.weight):
            return hip.layernorm_cl(x, _f32(self.weight), _f32(self.bias), self.eps, act)
        _need_device(x)
        return _act_torch(_LayerNormCL.apply(x, self.weight, self.bias, self.eps), act)

    def forward(self, x):
        _need_device(x)
        return from_cl(self._forward_cl(to_cl(x)))


class Permute(nn.Module):
    def __init__(self, *args):
        super().__init__()
        self.args = args

    def forward(self, x):
        return x.permute(*self.args).contiguous()


class Mlp(nn.Module):
    """modules.py:134-153."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0,
                 channels_first=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        assert channels_first, "the Tramba path is channel-first everywhere (Trambav6.py:21,34)"
        self.fc1 = Linear2d(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = Linear2d(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def _forward_cl(self, x, residual=None, pre_norm=None):
        """pre_norm: the LayerNorm2d the caller would apply to x first (folded into fc1 where possible)"""
        # training: the GELU is applied by fc2 (its gradient rides in fc2's input-gradient GEMM)
        defer = (not _infer(x, self.fc1.weight, self.fc2.weight)) and isinstance(self.act, nn.GELU) and self.drop.p == 0.0
        if defer:      # fc1 writes h and gelu(h) in one launch where it can; fc2 multiplies gelu(h) and owns the GELU's gradient
            xin = x if pre_norm is None else pre_norm._forward_cl(x)
            h, a = self.fc1._forward_gelu_pair_cl(xin)
            return self.fc2._forward_cl(h, residual=residual, gelu_in=True, gelu_ready=a)
        if pre_norm is not None:
            h = self.fc1._forward_norm_cl(x, pre_norm, act=hip.ACT_GELU)
        else:
            h = self.fc1._forward_cl(x, act=hip.ACT_GELU)
        h = self.drop(h)
        return self.drop(self.fc2._forward_cl(h, residual=residual))

    def forward(self, x):
        _need_device(x)
        return from_cl(self._forward_cl(to_cl(x)))


class _ShuffleNormCL(torch.autograd.Function):
    """rearrange 'b (p1 p2 c) h w -> b c (h p1) (w p2)' + LayerNorm2d(c) under autograd (PatchExpand / FreqExpand2D on the
    training path): forward tramba_shuffle_norm_cl (the shuffle is the store address of the LayerNorm), backward
    tramba_shuffle_norm_bwd_cl (the gradient is read through the shuffle) -- no permuted copy of either map."""

    @staticmethod
    def forward(ctx, xe, w, b, p, eps):
        xe = xe.contiguous()
        wf = w.detach().float().contiguous()
        ctx.save_for_backward(xe, wf)
        ctx.p, ctx.eps = p, eps
        ctx.defer = w.dtype == torch.float32 and b.dtype == torch.float32
        return hip.shuffle_norm_cl(xe, wf, b.detach().float().contiguous(), p, eps)

    @staticmethod
    def backward(ctx, gy):
        xe, wf = ctx.saved_tensors
        gy = gy.contiguous()
        if gy.dtype != xe.dtype:
            gy = gy.to(xe.dtype)
        dx, dw, db = hip.shuffle_norm_bwd_cl(xe, gy, wf, ctx.p, ctx.eps, defer=ctx.defer)
        return dx, dw, db, None, None


class _ExpandShuffleNorm(nn.Module):
    """1x1 expand -> 'b (p1 p2 c) h w -> b c (h p1) (w p2)' -> LayerNorm2d(c)."""

    scale = 2

    def _shuffle_norm(self, xe):
        p = self.scale
        if _infer(xe, self.norm.weight):
            return hip.shuffle_norm_cl(xe, _f32(self.norm.weight), _f32(self.norm.bias), p, self.norm.eps)
        if xe.is_cuda:
            return _ShuffleNormCL.apply(xe, self.norm.weight, self.norm.bias, p, self.norm.eps)
        b, h, w, cc = xe.shape
        c = cc // (p * p)
        y = xe.view(b, h, w, p, p, c).permute(0, 1, 3, 2, 4, 5).reshape(b, h * p, w * p, c)
        return self.norm._forward_cl(y)

    def _forward_cl(self, x):
        assert x.shape[1] == x.shape[2], "square maps only (modules.py:212)"
        return self._shuffle_norm(self.expand._forward_cl(x))

    def forward(self, x):
        _need_device(x)
        return from_cl(self._forward_cl(to_cl(x)))


class PatchExpand(_ExpandShuffleNorm):
    """modules.py:183-221."""

    def __init__(self, dim, dim_scale=2, norm_layer=LayerNorm2d, channel_first=True):
        super().__init__()
        assert channel_first and dim_scale == 2
        self.dim = dim
        self.channel_first = channel_first
        self.expand = Linear2d(dim, 2 * dim, bias=False)
        self.norm = norm_layer(dim // dim_scale)
        self.scale = 2


class FinalPatchExpand_X4(_ExpandShuffleNorm):
    """modules.py:224-274."""

    def __init__(self, dim, dim_scale=4, norm_layer=LayerNorm2d, channel_first=True):
        super().__init__()
        assert channel_first
        self.dim = dim
        self.channel_first = channel_first
        self.dim_scale = dim_scale
        self.expand = Linear2d(dim, (dim_scale ** 2) * dim, bias=False)
        self.output_dim = dim
        self.norm = norm_layer(self.output_dim)
        self.scale = dim_scale


class FreqExpand2D(_ExpandShuffleNorm):
    """modules.py:678-696."""

    def __init__(self, dim, norm_layer=LayerNorm2d, channel_first=True):
        super().__init__()
        assert channel_first
        self.dim = dim
        self.channel_first = channel_first
        self.expand = Linear2d(dim, 4 * dim, bias=False)
        self.norm = norm_layer(dim)
        self.scale = 2


# ----------------------------------------------------------------------------- DCT split
def _dct_filter(n: int) -> torch.Tensor:
    """DCT_2D.py:37-45: orthonormal DCT-II basis, built in Python floats, stored fp32."""
    m = torch.zeros(n, n)
    for v in range(n):
        s = math.sqrt(2.0) if v != 0 else 1.0
        for j in range(n):
            m[v, j] = math.cos(math.pi * (0.5 + j) * v / n) / math.sqrt(n) * s
    return m


class DCT2DSpatialTransformLayer_x(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.register_buffer("weight", _dct_filter(width))


class DCT2DSpatialTransformLayer_y(nn.Module):
    def __init__(self, height):
        super().__init__()
        self.register_buffer("weight", _dct_filter(height))


_COEF_SPLIT = {}
DCT_BWD_LOWP_INTERMEDIATE = True   # (scripts/ compare both forms; see _DCTSplitCL)


def _coef_split(w, lo):
    """(bf16 hi, bf16 lo) of one half of a DCT coefficient table -- rows [0, n/2) (lo) or [n/2, n) -- as the TN kernel's
    coefficient-major "a" operand, columns padded to the kernel's 16-byte rows.  Cached per table (address, version): the
    tables are constants of the architecture, and splitting them per call was 36 cast / subtract launches per training step."""
    key = (w.data_ptr(), w._version, w.shape[0], w.shape[1], lo)
    ent = _COEF_SPLIT.get(key)
    if ent is None:
        if len(_COEF_SPLIT) >= 64:       # (tables re-created over and over, e.g. models built in a loop: start again)
            _COEF_SPLIT.clear()
        h2 = w.shape[0] // 2
        q = (w if lo is None else w[:h2] if lo else w[h2:]).float().contiguous()     # lo = None: the whole table
        if q.shape[1] % 8:
            q = F.pad(q, (0, 8 - q.shape[1] % 8))
        hi, lo_ = _split16(q)
        ent = _COEF_SPLIT[key] = (hi, lo_, torch.stack((hi, lo_)).contiguous())     # [2]: both pieces for ONE two-batch launch
    return ent


def _tn_coef(split, n, x):
    """a^T @ x[g] for a split coefficient table a = hi + lo (T, N) and activations x (G, T, K): two passes of the 16-bit TN
    kernel's contraction on bf16 data (the coefficients keep fp32 accuracy, the data is what it is); other dtypes are split the same way
    (a third pass)."""
    ah, al, both = split
    if x.dtype == torch.bfloat16:
        out = hip.tn_shared_cl(both, x)       # hi and lo as two "batches" over the same x: one launch, no add
    else:
        xh, xl = _split16(x.float())
        out = hip.tn_shared_cl(ah, xh) + hip.tn_shared_cl(al, xh) + hip.tn_shared_cl(ah, xl)
    return out[:, :n] if out.shape[1] != n else out


class _DCTSplitCL(torch.autograd.Function):
    """DCT split with autograd (training path).  Forward = the inference kernel (tramba_dct_split_cl: LL and HH quadrants
    of Wy X Wx^T).  Backward: gX = Wy_lo^T gLow Wx_lo + Wy_hi^T gHigh Wx_hi -- each term two TN contractions on the
    library's grouped matrix-core kernel (over the coefficient index of W, then of H), coefficients in fp32 accuracy.  With
    bf16 activations the intermediate between the two contractions is a bf16 map like every other activation of the step
    (one cast; the fp32 validation mode keeps it in hi + lo pieces)."""

    @staticmethod
    def forward(ctx, x, wx, wy):
        ctx.save_for_backward(wx, wy)
        ctx.xdtype = x.dtype
        return hip.dct_split_cl(x.contiguous(), wx, wy)

    @staticmethod
    def backward(ctx, g_high, g_low):
        wx, wy = ctx.saved_tensors
        n = wx.shape[0]
        h2 = n // 2
        if ctx.xdtype == torch.bfloat16 and DCT_BWD_LOWP_INTERMEDIATE and g_high is not None and g_low is not None:
            # both quadrants' first contractions land in the two halves of ONE (B, v, W*C) map: the second contraction then
            # runs over all n rows of Wy at once (no sum of two full-size maps)
            bsz, _, _, c = g_low.shape
            z1 = torch.empty((bsz, n, n * c), dtype=torch.bfloat16, device=g_low.device)
            for g, lo in ((g_low, True), (g_high, False)):
                t = _tn_coef(_coef_split(wx, lo), n, g.contiguous().view(bsz * h2, h2, c))     # (B*V, W, C): contracted over u
                (z1[:, :h2] if lo else z1[:, h2:]).copy_(t.reshape(bsz, h2, n * c))
            gx = _tn_coef(_coef_split(wy, None), n, z1)                                        # (B, H, W*C): contracted over v
            return gx.view(bsz, n, n, c).to(ctx.xdtype), None, None
        gx = None
        for g, lo in ((g_low, True), (g_high, False)):
            if g is None:
                continue
            bsz, _, _, c = g.shape
            z1 = _tn_coef(_coef_split(wx, lo), n, g.contiguous().view(bsz * h2, h2, c))    # (B*V, W, C): contracted over u
            if ctx.xdtype == torch.bfloat16 and DCT_BWD_LOWP_INTERMEDIATE:
                z1 = z1.to(torch.bfloat16)
            z2 = _tn_coef(_coef_split(wy, lo), n, z1.reshape(bsz, h2, n * c))              # (B, H, W*C): contracted over v
            gx = z2 if gx is None else gx + z2
        return gx.view(-1, n, n, g_high.shape[-1] if g_high is not None else g_low.shape[-1]).to(ctx.xdtype), None, None


class DCT2D(nn.Module):
    """DCT_2D.py:6-29: Y = Wy X Wx^T; returns (high = HH quadrant, low = LL quadrant)."""

    def __init__(self, height, width):
        super().__init__()
        self.dct_x = DCT2DSpatialTransformLayer_x(width)
        self.dct_y = DCT2DSpatialTransformLayer_y(height)

    def _forward_cl(self, x):
        wx, wy = self.dct_x.weight, self.dct_y.weight
        if x.shape[1] != wy.shape[0] or x.shape[2] != wx.shape[0]:
            raise RuntimeError(f"DCT2D built for {wy.shape[0]}x{wx.shape[0]}, got {x.shape[1]}x{x.shape[2]}")
        if _infer(x):
            return hip.dct_split_cl(x, _f32(wx), _f32(wy))
        return _DCTSplitCL.apply(x, _f32(wx), _f32(wy))

    def forward(self, x):
        _need_device(x)
        high, low = self._forward_cl(to_cl(x))
        return from_cl(high), from_cl(low)


# ----------------------------------------------------------------------------- mamba parameter init
def Dt_init(dt_rank, d_inner, dt_scale=1.0, dt_init="random", dt_min=0.001, dt_max=0.1, dt_init_floor=1e-4):
    """mamba_init.py:7-31."""
    proj = nn.Linear(dt_rank, d_inner, bias=True)
    std = dt_rank ** -0.5 * dt_scale
    if dt_init == "constant":
        nn.init.constant_(proj.weight, std)
    elif dt_init == "random":
        nn.init.uniform_(proj.weight, -std, std)
    else:
        raise NotImplementedError(dt_init)
    dt = torch.exp(torch.rand(d_inner) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)).clamp(min=dt_init_floor)
    with torch.no_grad():
        proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))  # inverse softplus
    return proj


def A_log_init(d_state, d_inner, copies=-1, device=None, merge=True):
    """mamba_init.py:34-48 (S4D-real)."""
    a = torch.arange(1, d_state + 1, dtype=torch.float32, device=device).repeat(d_inner, 1)
    a_log = torch.log(a)
    if copies > 0:
        a_log = a_log.unsqueeze(0).repeat(copies, 1, 1)
        if merge:
            a_log = a_log.flatten(0, 1)
    p = nn.Parameter(a_log.contiguous())
    p._no_weight_decay = True
    return p


def D_init(d_inner, copies=-1, device=None, merge=True):
    """mamba_init.py:51-60."""
    d = torch.ones(d_inner, device=device)
    if copies > 0:
        d = d.unsqueeze(0).repeat(copies, 1)
        if merge:
            d = d.flatten(0, 1)
    p = nn.Parameter(d.contiguous())
    p._no_weight_decay = True
    return p


# ----------------------------------------------------------------------------- SS2D
_const_zeros = {}


def _zeros_const(shape, dtype, device):
    """A shared all-zero constant (never written): the zero bias of bias-free stencils, padding blocks -- a fresh
    torch.zeros per call is one fill launch each, hundreds per training step."""
    key = (tuple(shape), dtype, str(device))
    t = _const_zeros.get(key)
    if t is None:
        t = _const_zeros[key] = torch.zeros(shape, dtype=dtype, device=device)
    return t


class _XProjCL(torch.autograd.Function):
    """x_proj once in spatial order (vmamba.py:233-234 evaluated for all K directions at every pixel): x (.., D)
    activations dtype, w the RAW parameter (K, R + 2, D) -> (.., K*RG) fp32 rows in the scan kernels' padded group layout
    [R ranks, 0-pad to R8, B, C, 2 pad].  The padding lives inside the Function (one cat forward, one cat backward) instead
    of a dozen autograd slice ops per call; the output comes from the GEMM's fp32 accumulators (the rows feed softplus /
    exp).  Backward: tramba_linear_cl / tramba_wgrad_cl."""

    @staticmethod
    def forward(ctx, x, w):
        k, r2, d = w.shape
        r = r2 - 2
        rg = hip.ss2d_group_stride(r)
        ent = _lowp_shadow.get(id(w))
        if (ent is not None and ent[1] == w._version and ent[0].dtype == x.dtype and ent[2]() is w
                and ent[0].shape == (k * rg, d)):
            wa, ctx.wa_t = ent[0], ent[3]       # padded weight and its transpose, written after the optimizer step
        else:
            wl = w.detach().to(x.dtype)
            parts = [wl[:, :r]]
            if rg - 4 - r:
                parts.append(_zeros_const((k, rg - 4 - r, d), x.dtype, x.device))
            parts += [wl[:, r:], _zeros_const((k, 2, d), x.dtype, x.device)]
            wa, ctx.wa_t = torch.cat(parts, dim=1).view(k * rg, d), None
        ctx.save_for_backward(x, wa)
        ctx.meta = (w.dtype, k, r, rg)
        return hip.linear_cl(x.contiguous(), wa, out_dtype=torch.float32)

    @staticmethod
    def backward(ctx, g):
        x, wa = ctx.saved_tensors
        wdtype, k, r, rg = ctx.meta
        ga = g.to(x.dtype).reshape(-1, g.shape[-1]).contiguous()
        gx = _dgrad(ga, wa, ctx.wa_t).view(x.shape) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            gp = _wgrad(ga, x.reshape(-1, x.shape[-1]))[0].view(k, rg, -1)
            gw = torch.cat((gp[:, :r], gp[:, rg - 4:rg - 2]), dim=1).to(wdtype)
        return gx, gw


class _SS2DCoreCL(torch.autograd.Function):
    """scan gather + dt_proj + selective scan + CrossMerge of forward_corev2 (vmamba.py:230-262) on channels-last
    tensors with autograd, both directions HIP (training path): forward = the inference kernels, backward =
    tramba_ss2d_scan_bwd_cl (adjoint recurrence, gradients in sequence order) + the small projections as batched
    GEMMs.  x (B,L,D); xdbl (B,L,K*RG) f32; dt_w (K,D,R), dt_bias (K*D), a_neg (K*D) = -exp(A_logs), ds (K*D) f32.
    Returns the merged map (B,L,D) in x's dtype (before out_norm)."""

    @staticmethod
    def forward(ctx, x, xdbl, dt_w, dt_bias, a_neg, ds, order):
        x = x.contiguous()
        xdbl = xdbl.contiguous()
        dt_w, dt_bias, a_neg, ds = (t.detach().float().contiguous() for t in (dt_w, dt_bias, a_neg, ds))
        # per-direction outputs and the merged map travel in the activation dtype, as in inference (fp32 in the fp32
        # validation mode): half the bytes of the K-fold intermediate, and out_norm / GELU / their backward run on 2-byte maps
        # the forward launch also writes the state entering every tile: the backward kernel would recompute them in a sweep of
        # its own (a quarter of its time)
        states = hip.ss2d_scan_states(x, order)
        ys = hip.ss2d_scan_cl(x, xdbl, order, dt_w, dt_bias, a_neg, ds, x.dtype, states=states)
        ctx.save_for_backward(x, xdbl, dt_w, dt_bias, a_neg, ds, states)
        ctx.order = order
        return hip.ss2d_merge_sum_cl(ys, order, x.dtype)

    @staticmethod
    def backward(ctx, gym):
        x, xdbl, dt_w, dt_bias, a_neg, ds, states = ctx.saved_tensors
        order = ctx.order
        b, l, d = x.shape
        k, r = order.k, dt_w.shape[-1]
        rg = hip.ss2d_group_stride(r)
        r8 = rg - 4
        gym = gym.contiguous()
        if gym.dtype not in (torch.float32, x.dtype):
            gym = gym.float()
        # (dB / dC stay packed (B,K,L) arrays: accumulating them straight into the RG-strided columns of g_seq -- the
        # kernel can, `bc_stride` -- put every atomic on a cache line of its own and cost the scan backward 30 %)
        gu, graw, g_b, g_c, gpar = hip.ss2d_scan_bwd_cl(x, xdbl, order, dt_w, dt_bias, a_neg, ds, gym, states=states)
        g_seq = torch.zeros((b, k, l, rg), dtype=torch.float32, device=x.device)   # x_dbl-row gradients, sequence order
        gx = hip.ss2d_merge_sum_cl(gu, order, x.dtype)
        # row (l, k) of xdbl viewed as (B, L*K, RG) that sequence position (k, i) reads: table[k][i] * K + k.  One gather
        # and one index_add_ over all K directions instead of 4 launches per direction (660 small kernels per step)
        flat = getattr(order, "_flat_rows", None)
        if flat is None or flat.device != x.device:
            flat = (order.table.long() * k + torch.arange(k, device=x.device)[:, None]).reshape(-1)
            order._flat_rows = flat
        xdf = xdbl.view(b, l * k, rg)
        ranks = xdf[:, flat, :r].view(b, k, l, r)                                          # (B,K,L,R) f32
        # the two per-direction projections on the library's grouped kernels, straight on the (B,K,L,.) tensors:
        #   d(dt_projs_weight)[k] = sum_{b,l} graw^T ranks   (TN, tramba_wgrad_cl with groups = K, batches = B)
        #   d(ranks)              = graw @ dt_w[k]           (tramba_rows_gemm_cl into the first R floats of every row)
        if r % 8:                                   # the TN kernel moves 16-byte rows: pad the rank columns
            ranks = F.pad(ranks, (0, 8 - r % 8))
        if graw.dtype != torch.float32:
            cd = graw.dtype
            g_dtw = hip.wgrad_grouped_cl(graw, ranks.to(cd))                                # (K,D,R)
            dtw_t = torch.empty((k, r, d), dtype=cd, device=x.device).copy_(dt_w.transpose(1, 2))   # transpose + cast: 1 launch
            hip.rows_gemm_cl(graw.view(b * k, l, d), dtw_t, g_seq.view(b * k, l, rg), r)
        else:   # fp32 validation mode: three passes on bf16 splits
            gh, gl = _split16(graw)
            rh, rl = _split16(ranks)
            wh, wl = _split16(dt_w.transpose(1, 2).contiguous())                            # (K, R, D)
            g_dtw = hip.wgrad_grouped_cl(gh, rh) + hip.wgrad_grouped_cl(gh, rl) + hip.wgrad_grouped_cl(gl, rh)
            tmp = torch.empty_like(g_seq)
            hip.rows_gemm_cl(gh.view(b * k, l, d), wh, g_seq.view(b * k, l, rg), r)
            for ga_, wa_ in ((gh, wl), (gl, wh)):
                hip.rows_gemm_cl(ga_.view(b * k, l, d), wa_, tmp.view(b * k, l, rg), r)
                g_seq[..., :r] += tmp[..., :r]
        if g_dtw.shape[-1] != r:
            g_dtw = g_dtw[..., :r].contiguous()     # (a view would be cloned by autograd's gradient accumulation anyway)
        g_seq[..., r8] = g_b
        g_seq[..., r8 + 1] = g_c
        # back to spatial order.  Where every direction visits every pixel exactly once (raster, window, dilation orders) the
        # rows are permuted, not accumulated: one gather through the inverse permutation (deterministic, no zero fill);
        # the Helix lines revisit pixels, there the rows are summed by index_add_
        inv = getattr(order, "_flat_inverse", None)
        if inv is None or (torch.is_tensor(inv) and inv.device != x.device):
            if int(torch.unique(flat).numel()) == flat.numel() == l * k:      # (one host read per scan order, cached)
                inv = torch.empty_like(flat)
                inv[flat] = torch.arange(flat.numel(), device=flat.device)
            else:
                inv = False
            order._flat_inverse = inv
        if inv is not False:
            g_xd = g_seq.view(b, k * l, rg)[:, inv]
        else:
            g_xd = torch.zeros_like(xdf)
            g_xd.index_add_(1, flat, g_seq.view(b, k * l, rg))
        gp = hip.slab_sum(gpar)                                                           # (3,K,D): contiguous planes
        return (gx, g_xd.view(b, l, k * rg), g_dtw, gp[2].reshape(-1), gp[0].reshape(-1), gp[1].reshape(-1), None)


# ----------------------------------------------------------------------------- fused training path (round 3)
# The residual stream of a block under autograd, without framework launches between the library's kernels: a residual add is
# DEFERRED to the LayerNorm that follows it (one pass: x' = x + y * mask, n = LN(x')), whose backward hands back both the
# gradient of x' (its own contribution + the skip connection's) and that gradient times the stochastic-depth mask.
class _Deferred:
    """A pending residual add: stands for x + y * mask[sample] (mask (B) f32 or None)."""
    __slots__ = ("x", "y", "mask")

    def __init__(self, x, y, mask):
        self.x, self.y, self.mask = x, y, mask


class _AddMaskedF32(torch.autograd.Function):
    """x + y * mask[sample] with an fp32 (B) mask (the end of a stage, where no LayerNorm follows to absorb the add)."""

    @staticmethod
    def forward(ctx, x, y, mask):
        m = mask.to(y.dtype).view((-1,) + (1,) * (y.dim() - 1))
        ctx.save_for_backward(m)
        return torch.addcmul(x, y, m)

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        return (g if ctx.needs_input_grad[0] else None), (g * m if ctx.needs_input_grad[1] else None), None


def _resolve(v):
    """the tensor a value of the residual stream stands for"""
    if not isinstance(v, _Deferred):
        return v
    return v.x + v.y if v.mask is None else _AddMaskedF32.apply(v.x, v.y, v.mask)


class _AddLayerNormCL(torch.autograd.Function):
    """(x', n, act(n)) = (x + y * mask, LayerNorm(x'), its activation) from ONE launch (tramba_add_layernorm_cl).
    y None: no add -- x' is x itself, handed through when `passthrough` (so that the skip connection's gradient and the
    LayerNorm's meet inside this Function's backward, one launch, instead of in an autograd add), else None.
    act ACT_NONE: no activation output.  Backward: tramba_layernorm_bwd_res_cl."""

    @staticmethod
    def forward(ctx, x, y, mask, w, b, eps, act, passthrough):
        x = x.contiguous()
        wf = w.detach().float().contiguous()
        xs, n, na = hip.add_layernorm_cl(x, None if y is None else y.contiguous(), mask, wf, b.detach().float().contiguous(),
                                         eps, act, dual=act != hip.ACT_NONE)
        ctx.save_for_backward(x if xs is None else xs, wf, mask)
        ctx.eps, ctx.has_y = eps, y is not None
        ctx.defer = w.dtype == torch.float32 and b.dtype == torch.float32
        if na is not None:
            ctx.mark_non_differentiable(na)
        ctx.set_materialize_grads(False)
        if xs is None:
            xs = x if passthrough else None
        return xs, n, na

    @staticmethod
    def backward(ctx, g_xs, g_n, _g_na):
        xs, wf, mask = ctx.saved_tensors
        if g_n is None:      # the normalised map was not used: only the skip connection carries a gradient
            gy = None
            if ctx.has_y and g_xs is not None:
                gy = g_xs if mask is None else g_xs * mask.to(g_xs.dtype).view((-1,) + (1,) * (g_xs.dim() - 1))
            return g_xs, gy, None, None, None, None, None, None
        g_n = g_n.contiguous()
        if g_n.dtype != xs.dtype:
            g_n = g_n.to(xs.dtype)
        if g_xs is not None:
            g_xs = g_xs.contiguous()
            if g_xs.dtype != xs.dtype:
                g_xs = g_xs.to(xs.dtype)
        want_m = ctx.has_y and mask is not None and ctx.needs_input_grad[1]
        dx, dxm, dw, db = hip.layernorm_bwd_res_cl(xs, g_n, wf, ctx.eps, gres=g_xs, mask=mask if want_m else None,
                                                   want_masked=want_m, defer=ctx.defer)
        gy = (dxm if want_m else dx) if ctx.has_y else None
        return dx, gy, None, dw, db, None, None, None


def _add_ln(v, norm, act=hip.ACT_NONE):
    """(x', n, act(n) or None) for a value of the residual stream (tensor or _Deferred) and a LayerNorm2d"""
    if isinstance(v, _Deferred):
        return _AddLayerNormCL.apply(v.x, v.y, v.mask, norm.weight, norm.bias, norm.eps, act, True)
    return _AddLayerNormCL.apply(v, None, None, norm.weight, norm.bias, norm.eps, act, True)


# Packed depth-wise stencils of the TRAINING path.  The weights change once per step, at the optimizer: instead of one small
# pack launch per stencil inside every forward (39 per step on Tramba-V, ~7 us each: a thread per channel walking 49 strided
# taps), `refresh_dw_packs(model)` repacks all of them in ONE launch right after the optimizer step (train.train_step), into
# buffers that live as long as the model; a forward finds its pack here when the sources' version counters still match.
_dw_pack_store = {}       # data_ptr of the main weight -> [wt, bt, versions of the six sources, weakref to the main weight]
_dw_plans = None          # model -> (signature, prepared launch, sources); weak keys, and not in the model's __dict__ (the
                          # prepared launch holds ctypes pointer arrays, which copy.deepcopy(model) could not copy)


def _dw_versions(srcs):
    return tuple(-1 if t is None else t._version for t in srcs)


def _dw_packed(w, b, w3=None, b3=None, w5=None, b5=None):
    """(wt, bt) of hip.dw_pack for these parameters: the model's standing pack when it is current, a fresh one otherwise"""
    ent = _dw_pack_store.get(w.data_ptr())
    if ent is not None and ent[2] == _dw_versions((w, b, w3, b3, w5, b5)):
        p = ent[3]()
        if p is not None and p.data_ptr() == w.data_ptr() and p.shape == w.shape:
            return ent[0], ent[1]
    return hip.dw_pack(*[None if t is None else t.detach() for t in (w, b, w3, b3, w5, b5)])


def _dw_pack_sources(model):
    """the parameter groups (w, b, w3, b3, w5, b5) of every depth-wise stencil the fused training path packs"""
    out, folded = [], set()
    for m in model.modules():
        if isinstance(m, DWMSMlp):
            c3, c5, c7 = m.dwc3.dw_conv, m.dwc5.dw_conv, m.dwc7.dw_conv
            if all(c.bias is not None for c in (c3, c5, c7)):
                out.append((c7.weight, c7.bias, c3.weight, c3.bias, c5.weight, c5.bias))
                folded.update(id(c) for c in (c3, c5, c7))
    for m in model.modules():
        if isinstance(m, SS2D) and getattr(m, "with_dconv", False):
            out.append((m.conv2d.weight, m.conv2d.bias, None, None, None, None))
    return [g for g in out if all(t is None or (t.is_cuda and t.dtype == torch.float32) for t in g)]


@torch.no_grad()
def refresh_dw_packs(model):
    """Repack every depth-wise stencil of `model` (one launch) and declare the packs current.  Call after the optimizer step."""
    import weakref
    global _dw_plans
    if _dw_plans is None:
        _dw_plans = weakref.WeakKeyDictionary()
    plan = _dw_plans.get(model)
    srcs = plan[2] if plan is not None else _dw_pack_sources(model)
    if not srcs:
        return 0
    sig = tuple(g[0].data_ptr() for g in srcs)
    if plan is None or plan[0] != sig:
        srcs = _dw_pack_sources(model)
        sig = tuple(g[0].data_ptr() for g in srcs)
        for key in [k for k, ent in _dw_pack_store.items() if ent[3]() is None]:      # models that no longer exist
            del _dw_pack_store[key]
        items = []
        for g in srcs:
            w = g[0]
            c, ks = w.shape[0], w.shape[-1]
            ent = _dw_pack_store.get(w.data_ptr())
            if ent is None or ent[3]() is not w or tuple(ent[0].shape) != (ks * ks, c) or ent[0].device != w.device:
                ent = [torch.empty((ks * ks, c), dtype=torch.float32, device=w.device),
                       torch.empty((c,), dtype=torch.float32, device=w.device), None, weakref.ref(w)]
                _dw_pack_store[w.data_ptr()] = ent
            items.append(tuple(None if t is None else t.detach() for t in g) + (ent[0], ent[1]))
        plan = (sig, hip.dw_pack_multi_prepare(items), srcs)
        _dw_plans[model] = plan
    hip.dw_pack_multi_run(plan[1])
    # the kernel rewrote the standing packs through raw pointers: bump their version counters as an in-place torch op would, so
    # that a backward whose forward saved a pack BEFORE this refresh raises instead of using the new weights' stencil
    packs = []
    for g in srcs:
        ent = _dw_pack_store.get(g[0].data_ptr())
        if ent is not None:
            packs += [ent[0], ent[1]]
    if packs:
        torch.autograd.graph.increment_version(packs)
    restamp_dw_packs(model)
    return len(srcs)


def restamp_dw_packs(model):
    """Declare the standing packs of `model` current (after a hipGraph replay that repacked them itself and a version bump of
    the parameters: tramba_amd.graph.GraphedTrainStep)."""
    plan = None if _dw_plans is None else _dw_plans.get(model)
    if plan is None:
        return
    for g in plan[2]:
        ent = _dw_pack_store.get(g[0].data_ptr())
        if ent is not None and ent[3]() is g[0]:
            ent[2] = _dw_versions(g)


class _DwConvActCL(torch.autograd.Function):
    """(z, act(z)) with z = depth-wise conv(x) + bias from one launch (tramba_dwconv_dual_cl): SS2D's conv2d + SiLU
    (vmamba.py:283-285) on the training path.  z carries the gradient; act(z) is handed to the consumer, which owns the
    activation's backward.  Weights arrive in the reference layout (C,1,ks,ks) / (C) and are packed tap-major here
    (tramba_dw_pack); backward: the same stencil with mirrored taps, tramba_dwconv_wgrad_cl, tramba_dw_unpack_grad."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x = x.contiguous()
        wt, bt = _dw_packed(w, b)
        z, a = hip.dwconv_dual_cl(x, wt, bt, act)
        ctx.save_for_backward(x, wt)
        ctx.ks, ctx.has_bias, ctx.wdtype = w.shape[-1], b is not None, w.dtype
        ctx.mark_non_differentiable(a)
        ctx.set_materialize_grads(False)
        return z, a

    @staticmethod
    def backward(ctx, gz, _ga):
        if gz is None:
            return None, None, None, None
        x, wt = ctx.saved_tensors
        gz = gz.contiguous()
        if gz.dtype != x.dtype:
            gz = gz.to(x.dtype)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = hip.dwconv_dual_cl(gz, wt, _zeros_const((wt.shape[1],), torch.float32, wt.device), hip.ACT_NONE,
                                    want_pre=False, flip=True)[1]
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            defer = ctx.wdtype == torch.float32      # (leaf gradients, handed on as they are: nothing reads them before the optimizer)
            g7, _, _, gbs = hip.dw_unpack_grad(hip.dwconv_wgrad_table(x, gz, ctx.ks, defer=defer), ctx.ks, False,
                                               1 if ctx.has_bias else 0, defer=defer)
            gw = g7.to(ctx.wdtype)
            gb = gbs[0] if ctx.has_bias else None
        return gx, gw, gb, None


class _DwmsActCL(torch.autograd.Function):
    """(g, gelu(g)) with g = h + dw3(h) + dw5(h) + dw7(h) (vmamba.py:622-625) as ONE folded 7x7 stencil with a dual store;
    the fold (tramba_dw_pack) and its adjoint (tramba_dw_unpack_grad) are one small launch each."""

    @staticmethod
    def forward(ctx, h, w3, b3, w5, b5, w7, b7):
        h = h.contiguous()
        wt, bt = _dw_packed(w7, b7, w3, b3, w5, b5)
        g, a = hip.dwconv_dual_cl(h, wt, bt, hip.ACT_GELU)
        ctx.save_for_backward(h, wt)
        ctx.wdtype = w7.dtype
        ctx.mark_non_differentiable(a)
        ctx.set_materialize_grads(False)
        return g, a

    @staticmethod
    def backward(ctx, gg, _ga):
        if gg is None:
            return (None,) * 7
        h, wt = ctx.saved_tensors
        gg = gg.contiguous()
        if gg.dtype != h.dtype:
            gg = gg.to(h.dtype)
        gh = None
        if ctx.needs_input_grad[0]:
            gh = hip.dwconv_dual_cl(gg, wt, _zeros_const((wt.shape[1],), torch.float32, wt.device), hip.ACT_NONE,
                                    want_pre=False, flip=True)[1]
        defer = ctx.wdtype == torch.float32
        g7, g5, g3, gbs = hip.dw_unpack_grad(hip.dwconv_wgrad_table(h, gg, 7, defer=defer), 7, True, 3, defer=defer)
        t = ctx.wdtype
        return gh, g3.to(t), gbs[0], g5.to(t), gbs[1], g7.to(t), gbs[2]


def _xproj_padded(w, dtype):
    """the x_proj weight (K, R + 2, D) in the scan kernels' padded (K*RG, D) layout and its transpose (or None), `dtype`:
    the shadows written after the optimizer step when they are current, else built here"""
    k, r2, d = w.shape
    r = r2 - 2
    rg = hip.ss2d_group_stride(r)
    ent = _lowp_shadow.get(id(w))
    if (ent is not None and ent[1] == w._version and ent[0].dtype == dtype and ent[2]() is w
            and ent[0].shape == (k * rg, d)):
        return ent[0], ent[3]
    wl = w.detach().to(dtype)
    parts = [wl[:, :r]]
    if rg - 4 - r:
        parts.append(_zeros_const((k, rg - 4 - r, d), dtype, w.device))
    parts += [wl[:, r:], _zeros_const((k, 2, d), dtype, w.device)]
    return torch.cat(parts, dim=1).view(k * rg, d), None


class _SS2DInnerCL(torch.autograd.Function):
    """SiLU -> x_proj -> scan gather + dt_proj + selective scan -> CrossMerge of forward_corev2 (vmamba.py:230-262) on
    channels-last tensors under autograd, every launch the library's.  Inputs: z (B,L,D) the pre-activation of the SiLU in
    front of the core and xs = silu(z) (both from _DwConvActCL; this Function owns the SiLU's backward), the RAW parameters
    x_proj_weight (K,R+2,D), dt_projs_weight (K,D,R), dt_projs_bias (K,D), A_logs (K*D,1), Ds (K*D).  Returns the merged map
    (B,L,D) before out_norm.
    Forward: x_proj once in spatial order (GEMM, fp32 rows) -> tramba_ss2d_scan_cl (A = -exp(A_logs) formed in the kernel,
    tile-entry states saved) -> merge.  Backward: tramba_ss2d_scan_bwd_cl (dB / dC as per-channel-tile partials, no atomics)
    -> tramba_ss2d_bwd_prep_cl (rank rows gathered, partials summed) -> the two per-direction projections on the grouped
    matrix-core kernels -> tramba_ss2d_bwd_assemble_cl (x_dbl-row gradients back to spatial order through the inverse
    table, deterministic) -> x_proj input gradient (GEMM) -> tramba_ss2d_merge_grad_cl: (merge(gu) + that) * silu'(z)."""

    @staticmethod
    def forward(ctx, z, xs, xw, dt_w, dt_b, a_logs, ds, order):
        z, xs = z.contiguous(), xs.contiguous()
        wa, wa_t = _xproj_padded(xw, xs.dtype)
        xdbl = hip.linear_cl(xs, wa, out_dtype=torch.float32)
        dtw, dtb = dt_w.detach().float().contiguous(), dt_b.detach().float().reshape(-1).contiguous()
        al, dsf = a_logs.detach().float().reshape(-1).contiguous(), ds.detach().float().contiguous()
        states = hip.ss2d_scan_states(xs, order)
        ys = hip.ss2d_scan_cl(xs, xdbl, order, dtw, dtb, al, dsf, xs.dtype, states=states, a_log=True)
        ctx.save_for_backward(z, xs, xdbl, dtw, dtb, al, dsf, states, wa)
        ctx.wa_t, ctx.order = wa_t, order
        ent = _lowp_shadow.get(id(dt_w))
        ctx.dtw_t = ent[3] if (ent is not None and ent[1] == dt_w._version and ent[2]() is dt_w
                               and ent[3] is not None and ent[3].dtype == xs.dtype) else None
        ctx.meta = (xw.dtype, xw.shape, dt_w.dtype, dt_b.shape, a_logs.shape)
        ctx.par_f32 = dt_b.dtype == a_logs.dtype == ds.dtype == torch.float32    # (deferred sums go to fp32 leaves only)
        return hip.ss2d_merge_sum_cl(ys, order, xs.dtype)

    @staticmethod
    def backward(ctx, gym):
        z, xs, xdbl, dtw, dtb, al, dsf, states, wa = ctx.saved_tensors
        order = ctx.order
        xwdtype, xwshape, dtwdtype, dtbshape, alshape = ctx.meta
        b, l, d = xs.shape
        k, r = order.k, dtw.shape[-1]
        rg = hip.ss2d_group_stride(r)
        r8 = rg - 4
        gym = gym.contiguous()
        if gym.dtype not in (torch.float32, xs.dtype):
            gym = gym.float()
        gu, graw, bpart, cpart, gpar = hip.ss2d_scan_bwd_cl(xs, xdbl, order, dtw, dtb, al, dsf, gym, states=states,
                                                           a_log=True, bc_partials=True)
        cd = torch.bfloat16 if xs.dtype == torch.float32 else xs.dtype
        ranks, gseq = hip.ss2d_bwd_prep(xdbl, order, bpart, cpart, r, torch.float32 if xs.dtype == torch.float32 else cd)
        # the two per-direction projections on the (B,K,L,.) tensors as they are:
        #   d(dt_projs_weight)[k] = sum_{b,l} graw^T ranks   (TN, tramba_wgrad_cl with groups = K, batches = B)
        #   d(ranks)              = graw @ dt_w[k]           (tramba_rows_gemm_cl into the first R floats of every gseq row)
        if xs.dtype != torch.float32:
            # (K, D, R8); a leaf's gradient, handed on as it is when no columns are cut and no cast follows: its slab sums
            # join the step's batched ones
            g_dtw = hip.wgrad_grouped_cl(graw, ranks, defer=ranks.shape[-1] == r and dtwdtype == torch.float32)
            dtw_t = ctx.dtw_t            # (K, R, D): the shadow written after the optimizer step, when current
            if dtw_t is None:
                dtw_t = torch.empty((k, r, d), dtype=cd, device=xs.device).copy_(dtw.transpose(1, 2))   # transpose + cast
            hip.rows_gemm_cl(graw.view(b * k, l, d), dtw_t, gseq.view(b * k, l, rg), r)
        else:   # fp32 validation mode: three passes on bf16 splits
            gh, gl = _split16(graw)
            rh, rl = _split16(ranks)
            wh, wl = _split16(dtw.transpose(1, 2).contiguous())                                    # (K, R, D)
            g_dtw = hip.wgrad_grouped_cl(gh, rh) + hip.wgrad_grouped_cl(gh, rl) + hip.wgrad_grouped_cl(gl, rh)
            tmp = torch.empty_like(gseq)
            hip.rows_gemm_cl(gh.view(b * k, l, d), wh, gseq.view(b * k, l, rg), r)
            for ga_, wa_ in ((gh, wl), (gl, wh)):
                hip.rows_gemm_cl(ga_.view(b * k, l, d), wa_, tmp.view(b * k, l, rg), r)
                gseq[..., :r] += tmp[..., :r]
        if g_dtw.shape[-1] != r:
            g_dtw = g_dtw[..., :r].contiguous()
        g_xd = hip.ss2d_bwd_assemble(gseq, order, r, xs.dtype).view(b * l, k * rg)        # spatial order, activation dtype
        gz = None
        if ctx.needs_input_grad[0]:
            t = _dgrad(g_xd, wa, ctx.wa_t).view(b, l, d)                                   # the x_proj branch's share
            gz = hip.ss2d_merge_grad_cl(gu, order, t, z)                                   # (merge(gu) + t) * silu'(z)
        gxw = None
        if ctx.needs_input_grad[2]:
            if xs.dtype != torch.float32:
                # the GEMM's rows are in the scan kernels' padded layout (K x RG rows: R ranks, pad, B, C, pad); the parameter
                # keeps (K, R + 2, D): its row ranges are summed from the partial slabs straight into place
                segs = [s_ for g in range(k) for s_ in ((g * rg, r), (g * rg + r8, 2))]
                gxw = hip.wgrad_rows_cl(g_xd, xs.view(b * l, d), segs, defer=xwdtype == torch.float32).view(k, r + 2, d)
                gxw = gxw.to(xwdtype)
            else:
                gp_ = _wgrad(g_xd, xs.view(b * l, d))[0].view(k, rg, d)
                gxw = torch.cat((gp_[:, :r], gp_[:, r8:r8 + 2]), dim=1).to(xwdtype)
        gp = hip.slab_sum(gpar, defer=ctx.par_f32)                                             # (3,K,D): contiguous planes, leaf gradients
        return (gz, None, gxw, g_dtw.to(dtwdtype), gp[2].reshape(dtbshape), gp[0].reshape(alshape), gp[1].reshape(-1), None)



class SS2D(nn.Module):
    """vmamba.py:18-323, forward_type v2 / channel_first / disable_z (the only configuration the
    shipped models use).  ``scan`` / ``merge`` / ``k_group`` are the reference's plugin API."""

    def __init__(self, d_model=96, d_state=16, ssm_ratio=2.0, dt_rank="auto", act_layer=nn.SiLU,
                 d_conv=3, conv_bias=False, dropout=0.0, bias=False,
                 dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, initialize="v0",
                 forward_type="v2", channel_first=False, disable_z=True,
                 scan=CrossScan, merge=CrossMerge, k_group=4, **kwargs):
        super().__init__()
        if not channel_first or not disable_z:
            raise NotImplementedError("tramba_amd.SS2D implements the channel_first / disable_z configuration")
        if act_layer is not nn.SiLU:
            raise NotImplementedError("SS2D act_layer must be SiLU (fused into the depth-wise conv kernel)")
        d_inner = int(ssm_ratio * d_model)
        dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.channel_first = channel_first
        self.with_dconv = d_conv > 1
        self.disable_z = disable_z
        self.d_state, self.d_inner, self.dt_rank, self.k_group = d_state, d_inner, dt_rank, k_group
        self.scan, self.merge = scan, merge

        self.out_norm = LayerNorm2d(d_inner)
        self.in_proj = Linear2d(d_model, d_inner, bias=bias)
        self.act = act_layer()
        if self.with_dconv:
            self.conv2d = nn.Conv2d(d_inner, d_inner, groups=d_inner, bias=conv_bias, kernel_size=d_conv,
                                    padding=(d_conv - 1) // 2)
        x_proj = [nn.Linear(d_inner, dt_rank + d_state * 2, bias=False) for _ in range(k_group)]
        self.x_proj_weight = nn.Parameter(torch.stack([t.weight for t in x_proj], dim=0))  # (K, R+2N, D)
        self.out_act = nn.GELU()
        self.out_proj = Linear2d(d_inner, d_model, bias=bias)
        self.dropout = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()
        if initialize != "v0":
            raise NotImplementedError(initialize)
        dt_projs = [Dt_init(dt_rank, d_inner, dt_scale, dt_init, dt_min, dt_max, dt_init_floor) for _ in range(k_group)]
        self.dt_projs_weight = nn.Parameter(torch.stack([t.weight for t in dt_projs], dim=0))  # (K, D, R)
        self.dt_projs_bias = nn.Parameter(torch.stack([t.bias for t in dt_projs], dim=0))  # (K, D)
        self.A_logs = A_log_init(d_state, d_inner, copies=k_group, merge=True)  # (K*D, N)
        self.Ds = D_init(d_inner, copies=k_group, merge=True)  # (K*D)

    # -- the two core paths -----------------------------------------------------------------
    def _fused_ok(self, x):
        return (self.d_state == 1 and self.dt_rank <= 64
                and getattr(self.scan, "_tramba_family", None) is not None
                and getattr(self.merge, "_tramba_family", None) == self.scan._tramba_family
                and self.scan._tramba_k == self.k_group
                and _infer(x, self.x_proj_weight, self.dt_projs_weight, self.A_logs))

    def _padded_x_proj(self, dtype):
        """x_proj weight in the (K*RG, D) layout of the fused scan."""
        p = self.x_proj_weight
        return _cache(self).get(("xproj", dtype), (p,), lambda: hip.pad_x_proj_weight(p.detach().to(dtype)).contiguous())

    def _scan_params(self):
        """fp32 kernel-ready (dt_w, dt_bias, -exp(A_logs), Ds)."""
        ps = (self.dt_projs_weight, self.dt_projs_bias, self.A_logs, self.Ds)
        return _cache(self).get("scan", ps, lambda: (
            _f32(self.dt_projs_weight).clone(), _f32(self.dt_projs_bias).reshape(-1).clone(),
            (-torch.exp(self.A_logs.detach().float())).reshape(-1).contiguous(), _f32(self.Ds).clone()))

    def _core_fused_cl(self, x):
        """x (B,H,W,D) after conv+SiLU -> GELU(out_norm(merge(scan(...)))), all HIP."""
        b, h, w, d = x.shape
        k, r = self.k_group, self.dt_rank
        order = hip.scan_order(self.scan._tramba_family, h, w, x.device)
        xf = x.view(b, h * w, d)
        xdbl = hip.linear_cl(xf, self._padded_x_proj(x.dtype), out_dtype=torch.float32)  # x_proj once, spatial order
        dt_w, dt_b, a_neg, ds = self._scan_params()
        # the per-direction outputs travel between the scan and the merge kernel in the activation dtype (fp32 sums and
        # LayerNorm statistics inside the merge): half the bytes of the K-fold intermediate at 16-bit activations
        ys_dtype = x.dtype
        ys = hip.ss2d_scan_cl(xf, xdbl, order, dt_w, dt_b, a_neg, ds, ys_dtype)
        y = hip.ss2d_merge_norm_cl(ys, order, _f32(self.out_norm.weight), _f32(self.out_norm.bias),
                                   self.out_norm.eps, hip.ACT_GELU, x.dtype)
        return y.view(b, h, w, d)

    def _train_fused_ok(self, x):
        return (x.is_cuda and self.d_state == 1 and self.dt_rank <= 64
                and getattr(self.scan, "_tramba_family", None) is not None
                and getattr(self.merge, "_tramba_family", None) == self.scan._tramba_family
                and self.scan._tramba_k == self.k_group)

    def _core_train_cl(self, x):
        """Training path of forward_corev2 without leaving channels-last: x_proj once in spatial order (autograd
        GEMM), then _SS2DCoreCL and out_norm (HIP LayerNorm autograd); returns the PRE-activation of vmamba.py:270's GELU."""
        b, h, w, d = x.shape
        order = hip.scan_order(self.scan._tramba_family, h, w, x.device)
        xf = x.reshape(b, h * w, d)
        xdbl = _XProjCL.apply(xf, self.x_proj_weight)
        a_neg = -torch.exp(self.A_logs.float()).reshape(-1)
        ym = _SS2DCoreCL.apply(xf, xdbl, self.dt_projs_weight.float(), self.dt_projs_bias.float().reshape(-1), a_neg,
                               self.Ds.float(), order)
        # out_norm only: the GELU (vmamba.py:270) is applied by out_proj, whose input-gradient GEMM carries its gradient
        y = self.out_norm._forward_cl(ym.view(b, h, w, d))
        return y.to(x.dtype)

    def _core_plugin_cl(self, x):
        """Reference graph (vmamba.py:230-273) with the scan classes as plugins; autograd-capable."""
        b, h, w, d = x.shape
        k, r, n = self.k_group, self.dt_rank, self.d_state
        l = h * w
        xn = from_cl(x).contiguous()  # (B,D,H,W), NCHW contiguous as the plugin API expects
        xs = self.scan.apply(xn)  # (B,K,D,L)
        # the reference's two grouped conv1d (vmamba.py:233-236) as batched matmuls over the K groups
        x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, self.x_proj_weight.to(xs.dtype))
        dts, bs, cs = torch.split(x_dbl, [r, n, n], dim=2)
        dts = torch.einsum("bkrl,kdr->bkdl", dts, self.dt_projs_weight.to(xs.dtype))
        a_neg = -torch.exp(self.A_logs.float())
        ys = SelectiveScanOflex.apply(
            xs.view(b, -1, l), dts.contiguous().view(b, -1, l), a_neg, bs.contiguous().view(b, k, n, l),
            cs.contiguous().view(b, k, n, l), self.Ds.float(), self.dt_projs_bias.view(-1).float(), True, 1, 1, True)
        y = self.merge.apply(ys.view(b, k, -1, h, w))  # (B,D,L) fp32
        y = to_cl(y.view(b, -1, h, w))
        y = self.out_norm._forward_cl(y, act=hip.ACT_GELU)
        return y.to(x.dtype)

    def _train_path_ok(self, x):
        """the fused training path: built-in scan family, N = 1, depth-wise conv present, autograd on"""
        return (self.with_dconv and self._train_fused_ok(x) and isinstance(self.dropout, nn.Identity)
                and not _infer(x, self.in_proj.weight, self.x_proj_weight, self.dt_projs_weight, self.A_logs))

    def _forward_train_cl(self, xn):
        """forwardv2 (vmamba.py:275-291) under autograd on a channels-last map that has ALREADY been normalised by the
        caller; returns out_proj's output (the residual branch, before any add).  Seven launches: in_proj, conv + SiLU (dual
        store), x_proj, scan, merge, out_norm + GELU (dual store), out_proj."""
        b, h, w, _ = xn.shape
        d = self.d_inner
        xi = _LinearTrainCL.apply(xn, self.in_proj.weight, self.in_proj.bias)
        z, xs = _DwConvActCL.apply(xi, self.conv2d.weight, self.conv2d.bias, hip.ACT_SILU)
        order = hip.scan_order(self.scan._tramba_family, h, w, xn.device)
        ym = _SS2DInnerCL.apply(z.view(b, h * w, d), xs.view(b, h * w, d), self.x_proj_weight, self.dt_projs_weight,
                                self.dt_projs_bias, self.A_logs, self.Ds, order)
        # out_norm and the GELU of vmamba.py:270 in one pass; out_proj's input-gradient GEMM carries the GELU's gradient
        _, yn, ya = _AddLayerNormCL.apply(ym.view(b, h, w, d), None, None, self.out_norm.weight, self.out_norm.bias,
                                          self.out_norm.eps, hip.ACT_GELU, False)
        return _GeluLinearTrainCL.apply(yn, self.out_proj.weight, self.out_proj.bias, ya)

    def _forward_cl(self, x, residual=None, pre_norm=None):
        """pre_norm: the LayerNorm2d the caller would apply to x first (folded into in_proj where possible)"""
        if self._train_path_ok(x):
            y = self._forward_train_cl(x if pre_norm is None else pre_norm._forward_cl(x))
            return y if residual is None else y + residual
        x = self.in_proj._forward_norm_cl(x, pre_norm) if pre_norm is not None else self.in_proj._forward_cl(x)
        if self.with_dconv:
            if _infer(x, self.conv2d.weight):
                cv = self.conv2d
                wt, bt = _cache(self).get("dw", (cv.weight, cv.bias), lambda: hip.dw_pack(cv.weight, cv.bias))
                x = hip.dwconv_cl(x, wt, bt, hip.ACT_SILU)
            else:
                x = F.silu(_dwconv_train_cl(x, self.conv2d))
        else:
            x = F.silu(x)
        gelu_in = False
        if self._fused_ok(x):
            y = self._core_fused_cl(x)
        elif self._train_fused_ok(x):
            y = self._core_train_cl(x)
            gelu_in = True
        else:
            y = self._core_plugin_cl(x)
        return self.dropout(self.out_proj._forward_cl(y, residual=residual, gelu_in=gelu_in))

    def forward(self, x):
        _need_device(x)
        return from_cl(self._forward_cl(to_cl(x)))


class _AddMasked(torch.autograd.Function):
    """x + y * mask (stochastic depth on the residual branch, mask = keep / keep_prob per sample): one launch forward, one
    backward (gy = g * mask; gx is g itself) -- autograd's own graph for addcmul issues three small launches per block."""

    @staticmethod
    def forward(ctx, x, y, mask):
        ctx.save_for_backward(mask)
        return torch.addcmul(x, y, mask)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return (g if ctx.needs_input_grad[0] else None), (g * mask if ctx.needs_input_grad[1] else None), None


class _ResidualBlock(nn.Module):
    def _residual(self, x, branch, drop_path):
        """x + drop_path(branch(x_normed)); the add is fused into the branch's last GEMM when possible."""
        if isinstance(drop_path, DropPath) and drop_path.training and drop_path.drop_prob > 0.0:
            y = branch(None)
            return _AddMasked.apply(x, y, drop_path.mask_for(y))   # x + y * mask / keep in one launch
        return branch(x)

    def _train_stream_ok(self, x):
        """autograd on, device tensor, and an SS2D that takes the fused training path: the block runs on the deferred
        residual stream (_Deferred / _add_ln)"""
        t = x.x if isinstance(x, _Deferred) else x
        return self.op._train_path_ok(t)

    def _forward_stream(self, x, norm_a, norm_b, defer):
        """x1 = x + dp(op(norm_a(x))); x2 = x1 + dp(mlp(norm_b(x1))) (vmamba.py:384-396) with every residual add folded into
        the LayerNorm that follows it (and its gradient into that LayerNorm's backward); the last add is left pending for
        the next block's first LayerNorm when `defer`."""
        dp = self.drop_path
        x0, n1, _ = _add_ln(x, norm_a)
        y1 = self.op._forward_train_cl(n1)
        x1, n2, _ = _add_ln(_Deferred(x0, y1, dp.mask_f32(y1) if isinstance(dp, DropPath) else None), norm_b)
        y2 = self.mlp._forward_cl(n2)
        out = _Deferred(x1, y2, dp.mask_f32(y2) if isinstance(dp, DropPath) else None)
        return out if defer else _resolve(out)


def _run_blocks(blocks, x):
    """a stage's blocks in sequence on the residual stream: adds stay pending between blocks, the stage's output is a tensor"""
    for blk in blocks:
        x = blk._forward_cl(x, defer=True)
    return _resolve(x)


class VSSBlock(_ResidualBlock):
    """vmamba.py:327-396 (pre-norm): x += SS2D(LN(x)); x += Mlp(LN(x))."""

    def __init__(self, hidden_dim, drop_path=0.0, norm_layer=LayerNorm2d, channel_first=True,
                 ssm_d_state=1, ssm_ratio=2.0, ssm_dt_rank="auto", ssm_act_layer=nn.SiLU, ssm_conv=3,
                 ssm_conv_bias=False, ssm_drop_rate=0.0, ssm_init="v0", forward_type="v05",
                 mlp_ratio=4.0, mlp_act_layer=nn.GELU, mlp_drop_rate=0.0, post_norm=False, **kwargs):
        super().__init__()
        assert channel_first and not post_norm and ssm_ratio > 0 and mlp_ratio > 0
        self.norm = norm_layer(hidden_dim)
        self.op = SS2D(d_model=hidden_dim, d_state=ssm_d_state, ssm_ratio=ssm_ratio, dt_rank=ssm_dt_rank,
                       act_layer=ssm_act_layer, d_conv=ssm_conv, conv_bias=ssm_conv_bias, dropout=ssm_drop_rate,
                       initialize=ssm_init, forward_type=forward_type, channel_first=channel_first)
        self.drop_path = DropPath(drop_path)
        self.norm2 = norm_layer(hidden_dim)
        self.mlp = Mlp(in_features=hidden_dim, hidden_features=int(hidden_dim * mlp_ratio), act_layer=mlp_act_layer,
                       drop=mlp_drop_rate, channels_first=channel_first)

    def _forward_cl(self, x, defer=False):
        if self._train_stream_ok(x):
            return self._forward_stream(x, self.norm, self.norm2, defer)
        x = _resolve(x)
        x = self._residual(x, lambda res: self.op._forward_cl(x, residual=res, pre_norm=self.norm), self.drop_path)
        x = self._residual(x, lambda res: self.mlp._forward_cl(x, residual=res, pre_norm=self.norm2), self.drop_path)
        return x

    def forward(self, x):
        _need_device(x)
        return from_cl(self._forward_cl(to_cl(x)))


class DWConv(nn.Module):
    """vmamba.py:595-603."""

    def __init__(self, in_channels, kernel_size, bias):
        super().__init__()
        self.dw_conv = nn.Conv2d(in_channels, in_channels, kernel_size, stride=1, padding=(kernel_size - 1) // 2,
                                 bias=bias, groups=in_channels)

    def _forward_cl(self, x):
        c = self.dw_conv
        if _infer(x, c.weight):
            wt, bt = _cache(self).get("dw", (c.weight, c.bias), lambda: hip.dw_pack(c.weight, c.bias))
            return hip.dwconv_cl(x, wt, bt, hip.ACT_NONE)
        return _dwconv_train_cl(x, c)

    def forward(self, x):
        _need_device(x)
        return from_cl(self._forward_cl(to_cl(x)))


class DWMSMlp(nn.Module):
    """vmamba.py:606-629: fc1 -> h + dw3(h) + dw5(h) + dw7(h) -> GELU -> fc2."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0,
                 channels_first=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        assert channels_first
        self.fc1 = Linear2d(in_features, hidden_features)
        self.act = act_layer()
        self.dwc3 = DWConv(hidden_features, 3, True)
        self.dwc5 = DWConv(hidden_features, 5, True)
        self.dwc7 = DWConv(hidden_features, 7, True)
        self.fc2 = Linear2d(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def _forward_cl(self, x, residual=None, pre_norm=None):
        h = self.fc1._forward_norm_cl(x, pre_norm) if pre_norm is not None else self.fc1._forward_cl(x)
        c3, c5, c7 = self.dwc3.dw_conv, self.dwc5.dw_conv, self.dwc7.dw_conv
        if _infer(h, c3.weight, c5.weight, c7.weight):
            wt, bt = _cache(self).get(
                "dwms", (c3.weight, c3.bias, c5.weight, c5.bias, c7.weight, c7.bias),
                lambda: hip.dw_pack(c7.weight, c7.bias, c3.weight, c3.bias, c5.weight, c5.bias))
            g = hip.dwconv_cl(h, wt, bt, hip.ACT_GELU)
            defer = False
        elif self.drop.p == 0.0 and h.is_cuda and all(c.bias is not None for c in (c3, c5, c7)):
            # training: the folded stencil writes the pre-activation and its GELU in one launch; fc2 owns the GELU's
            # gradient (it rides in fc2's input-gradient GEMM)
            g, ga = _DwmsActCL.apply(h, c3.weight, c3.bias, c5.weight, c5.bias, c7.weight, c7.bias)
            return self.fc2._forward_cl(g, residual=residual, gelu_in=True, gelu_ready=ga)
        else:
            defer = self.drop.p == 0.0     # the GELU is applied by fc2 (gradient in fc2's input-gradient GEMM)
            g = _dwms_train_cl(h, c3, c5, c7)
            g = g if defer else F.gelu(g)
        g = self.drop(g)
        return self.drop(self.fc2._forward_cl(g, residual=residual, gelu_in=defer))

    def forward(self, x):
        _need_device(x)
        return from_cl(self._forward_cl(to_cl(x)))


class MultiScaleDecoderBlock(_ResidualBlock):
    """vmamba.py:632-704: Helix-SS2D (K=8) + multi-scale depth-wise MLP."""

    def __init__(self, hidden_dim, drop_path=0.0, norm_layer=LayerNorm2d, channel_first=True,
                 ssm_d_state=1, ssm_ratio=2.0, ssm_dt_rank="auto", ssm_act_layer=nn.SiLU, ssm_conv=3,
                 ssm_conv_bias=False, ssm_drop_rate=0.0, ssm_init="v0", forward_type="v05",
                 mlp_ratio=4.0, mlp_act_layer=nn.GELU, mlp_drop_rate=0.0, post_norm=False,
                 scan=CrossScan_Line, merge=CrossMerge_Line, k_group=8, **kwargs):
        super().__init__()
        assert channel_first and not post_norm and ssm_ratio > 0 and mlp_ratio > 0
        self.norm1 = norm_layer(hidden_dim)
        self.op = SS2D(d_model=hidden_dim, d_state=ssm_d_state, ssm_ratio=ssm_ratio, dt_rank=ssm_dt_rank,
                       act_layer=ssm_act_layer, d_conv=ssm_conv, conv_bias=ssm_conv_bias, dropout=ssm_drop_rate,
                       initialize=ssm_init, forward_type=forward_type, channel_first=channel_first, disable_z=True,
                       scan=scan, merge=merge, k_group=k_group)
        self.drop_path = DropPath(drop_path)
        self.norm2 = norm_layer(hidden_dim)
        self.mlp = DWMSMlp(in_features=hidden_dim, hidden_features=int(hidden_dim * mlp_ratio), act_layer=mlp_act_layer,
                           drop=mlp_drop_rate, channels_first=channel_first)

    def _forward_cl(self, x, defer=False):
        if self._train_stream_ok(x):
            return self._forward_stream(x, self.norm1, self.norm2, defer)
        x = _resolve(x)
        x = self._residual(x, lambda res: self.op._forward_cl(x, residual=res, pre_norm=self.norm1), self.drop_path)
        x = self._residual(x, lambda res: self.mlp._forward_cl(x, residual=res, pre_norm=self.norm2), self.drop_path)
        return x

    def forward(self, x):
        _need_device(x)
        return from_cl(self._forward_cl(to_cl(x)))


# ----------------------------------------------------------------------------- Dual-Frequency VSS
class FreqSS2Dv6(nn.Module):
    """freq_mamba.py:11-57: DCT split -> Hi-Fi (window scan) / Lo-Fi (dilated scan) SS2D -> sigmoid gate."""

    def __init__(self, dim, input_resolution=None, norm_layer=LayerNorm2d,
                 l_scan=CrossScan_Dilation, h_scan=CrossScan_Window,
                 l_merge=CrossMerge_Dilation, h_merge=CrossMerge_Window):
        super().__init__()
        self.dim = dim
        self.DCT2D = DCT2D(input_resolution[0], input_resolution[1])
        ss = dict(d_model=dim, d_state=1, ssm_ratio=2.0, dt_rank="auto", act_layer=nn.SiLU, d_conv=3, conv_bias=False,
                  dropout=0.0, initialize="v0", channel_first=True, bias=False, disable_z=True, k_group=4)
        self.l_expand = FreqExpand2D(dim)
        self.l_ssm = SS2D(scan=l_scan, merge=l_merge, **ss)
        self.h_expand = FreqExpand2D(dim)
        self.h_ssm = SS2D(scan=h_scan, merge=h_merge, **ss)
        self.sig = nn.Sigmoid()
        self.concat_back_dim = Linear2d(dim * 2, dim, bias=False)

    def _forward_cl(self, x):
        high, low = self.DCT2D._forward_cl(x)
        high = self.h_expand._forward_cl(high)
        low = self.l_expand._forward_cl(low)
        hifi = self.h_ssm._forward_cl(high)
        lofi = self.l_ssm._forward_cl(low)
        # x * sigmoid(W [hifi | lofi]): two-source GEMM with the gate in its epilogue (freq_mamba.py:52-56)
        return self.concat_back_dim._forward_cat_cl(hifi, lofi, act=hip.ACT_SIGMOID_GATE, residual=x)

    def forward(self, x):
        _need_device(x)
        return from_cl(self._forward_cl(to_cl(x)))


class FreqBlockv6(nn.Module):
    """freq_mamba.py:60-82."""

    def __init__(self, dim=128, input_resolution=None, num_heads=6, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0,
                 l_scan=CrossScan_Dilation, h_scan=CrossScan_Window, l_merge=CrossMerge_Dilation,
                 h_merge=CrossMerge_Window, act_layer=nn.GELU, norm_layer=LayerNorm2d, local_ws=2, alpha=0.5):
        super().__init__()
        self.input_resolution = input_resolution
        self.dim = dim
        self.mlp_ratio = mlp_ratio
        self.norm1 = LayerNorm2d(dim)
        self.attn = FreqSS2Dv6(dim, input_resolution=input_resolution, norm_layer=norm_layer, l_scan=l_scan,
                               h_scan=h_scan, l_merge=l_merge, h_merge=h_merge)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = LayerNorm2d(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=0.0,
                       channels_first=True)

    def _forward_cl(self, x):
        x = x + self.drop_path(self.attn._forward_cl(self.norm1._forward_cl(x)))
        if isinstance(self.drop_path, nn.Identity):
            return self.mlp._forward_cl(x, residual=x, pre_norm=self.norm2)
        return x + self.drop_path(self.mlp._forward_cl(x, pre_norm=self.norm2))

    def forward(self, x):
        _need_device(x)
        return from_cl(self._forward_cl(to_cl(x)))


# ----------------------------------------------------------------------------- encoder
def _init_weights(m: nn.Module):
    """vmamba.py:459-471 / Trambav6.py:86-97."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)
    elif isinstance(m, (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
        nn.init.kaiming_normal_(m.weight, a=1e-2)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)


def _conv_cl(conv: nn.Conv2d, x_cl):
    """A dense 3x3 / stride-2 conv on a channels-last activation: implicit GEMM on the MFMA tile kernel
    for 16-bit inference, MIOpen (NHWC) otherwise (fp32 validation mode, training)."""
    if (_infer(x_cl, conv.weight) and x_cl.dtype != torch.float32 and conv.kernel_size == (3, 3)
            and conv.stride == (2, 2) and conv.padding == (1, 1) and conv.in_channels % 64 == 0):
        wk = _cache(conv).get(("kmajor", x_cl.dtype), (conv.weight,),
                              lambda: conv.weight.detach().permute(0, 2, 3, 1).reshape(conv.out_channels, -1)
                              .to(x_cl.dtype).contiguous())
        return hip.conv3x3s2_cl(x_cl, wk, _f32(conv.bias))
    if (torch.is_grad_enabled() and (x_cl.requires_grad or conv.weight.requires_grad) and conv.kernel_size == (3, 3)
            and conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1] and conv.groups == 1
            and conv.dilation == (1, 1)):
        return _ConvIm2colCL.apply(x_cl, conv.weight, conv.bias, conv.stride, conv.padding)
    x = from_cl(x_cl)
    b = None if conv.bias is None else conv.bias.to(x.dtype)
    w = conv.weight.to(x.dtype).contiguous(memory_format=torch.channels_last)
    return to_cl(F.conv2d(x, w, b, stride=conv.stride, padding=conv.padding))


class VSSMEncoder(nn.Module):
    """vmamba.py:399-518 (VMamba-B backbone: dims 128..1024, depths [2,2,15,2])."""

    def __init__(self, patch_size=4, in_chans=3, depths=[2, 2, 15, 2], dims=128, drop_path_rate=0.6,
                 patch_norm=True, norm_layer="LN2D", posembed=False, imgsize=224, **kwargs):
        super().__init__()
        assert norm_layer.lower() == "ln2d" and patch_norm and not posembed
        self.channel_first = True
        self.num_layers = len(depths)
        if isinstance(dims, int):
            dims = [int(dims * 2 ** i) for i in range(self.num_layers)]
        self.num_features = dims[-1]
        self.dims = dims
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.pos_embed = None
        stride = patch_size // 2
        self.patch_embed = nn.Sequential(
            nn.Conv2d(in_chans, dims[0] // 2, kernel_size=stride + 1, stride=stride, padding=1),
            nn.Identity(), LayerNorm2d(dims[0] // 2), nn.Identity(), nn.GELU(),
            nn.Conv2d(dims[0] // 2, dims[0], kernel_size=stride + 1, stride=stride, padding=1),
            nn.Identity(), LayerNorm2d(dims[0]))
        self.layers = nn.ModuleList()
        self.downsample = nn.ModuleList()
        for i in range(self.num_layers):
            if i < self.num_layers - 1:
                self.downsample.append(nn.Sequential(
                    nn.Identity(), nn.Conv2d(dims[i], dims[i] * 2, kernel_size=3, stride=2, padding=1),
                    nn.Identity(), LayerNorm2d(dims[i] * 2)))
            else:
                self.downsample.append(nn.Identity())
            blocks = [VSSBlock(hidden_dim=dims[i], drop_path=dp, norm_layer=LayerNorm2d, channel_first=True)
                      for dp in dpr[sum(depths[:i]):sum(depths[:i + 1])]]
            self.layers.append(nn.Sequential(OrderedDict(blocks=nn.Sequential(*blocks))))
        self.apply(_init_weights)

    def _forward_cl(self, x_img, on_stage=None):
        """x_img: NCHW image batch.  Returns [image, s1, s2, s3, s4] with s* channels-last (B,H,W,C).
        on_stage(i, feature) is called as soon as stage i's output exists (the decoder's guide branch of that
        resolution starts there, on its own stream)."""
        feats = [x_img]
        pe = self.patch_embed
        if (_infer(x_img, pe[0].weight) and pe[0].in_channels == 3 and pe[0].out_channels == 64
                and pe[0].kernel_size == (3, 3) and pe[0].stride == (2, 2) and x_img.is_contiguous()):
            act_dtype = pe[5].weight.dtype if pe[5].weight.dtype != torch.float32 else x_img.dtype
            x = hip.stem_conv_ln_gelu(x_img, _f32(pe[0].weight), _f32(pe[0].bias), _f32(pe[2].weight),
                                      _f32(pe[2].bias), pe[2].eps, act_dtype)
        else:
            x = _conv_cl(pe[0], to_cl(x_img))
            x = pe[2]._forward_cl(x, act=hip.ACT_GELU)
        x = _conv_cl(pe[5], x)
        x = pe[7]._forward_cl(x)
        for s, layer in enumerate(self.layers):
            x = _run_blocks(layer.blocks, x)
            feats.append(x)
            if on_stage is not None:
                on_stage(s, x)
            if s < self.num_layers - 1:
                ds = self.downsample[s]
                x = ds[3]._forward_cl(_conv_cl(ds[1], x))
        return feats

    def forward(self, x):
        _need_device(x)
        feats = self._forward_cl(x)
        return [feats[0]] + [from_cl(f) for f in feats[1:]]


def load_pretrained_Base(model, ckpt_path=""):
    """vmamba.py:707-732: load a VMamba classification checkpoint (`{"model": state_dict}`) into the encoder.  Same rules:
    keys containing "classifier" are skipped; `layers.<i>.downsample.*` is renamed `downsample.<i>.*` and must then exist
    (AssertionError otherwise); a key the encoder does not have is reported and ignored; a shape mismatch is an
    AssertionError -- except a 1x1-conv weight `(out, in, 1, 1)` for a Linear2d `(out, in)`, which is viewed (the
    reshape of modules.py:15-19); parameters the checkpoint does not cover keep their initial values and
    are reported."""
    import re
    print(f"Loading weights from: {ckpt_path}")
    ckpt = torch.load(ckpt_path, map_location="cpu")
    model_dict = model.state_dict()
    loaded = set()
    for k, v in ckpt["model"].items():
        if "classifier" in k:
            print(f"Passing weights: {k}")
            continue
        if "downsample" in k:
            i_ds = int(re.findall(r"layers\.(\d+)\.downsample", k)[0])
            k = k.replace(f"layers.{i_ds}.downsample", f"downsample.{i_ds}")
            assert k in model_dict, k
        if k in model_dict:
            want = model_dict[k].shape
            if v.shape != want and v.dim() == 4 and tuple(v.shape[2:]) == (1, 1) and tuple(v.shape[:2]) == tuple(want):
                v = v.view(want)          # (out,in,1,1) conv weight -> Linear2d
            assert v.shape == want, f"module: {k} Shape mismatch: {v.shape} vs {want}"
            model_dict[k] = v
            loaded.add(k)
        else:
            print(f"Module can not find: {k}")
    model.load_state_dict(model_dict)
    for k in model_dict:
        if k not in loaded:
            print(f"Module {k} has not been inited!")
    return model
