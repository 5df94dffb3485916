"""Oracle: operator-level CPU restatements (plain torch-CPU, any float dtype).

TEST INFRASTRUCTURE ONLY.  All tensors are logical NCHW like the reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import scan_tables
from .selective_scan import selective_scan


# ----------------------------------------------------------------------------- scans
def _tbl(family, h, w, device):
    return torch.from_numpy(scan_tables.table(family, h, w)).to(device)


def cross_scan(x: torch.Tensor, family: str) -> torch.Tensor:
    """(B,C,H,W) -> (B,K,C,L): xs[b,k,c,l] = x[b,c,table[k,l]].
    csms6s.py:13-22 (raster), :64-72 (dilation), :113-121 (window), :161-172 (helix)."""
    b, c, h, w = x.shape
    t = _tbl(family, h, w, x.device)
    return x.reshape(b, c, h * w)[:, :, t].permute(0, 2, 1, 3).contiguous()


def cross_merge(ys: torch.Tensor, family: str, h: int, w: int) -> torch.Tensor:
    """(B,K,C,L) -> (B,C,L): y[b,c,p] = sum_{k,l: table[k,l]==p} ys[b,k,c,l].
    csms6s.py:34-42, :83-91, :132-140, :188-201 (+ scatter_add_ in SpiralLine.py:109-133,
    Window.py:62-86, Dilation.py:72-96).  Deterministic summation order: k then l."""
    b, k, c, l = ys.shape
    t = _tbl(family, h, w, ys.device)
    out = torch.zeros(b, c, h * w, dtype=ys.dtype, device=ys.device)
    for kk in range(k):
        out.index_add_(2, t[kk], ys[:, kk])
    return out


# ----------------------------------------------------------------------------- basic modules
def linear2d(x, weight, bias=None):
    """modules.py:10-19: 1x1 conv with an (out,in) [or (out,in,1,1)] weight."""
    w = weight.reshape(weight.shape[0], -1)
    return F.conv2d(x, w[:, :, None, None], bias)


def layernorm2d(x, weight, bias, eps=1e-5):
    """modules.py:22-27: LayerNorm over the channel dim of an NCHW tensor."""
    y = F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), weight, bias, eps)
    return y.permute(0, 3, 1, 2).contiguous()


def pixel_shuffle_groups(x, p):
    """einops 'b (p1 p2 c) h w -> b c (h p1) (w p2)' (modules.py:212, 246, 690)."""
    b, c1, h, w = x.shape
    c = c1 // (p * p)
    return x.reshape(b, p, p, c, h, w).permute(0, 3, 4, 1, 5, 2).reshape(b, c, h * p, w * p)


def dct_matrix(n, dtype=torch.float32):
    """DCT_2D.py:37-45 / :61-69: W[v,j] = cos(pi (j+.5) v / n)/sqrt(n) * (sqrt2 if v>0).
    Built in Python floats then stored in fp32, like the reference buffer."""
    m = torch.zeros(n, n, dtype=torch.float32)
    for v in range(n):
        for j in range(n):
            val = math.cos(math.pi * (0.5 + j) * v / n) / math.sqrt(n)
            if v != 0:
                val = val * math.sqrt(2)
            m[v, j] = val
    return m.to(dtype)


def dct2d_split(x, wx, wy):
    """DCT_2D.py:12-29: Y = Wy X Wx^T; high = Y[h/2:, w/2:], low = Y[:h/2, :w/2].
    Returns (high, low) like the reference."""
    y = torch.einsum("bchw,uw->bchu", x, wx.to(x.dtype))
    y = torch.einsum("bchu,vh->bcvu", y, wy.to(x.dtype))
    hh, hw = y.shape[2] // 2, y.shape[3] // 2
    return y[:, :, hh:, hw:], y[:, :, :hh, :hw]


# ----------------------------------------------------------------------------- loss / metric
def iou_loss(pred, mask):
    """utils/loss.py:6-11."""
    p = torch.sigmoid(pred)
    inter = (p * mask).sum(dim=(2, 3))
    union = (p + mask).sum(dim=(2, 3))
    return (1 - (inter + 1) / (union - inter + 1)).mean()


def tramba_loss(outputs, label):
    """train.py:76-85 (4 outputs) / :58-66 (3 outputs): every output is bilinearly resized to
    the label size (the last already has it), loss = sum_i BCEWithLogits + IoU, weights 1."""
    hh, ww = label.shape[-2:]
    total = 0
    for o in outputs:
        if o.shape[-2:] != (hh, ww):
            o = F.interpolate(o, (hh, ww), mode="bilinear")
        total = total + F.binary_cross_entropy_with_logits(o, label) + iou_loss(o, label)
    return total


def adam_steps(params, grads_per_step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """The optimizer of train.py:266-280 -- `torch.optim.Adam(param_groups, lr)`, all defaults -- restated from its update rule
    (torch/optim/adam.py, _single_tensor_adam: amsgrad off, maximize off), in fp64: for step t = 1, 2, ... and every tensor
      g += weight_decay p;  m = m + (1 - b1)(g - m);  v = b2 v + (1 - b2) g g;
      p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).
    params: tensors; grads_per_step: one list of gradients per step.  Returns (params, exp_avg, exp_avg_sq) after the steps.
    Pinned against torch.optim.Adam itself on the CPU (tests/test_oracle.py)."""
    b1, b2 = betas
    p = [x.detach().double().clone() for x in params]
    m = [torch.zeros_like(x) for x in p]
    v = [torch.zeros_like(x) for x in p]
    for t, grads in enumerate(grads_per_step, 1):
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        for i, g in enumerate(grads):
            g = g.detach().double()
            if weight_decay:
                g = g + weight_decay * p[i]
            m[i] = m[i] + (1 - b1) * (g - m[i])
            v[i] = b2 * v[i] + (1 - b2) * g * g
            p[i] = p[i] - (lr / bc1) * m[i] / (v[i].sqrt() / math.sqrt(bc2) + eps)
    return p, m, v


def mae_metric(pred: np.ndarray, gt: np.ndarray) -> float:
    """Evaluation/metrics.py:13-19, 88-104: min-max normalise pred, gt -> bool, mean |pred-gt|."""
    gt = gt.astype(bool)
    if pred.max() != pred.min():
        pred = (pred - pred.min()) / (pred.max() - pred.min())
    return float(np.mean(np.abs(pred - gt)))


# ----------------------------------------------------------------------------- SS2D core
def ss2d_core(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, family, merge=True):
    """vmamba.py:230-257 (non-cascade branch, no_einsum=True): scan -> grouped x_proj ->
    split [R,N,N] -> grouped dt_proj -> selective scan -> merge.  Returns (B, D, H, W)
    BEFORE out_norm; merge=False: the per-direction scan outputs (B, K, L, D), sequence order,
    channels last (what the HIP scan kernel writes before its merge kernel)."""
    b, d, h, w = x.shape
    k, _, r = dt_projs_weight.shape
    n = A_logs.shape[1]
    l = h * w
    xs = cross_scan(x, family)  # (B,K,D,L)
    x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, x_proj_weight)
    dts, bs, cs = torch.split(x_dbl, [r, n, n], dim=2)
    dts = torch.einsum("bkrl,kdr->bkdl", dts, dt_projs_weight)
    a = -torch.exp(A_logs.float()).to(x.dtype)
    ys = selective_scan(
        xs.reshape(b, k * d, l), dts.reshape(b, k * d, l).contiguous(), a,
        bs.contiguous(), cs.contiguous(), Ds.to(x.dtype), dt_projs_bias.reshape(-1).to(x.dtype), True)
    if not merge:
        return ys.reshape(b, k, d, l).permute(0, 1, 3, 2).contiguous()
    y = cross_merge(ys.reshape(b, k, d, l), family, h, w)
    return y.reshape(b, d, h, w)
