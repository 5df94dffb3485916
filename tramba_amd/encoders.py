"""Swin-B and PVTv2-b4 encoders of Tramba-S / Tramba-P (Trambav6_enc.py:167-192), state_dict-compatible with the
reference's Models/encoder/{swin_encoder,pvtv2_encoder}.py (so `swin_base_patch4_window12_384_22k.pth` /
`pvt_v2_b4.pth` and Tramba-S / Tramba-P checkpoints load by name).

Token-major throughout: a (B, H*W, C) token tensor IS a channels-last (B, H, W, C) map, which is what the
decoder's HIP kernels take -- the reference's reshape/permute/contiguous hand-offs disappear.  In inference the
LayerNorms, every Linear (bias, GELU and the residual add fused in the GEMM epilogue) and PVT's depth-wise 3x3 run on
the library's HIP kernels; attention itself is stock `scaled_dot_product_attention` (no new kernels),
with Swin's relative-position bias and shift mask folded into one cached additive mask per block.  With autograd on
everything is stock torch ops on the device.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip
from .modules import DropPath, _cache, _f32, _infer


def _lin(m: nn.Linear, x, act=hip.ACT_NONE, residual=None):
    if _infer(x, m.weight):
        w = m.weight if m.weight.dtype == x.dtype else _cache(m).get(("w", x.dtype), (m.weight,), lambda: m.weight.to(x.dtype))
        return hip.linear_cl(x.contiguous(), w, None if m.bias is None else _f32(m.bias), residual, act)
    y = F.linear(x, m.weight.to(x.dtype), None if m.bias is None else m.bias.to(x.dtype))
    if act == hip.ACT_GELU:
        y = F.gelu(y)
    return y if residual is None else y + residual


def _ln(m: nn.LayerNorm, x):
    if _infer(x, m.weight):
        return hip.layernorm_cl(x.contiguous(), _f32(m.weight), _f32(m.bias), m.eps)
    return F.layer_norm(x.float(), m.normalized_shape, m.weight.float(), m.bias.float(), m.eps).to(x.dtype)


def _conv(m: nn.Conv2d, x):
    return F.conv2d(x, m.weight.to(x.dtype), None if m.bias is None else m.bias.to(x.dtype), m.stride, m.padding, 1, m.groups)


def _init_linear_ln(m):
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)


# ============================================================================= PVTv2 (pvtv2_encoder.py)
class _PvtDW(nn.Module):
    """pvtv2_encoder.py:373-384 (`mlp.dwconv.dwconv`): depth-wise 3x3 with bias on the token map."""

    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)


class _PvtMlp(nn.Module):
    """pvtv2_encoder.py:19-54: fc1 -> dw3x3 -> GELU -> fc2."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.dwconv = _PvtDW(hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x, h, w, residual=None):
        b, n, _ = x.shape
        y = _lin(self.fc1, x)
        conv = self.dwconv.dwconv
        if _infer(y, conv.weight):
            taps = _cache(self).get("taps", (conv.weight,), lambda: conv.weight.detach().float().reshape(-1, 9).t().contiguous())
            y = hip.dwconv_cl(y.view(b, h, w, -1), taps, _f32(conv.bias), hip.ACT_GELU).view(b, n, -1)
        else:
            y = _conv(conv, y.transpose(1, 2).reshape(b, -1, h, w)).flatten(2).transpose(1, 2)
            y = F.gelu(y)
        return _lin(self.fc2, y, residual=residual)


class _PvtAttention(nn.Module):
    """pvtv2_encoder.py:57-116: spatial-reduction attention (keys / values from an sr x sr strided conv of the map)."""

    def __init__(self, dim, num_heads, qkv_bias, sr_ratio):
        super().__init__()
        assert dim % num_heads == 0
        self.dim, self.num_heads, self.sr_ratio = dim, num_heads, sr_ratio
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)

    def forward(self, x, h, w, residual=None):
        b, n, c = x.shape
        nh = self.num_heads
        q = _lin(self.q, x).view(b, n, nh, c // nh).transpose(1, 2)
        if self.sr_ratio > 1:
            xr = _conv(self.sr, x.transpose(1, 2).reshape(b, c, h, w)).flatten(2).transpose(1, 2)
            xr = _ln(self.norm, xr)
        else:
            xr = x
        kv = _lin(self.kv, xr).view(b, -1, 2, nh, c // nh).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(q, kv[0], kv[1], scale=self.scale)
        return _lin(self.proj, o.transpose(1, 2).reshape(b, n, c), residual=residual)


class _PvtBlock(nn.Module):
    """pvtv2_encoder.py:119-156."""

    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, drop_path, sr_ratio, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _PvtAttention(dim, num_heads, qkv_bias, sr_ratio)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _PvtMlp(dim, int(dim * mlp_ratio))

    def forward(self, x, h, w):
        if not self.training:                      # drop_path is the identity: residual adds ride in the GEMM epilogues
            x = self.attn(_ln(self.norm1, x), h, w, residual=x)
            return self.mlp(_ln(self.norm2, x), h, w, residual=x)
        x = x + self.drop_path(self.attn(_ln(self.norm1, x), h, w))
        return x + self.drop_path(self.mlp(_ln(self.norm2, x), h, w))


class _OverlapPatchEmbed(nn.Module):
    """pvtv2_encoder.py:159-199: overlapping strided conv + LayerNorm -> tokens."""

    def __init__(self, patch, stride, cin, dim):
        super().__init__()
        self.proj = nn.Conv2d(cin, dim, kernel_size=patch, stride=stride, padding=patch // 2)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        x = _conv(self.proj, x)
        h, w = x.shape[-2:]
        return _ln(self.norm, x.flatten(2).transpose(1, 2)), h, w


class PyramidVisionTransformerImpr(nn.Module):
    """pvtv2_encoder.py:202-366.  forward(x) -> [stage4, stage3, stage2, stage1] NCHW maps (deepest first, :358)."""

    def __init__(self, embed_dims=(64, 128, 256, 512), num_heads=(1, 2, 4, 8), mlp_ratios=(4, 4, 4, 4), qkv_bias=False,
                 drop_path_rate=0.0, depths=(3, 4, 6, 3), sr_ratios=(8, 4, 2, 1), eps=1e-5, in_chans=3):
        super().__init__()
        self.depths = list(depths)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        cur = 0
        for s in range(4):
            embed = _OverlapPatchEmbed(7 if s == 0 else 3, 4 if s == 0 else 2, in_chans if s == 0 else embed_dims[s - 1],
                                       embed_dims[s])
            blocks = nn.ModuleList([_PvtBlock(embed_dims[s], num_heads[s], mlp_ratios[s], qkv_bias, dpr[cur + i], sr_ratios[s],
                                              eps) for i in range(depths[s])])
            cur += depths[s]
            setattr(self, f"patch_embed{s + 1}", embed)
            setattr(self, f"block{s + 1}", blocks)
            setattr(self, f"norm{s + 1}", nn.LayerNorm(embed_dims[s], eps=eps))
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        _init_linear_ln(m)
        if isinstance(m, nn.Conv2d):                                    # pvtv2_encoder.py:269-274
            fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
            m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
            if m.bias is not None:
                m.bias.data.zero_()

    def features_cl(self, x):
        """[(B, H, W, C) channels-last map per stage], shallow first."""
        outs = []
        b = x.shape[0]
        for s in range(1, 5):
            x, h, w = getattr(self, f"patch_embed{s}")(x)
            for blk in getattr(self, f"block{s}"):
                x = blk(x, h, w)
            x = _ln(getattr(self, f"norm{s}"), x)
            outs.append(x.view(b, h, w, -1))
            if s < 4:
                x = outs[-1].permute(0, 3, 1, 2)                        # the next strided conv reads NCHW (a view)
        return outs

    def forward(self, x):
        return [o.permute(0, 3, 1, 2).contiguous() for o in self.features_cl(x)][::-1]


def pvt_v2_b4():
    """pvtv2_encoder.py:433-439."""
    return PyramidVisionTransformerImpr(embed_dims=(64, 128, 320, 512), num_heads=(1, 2, 5, 8), mlp_ratios=(8, 8, 4, 4),
                                        qkv_bias=True, eps=1e-6, depths=(3, 8, 27, 3), sr_ratios=(8, 4, 2, 1),
                                        drop_path_rate=0.1)


# ============================================================================= Swin (swin_encoder.py)
def _windows(x, ws):
    """(B, H, W, C) -> (B, nW, ws*ws, C)   (swin_encoder.py:36-48, batch kept as its own axis)"""
    b, h, w, c = x.shape
    return x.view(b, h // ws, ws, w // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(b, (h // ws) * (w // ws), ws * ws, c)


def _unwindows(xw, ws, h, w):
    """inverse of _windows (swin_encoder.py:51-65)"""
    b = xw.shape[0]
    return xw.view(b, h // ws, w // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, -1)


class _WindowAttention(nn.Module):
    """swin_encoder.py:68-147: window attention with a learned relative-position bias."""

    def __init__(self, dim, ws, num_heads, qkv_bias=True):
        super().__init__()
        self.dim, self.ws, self.num_heads = dim, ws, num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        ar = torch.arange(ws)
        coords = torch.stack(torch.meshgrid(ar, ar, indexing="ij")).flatten(1)             # (2, ws*ws)
        rel = coords[:, :, None] - coords[:, None, :] + (ws - 1)                           # (2, N, N), each in [0, 2ws-2]
        self.register_buffer("relative_position_index", rel[0] * (2 * ws - 1) + rel[1])
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def bias_mask(self, shift_mask, dtype):
        """(1, 1 or nW, nH, N, N) additive term: relative-position bias (+ the shifted-window mask)"""
        def build():
            n = self.ws * self.ws
            bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(n, n, -1)
            bias = bias.permute(2, 0, 1)[None, None].float()                                # (1, 1, nH, N, N)
            if shift_mask is not None:
                bias = bias + shift_mask[None, :, None].float()                            # (1, nW, nH, N, N)
            return bias.to(dtype).contiguous()
        if _infer(self.relative_position_bias_table):
            return _cache(self).get(("bias", dtype), (self.relative_position_bias_table,), build)
        return build()

    def forward(self, xw, shift_mask):
        b, nw, n, c = xw.shape
        nh = self.num_heads
        qkv = _lin(self.qkv, xw).view(b, nw, n, 3, nh, c // nh).permute(3, 0, 1, 4, 2, 5)   # (3, B, nW, nH, N, hd)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], attn_mask=self.bias_mask(shift_mask, xw.dtype),
                                           scale=self.scale)
        return _lin(self.proj, o.transpose(2, 3).reshape(b, nw, n, c))


class _SwinMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x, residual=None):
        return _lin(self.fc2, _lin(self.fc1, x, act=hip.ACT_GELU), residual=residual)


class SwinTransformerBlock(nn.Module):
    """swin_encoder.py:166-273."""

    def __init__(self, dim, input_resolution, num_heads, window_size, shift_size, mlp_ratio, drop_path):
        super().__init__()
        self.input_resolution = input_resolution
        if min(input_resolution) <= window_size:          # one window covers the map: no partition, no shift (:197-200)
            shift_size, window_size = 0, min(input_resolution)
        self.window_size, self.shift_size = window_size, shift_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _WindowAttention(dim, window_size, num_heads)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _SwinMlp(dim, int(dim * mlp_ratio))
        mask = None
        if shift_size > 0:                                 # regions that wrap around under the cyclic shift (:213-232)
            h, w = input_resolution
            region = torch.zeros(1, h, w, 1)
            cuts = (slice(0, -window_size), slice(-window_size, -shift_size), slice(-shift_size, None))
            for i, hs in enumerate(cuts):
                for j, wsl in enumerate(cuts):
                    region[:, hs, wsl, :] = i * 3 + j
            ids = _windows(region, window_size)[0, :, :, 0]                                 # (nW, N)
            diff = ids[:, None, :] - ids[:, :, None]
            mask = torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))
        self.register_buffer("attn_mask", mask)

    def forward(self, x):
        h, w = self.input_resolution
        b, l, c = x.shape
        ws, sh = self.window_size, self.shift_size
        y = _ln(self.norm1, x).view(b, h, w, c)
        if sh > 0:
            y = torch.roll(y, shifts=(-sh, -sh), dims=(1, 2))
        y = _unwindows(self.attn(_windows(y, ws), self.attn_mask), ws, h, w)
        if sh > 0:
            y = torch.roll(y, shifts=(sh, sh), dims=(1, 2))
        y = y.reshape(b, l, c)
        if not self.training:
            x = x + y
            return self.mlp(_ln(self.norm2, x), residual=x)
        x = x + self.drop_path(y)
        return x + self.drop_path(self.mlp(_ln(self.norm2, x)))


class PatchMerging(nn.Module):
    """swin_encoder.py:294-331: 2x2 neighbours -> 4C channels -> LayerNorm -> Linear(4C, 2C)."""

    def __init__(self, input_resolution, dim):
        super().__init__()
        self.input_resolution = input_resolution
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x):
        h, w = self.input_resolution
        b, l, c = x.shape
        x = x.view(b, h // 2, 2, w // 2, 2, c)
        # channel order of the reference's cat([x0, x1, x2, x3]): (row parity, col parity) = (0,0), (1,0), (0,1), (1,1)
        x = x.permute(0, 1, 3, 4, 2, 5).reshape(b, (h // 2) * (w // 2), 4 * c)
        return _lin(self.reduction, _ln(self.norm, x))


class BasicLayer(nn.Module):
    """swin_encoder.py:343-399."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio, drop_path, downsample):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2,
                                 mlp_ratio, drop_path[i]) for i in range(depth)])
        self.downsample = PatchMerging(input_resolution, dim) if downsample else None

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return x if self.downsample is None else self.downsample(x)


class _SwinPatchEmbed(nn.Module):
    """swin_encoder.py:413-450."""

    def __init__(self, img_size, patch, cin, dim):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patches_resolution = [img_size // patch, img_size // patch]
        self.proj = nn.Conv2d(cin, dim, kernel_size=patch, stride=patch)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        if tuple(x.shape[-2:]) != self.img_size:
            raise RuntimeError(f"Input image size {tuple(x.shape[-2:])} doesn't match model {self.img_size}")
        return _ln(self.norm, _conv(self.proj, x).flatten(2).transpose(1, 2))


class SwinTransformer(nn.Module):
    """swin_encoder.py:461-594.  forward(x) -> [layer3 out, layer2 out, layer1 out, layer0 out, patch embedding] as NCHW
    maps (deepest first, :590-594); Tramba-S uses all but the first (Trambav6_enc.py:210-211), so `features_cl(x,
    last=False)` skips the last stage's blocks, whose output nothing reads."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24),
                 window_size=7, mlp_ratio=4.0, drop_path_rate=0.1):
        super().__init__()
        self.num_layers = len(depths)
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        self.patch_embed = _SwinPatchEmbed(img_size, patch_size, in_chans, embed_dim)
        res = self.patch_embed.patches_resolution
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList([
            BasicLayer(int(embed_dim * 2 ** i), (res[0] // 2 ** i, res[1] // 2 ** i), depths[i], num_heads[i], window_size,
                       mlp_ratio, dpr[sum(depths[:i]):sum(depths[:i + 1])], downsample=i < self.num_layers - 1)
            for i in range(self.num_layers)])
        self.apply(_init_linear_ln)

    def features_cl(self, x, last=True):
        """[(B, H, W, C)] shallow first: patch embedding, then each layer's output (after its PatchMerging)."""
        b = x.shape[0]
        x = self.patch_embed(x)
        feats = []
        for i, layer in enumerate(self.layers):
            side = int(round(math.sqrt(x.shape[1])))
            feats.append(x.view(b, side, side, -1))
            if i == self.num_layers - 1 and not last:
                return feats
            x = layer(x)
        side = int(round(math.sqrt(x.shape[1])))
        feats.append(x.view(b, side, side, -1))
        return feats

    def forward(self, x):
        return [f.permute(0, 3, 1, 2).contiguous() for f in self.features_cl(x)][::-1]
