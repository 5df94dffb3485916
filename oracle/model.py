"""Oracle: functional CPU forward of the Tramba models over a reference-named state_dict.

TEST INFRASTRUCTURE ONLY.  Deliberately *functional* (dict of tensors in, tensors out) so
it shares no class with the product's nn.Module tree; it accepts the reference's own
``state_dict()`` as well as the product's (same 679 names for Tramba-V).

Every function cites the reference code it restates.  Eval-mode semantics (DropPath and
Dropout are identities; BatchNorm uses running statistics).
"""
import torch
import torch.nn.functional as F

from . import ops


class SD:
    """state_dict view with a name prefix."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __call__(self, name):
        return self.sd[self.prefix + name]

    def has(self, name):
        return (self.prefix + name) in self.sd

    def sub(self, name):
        return SD(self.sd, self.prefix + name + ".")


def _ln(p: SD, x):
    return ops.layernorm2d(x, p("weight"), p("bias"))


def _lin(p: SD, x):
    return ops.linear2d(x, p("weight"), p("bias") if p.has("bias") else None)


def ss2d(p: SD, x, family):
    """vmamba.py:275-291 forwardv2 + :230-273 forward_corev2 (channel_first, disable_z)."""
    x = ops.linear2d(x, p("in_proj.weight"))
    wc = p("conv2d.weight")
    x = F.conv2d(x, wc, p("conv2d.bias") if p.has("conv2d.bias") else None,
                 padding=(wc.shape[-1] - 1) // 2, groups=wc.shape[0])
    x = F.silu(x)
    y = ops.ss2d_core(x, p("x_proj_weight"), p("dt_projs_weight"), p("dt_projs_bias"),
                      p("A_logs"), p("Ds"), family)
    y = _ln(p.sub("out_norm"), y)
    y = F.gelu(y)
    return ops.linear2d(y, p("out_proj.weight"))


def mlp(p: SD, x):
    """modules.py:134-153."""
    return _lin(p.sub("fc2"), F.gelu(_lin(p.sub("fc1"), x)))


def vss_block(p: SD, x):
    """vmamba.py:384-396 (pre-norm)."""
    x = x + ss2d(p.sub("op"), _ln(p.sub("norm"), x), "raster")
    return x + mlp(p.sub("mlp"), _ln(p.sub("norm2"), x))


def vssm_encoder(p: SD, x, depths=(2, 2, 15, 2)):
    """vmamba.py:474-489 (patch embed), :505-518 (forward), :450-457 (downsample)."""
    feats = [x]
    pe = p.sub("patch_embed")
    x = F.conv2d(x, pe("0.weight"), pe("0.bias"), stride=2, padding=1)
    x = F.gelu(_ln(pe.sub("2"), x))
    x = F.conv2d(x, pe("5.weight"), pe("5.bias"), stride=2, padding=1)
    x = _ln(pe.sub("7"), x)
    for s, depth in enumerate(depths):
        for j in range(depth):
            x = vss_block(p.sub(f"layers.{s}.blocks.{j}"), x)
        feats.append(x)
        if s < len(depths) - 1:
            ds = p.sub(f"downsample.{s}")
            x = F.conv2d(x, ds("1.weight"), ds("1.bias"), stride=2, padding=1)
            x = _ln(ds.sub("3"), x)
    return feats


def dwms_mlp(p: SD, x):
    """vmamba.py:622-629: fc1 -> h + dw3(h) + dw5(h) + dw7(h) -> GELU -> fc2."""
    h = _lin(p.sub("fc1"), x)
    acc = h
    for ksz in (3, 5, 7):
        q = p.sub(f"dwc{ksz}.dw_conv")
        acc = acc + F.conv2d(h, q("weight"), q("bias"), padding=(ksz - 1) // 2, groups=h.shape[1])
    return _lin(p.sub("fc2"), F.gelu(acc))


def multiscale_decoder_block(p: SD, x):
    """vmamba.py:700-704 (Helix-SS2D K=8 + DWMSMlp)."""
    x = x + ss2d(p.sub("op"), _ln(p.sub("norm1"), x), "helix")
    return x + dwms_mlp(p.sub("mlp"), _ln(p.sub("norm2"), x))


def _expand_shuffle_norm(p: SD, x, scale):
    """modules.py:209-218 (PatchExpand), :240-249 (FinalPatchExpand_X4), :687-696 (FreqExpand2D)."""
    x = ops.linear2d(x, p("expand.weight"))
    x = ops.pixel_shuffle_groups(x, scale)
    return _ln(p.sub("norm"), x)


def freq_ss2d(p: SD, x):
    """freq_mamba.py:46-57."""
    high, low = ops.dct2d_split(x, p("DCT2D.dct_x.weight"), p("DCT2D.dct_y.weight"))
    high = _expand_shuffle_norm(p.sub("h_expand"), high, 2)
    low = _expand_shuffle_norm(p.sub("l_expand"), low, 2)
    hifi = ss2d(p.sub("h_ssm"), high, "window")
    lofi = ss2d(p.sub("l_ssm"), low, "dilation")
    attn = ops.linear2d(torch.cat((hifi, lofi), 1), p("concat_back_dim.weight"))
    return torch.sigmoid(attn) * x


def freq_block(p: SD, x):
    """freq_mamba.py:79-82."""
    x = x + freq_ss2d(p.sub("attn"), _ln(p.sub("norm1"), x))
    return x + mlp(p.sub("mlp"), _ln(p.sub("norm2"), x))


def vssm_decoder(p: SD, skips, depth_per_stage=2):
    """Trambav6.py:114-139 / Trambav6_enc.py:131-159 (deep supervision on)."""
    n_stage = len(skips) - 1  # skips[0] is the raw image
    x_low = skips[-1]
    outs = []
    for s in range(n_stage):
        if s < n_stage - 1:
            x = _expand_shuffle_norm(p.sub(f"expand_layers.{s}"), x_low, 2)
            mid = freq_block(p.sub(f"guide_layers.{s}"), skips[-(s + 2)])
            x = _lin(p.sub(f"concat_back_dim.{s}"), torch.cat((x, mid), 1))
            for j in range(depth_per_stage):
                x = multiscale_decoder_block(p.sub(f"stage_layers.{s}.blocks.{j}"), x)
        else:
            x = _expand_shuffle_norm(p.sub(f"expand_layers.{s}"), x_low, 4)
        seg = p.sub(f"seg_layers.{s}")
        outs.append(F.conv2d(x, seg("weight"), seg("bias")))
        x_low = x
    return outs


def tramba_v(sd, x):
    """Trambav6.py:151-154 BaseUMamba.forward."""
    skips = vssm_encoder(SD(sd, "vssm_encoder."), x)
    return vssm_decoder(SD(sd, "decoder."), skips)


# ----------------------------------------------------------------------------- Tramba-R
def _bn(p: SD, x):
    return F.batch_norm(x, p("running_mean"), p("running_var"), p("weight"), p("bias"), False, 0.0, 1e-5)


def _bottleneck(p: SD, x, stride):
    """resnet_encoder.py:62-79."""
    out = F.relu(_bn(p.sub("bn1"), F.conv2d(x, p("conv1.weight"))))
    out = F.relu(_bn(p.sub("bn2"), F.conv2d(out, p("conv2.weight"), stride=stride, padding=1)))
    out = _bn(p.sub("bn3"), F.conv2d(out, p("conv3.weight")))
    if p.has("downsample.0.weight"):
        x = _bn(p.sub("downsample.1"), F.conv2d(x, p("downsample.0.weight"), stride=stride))
    return F.relu(out + x)


def resnet50_encoder(p: SD, x):
    """resnet_encoder.py:81-110; returns (out5, out4, out3, out2, out1) like the reference."""
    o1 = F.relu(_bn(p.sub("bn1"), F.conv2d(x, p("conv1.weight"), stride=2, padding=3)))
    o1 = F.max_pool2d(o1, kernel_size=3, stride=2, padding=1)
    feats, cur = [o1], o1
    for li, (blocks, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2)), start=1):
        for j in range(blocks):
            cur = _bottleneck(p.sub(f"layer{li}.{j}"), cur, stride if j == 0 else 1)
        feats.append(cur)
    return tuple(feats[::-1])


def tramba_r(sd, x):
    """Trambav6_enc.py:208-219: skips = [x] + outs[1:-1][::-1]."""
    outs = resnet50_encoder(SD(sd, "encoder."), x)
    skips = [x] + list(outs[1:-1][::-1])
    return vssm_decoder(SD(sd, "decoder."), skips)


# ----------------------------------------------------------------------------- Tramba-P (PVTv2-b4)
def _tok_ln(p: SD, x, eps):
    return F.layer_norm(x, (x.shape[-1],), p("weight"), p("bias"), eps)


def _tok_lin(p: SD, x):
    return F.linear(x, p("weight"), p("bias") if p.has("bias") else None)


def _pvt_attention(p: SD, x, h, w, heads, sr):
    """pvtv2_encoder.py:95-116: softmax(q k^T / sqrt(hd)) v with k, v from the sr-strided conv of the map."""
    b, n, c = x.shape
    hd = c // heads
    q = _tok_lin(p.sub("q"), x).reshape(b, n, heads, hd).transpose(1, 2)
    if sr > 1:
        m = F.conv2d(x.transpose(1, 2).reshape(b, c, h, w), p("sr.weight"), p("sr.bias"), stride=sr)
        x = _tok_ln(p.sub("norm"), m.flatten(2).transpose(1, 2), 1e-5)       # `self.norm = nn.LayerNorm(dim)`: default eps
    kv = _tok_lin(p.sub("kv"), x).reshape(b, -1, 2, heads, hd).permute(2, 0, 3, 1, 4)
    att = torch.softmax(q @ kv[0].transpose(-2, -1) * hd ** -0.5, dim=-1)
    return _tok_lin(p.sub("proj"), (att @ kv[1]).transpose(1, 2).reshape(b, n, c))


def _pvt_mlp(p: SD, x, h, w):
    """pvtv2_encoder.py:47-54, 373-384."""
    b, n, _ = x.shape
    y = _tok_lin(p.sub("fc1"), x)
    c = y.shape[-1]
    y = F.conv2d(y.transpose(1, 2).reshape(b, c, h, w), p("dwconv.dwconv.weight"), p("dwconv.dwconv.bias"), padding=1, groups=c)
    return _tok_lin(p.sub("fc2"), F.gelu(y.flatten(2).transpose(1, 2)))


def pvt_v2_b4_encoder(p: SD, x):
    """pvtv2_encoder.py:321-358 with the b4 configuration (:433-439); returns [stage1..stage4] NCHW (shallow first)."""
    heads, srs, depths = (1, 2, 5, 8), (8, 4, 2, 1), (3, 8, 27, 3)
    outs = []
    for s in range(4):
        pe = p.sub(f"patch_embed{s + 1}")
        k = pe("proj.weight").shape[-1]
        x = F.conv2d(x, pe("proj.weight"), pe("proj.bias"), stride=4 if s == 0 else 2, padding=k // 2)
        b, c, h, w = x.shape
        x = _tok_ln(pe.sub("norm"), x.flatten(2).transpose(1, 2), 1e-5)      # OverlapPatchEmbed's own LayerNorm: default eps
        for i in range(depths[s]):
            blk = p.sub(f"block{s + 1}.{i}")
            x = x + _pvt_attention(blk.sub("attn"), _tok_ln(blk.sub("norm1"), x, 1e-6), h, w, heads[s], srs[s])
            x = x + _pvt_mlp(blk.sub("mlp"), _tok_ln(blk.sub("norm2"), x, 1e-6), h, w)
        x = _tok_ln(p.sub(f"norm{s + 1}"), x, 1e-6).reshape(b, h, w, c).permute(0, 3, 1, 2)
        outs.append(x)
    return outs


def tramba_p(sd, x):
    """Trambav6_enc.py:208-217: the encoder returns deepest first, skips += outs[::-1]."""
    return vssm_decoder(SD(sd, "decoder."), [x] + pvt_v2_b4_encoder(SD(sd, "encoder."), x))


# ----------------------------------------------------------------------------- Tramba-S (Swin-B, window 12, 384)
def _swin_windows(x, ws):
    b, h, w, c = x.shape
    return x.reshape(b, h // ws, ws, w // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, c)


def _swin_block(p: SD, x, res, heads, ws, shift):
    """swin_encoder.py:236-273 + 116-147."""
    b, l, c = x.shape
    if res <= ws:
        ws, shift = res, 0
    y = _tok_ln(p.sub("norm1"), x, 1e-5).reshape(b, res, res, c)
    if shift:
        y = torch.roll(y, (-shift, -shift), (1, 2))
    win = _swin_windows(y, ws)                                                          # (B*nW, N, C)
    n, hd = ws * ws, c // heads
    qkv = _tok_lin(p.sub("attn.qkv"), win).reshape(-1, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
    att = (qkv[0] * hd ** -0.5) @ qkv[1].transpose(-2, -1)
    bias = p("attn.relative_position_bias_table")[p("attn.relative_position_index").reshape(-1)].reshape(n, n, heads)
    att = att + bias.permute(2, 0, 1)[None]
    if shift:
        mask = p("attn_mask")                                                           # (nW, N, N), 0 / -100
        nw = mask.shape[0]
        att = (att.reshape(b, nw, heads, n, n) + mask[None, :, None]).reshape(-1, heads, n, n)
    out = (torch.softmax(att, -1) @ qkv[2]).transpose(1, 2).reshape(-1, n, c)
    out = _tok_lin(p.sub("attn.proj"), out)
    out = out.reshape(b, res // ws, res // ws, ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(b, res, res, c)
    if shift:
        out = torch.roll(out, (shift, shift), (1, 2))
    x = x + out.reshape(b, l, c)
    return x + _tok_lin(p.sub("mlp.fc2"), F.gelu(_tok_lin(p.sub("mlp.fc1"), _tok_ln(p.sub("norm2"), x, 1e-5))))


def swin_b_encoder(p: SD, x, depths=(2, 2, 18, 2), heads=(4, 8, 16, 32), ws=12):
    """swin_encoder.py:563-594: features before every layer plus the last layer's output, returned deepest first."""
    x = F.conv2d(x, p("patch_embed.proj.weight"), p("patch_embed.proj.bias"), stride=4)
    b, c, res, _ = x.shape
    x = _tok_ln(p.sub("patch_embed.norm"), x.flatten(2).transpose(1, 2), 1e-5)
    feats = []
    for li, depth in enumerate(depths):
        feats.append(x.reshape(b, res, res, -1).permute(0, 3, 1, 2))
        for i in range(depth):
            x = _swin_block(p.sub(f"layers.{li}.blocks.{i}"), x, res, heads[li], ws, 0 if i % 2 == 0 else ws // 2)
        if li < len(depths) - 1:                                                        # PatchMerging, :310-331
            c = x.shape[-1]
            g = x.reshape(b, res, res, c)
            g = torch.cat([g[:, 0::2, 0::2], g[:, 1::2, 0::2], g[:, 0::2, 1::2], g[:, 1::2, 1::2]], -1).reshape(b, -1, 4 * c)
            dn = p.sub(f"layers.{li}.downsample")
            x = F.linear(_tok_ln(dn.sub("norm"), g, 1e-5), dn("reduction.weight"))
            res //= 2
    feats.append(x.reshape(b, res, res, -1).permute(0, 3, 1, 2))
    return feats[::-1]


def tramba_s(sd, x):
    """Trambav6_enc.py:208-211: skips += outs[1:][::-1] (the last layer's output is not used)."""
    outs = swin_b_encoder(SD(sd, "encoder."), x)
    return vssm_decoder(SD(sd, "decoder."), [x] + list(outs[1:][::-1]))
