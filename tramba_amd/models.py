"""Model assembly with the reference's module surface (Trambav6.py, Trambav6_enc.py, get_model.py).

``BaseUMamba.forward(x:(B,3,S,S)) -> [logits...]`` (last = full resolution), parameter names
``vssm_encoder.* / encoder.* / decoder.*`` (train.py:266-269 splits the optimizer on the substring
"encoder"), 679 state_dict entries for Tramba-V so ``load_state_dict(strict=True)`` of a reference
checkpoint works (test_TSOD.py:36-38).
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip
from .modules import (FinalPatchExpand_X4, FreqBlockv6, LayerNorm2d, Linear2d, MultiScaleDecoderBlock, PatchExpand,
                      VSSMEncoder, _infer, _init_weights, _run_blocks, model_mask_pool, _need_device, from_cl, load_pretrained_Base,
                      to_cl)
from .ops import CrossMerge_Line, CrossScan_Line


# Independent branches of the inference graph run on separate HIP streams; profiling passes that want every kernel timed
# alone set this False (a scheduling choice: the kernels are the same either way).
OVERLAP_BRANCHES = True
# r04: the same fork under autograd (the training step).  The autograd engine runs every backward node on the stream its forward ran
# on and synchronises where a gradient changes hands, so the guide branches' backward overlaps the encoder's deep stages as their
# forward does: 30.50 -> 29.99 ms per step as one hipGraph (scripts/ab_train.py, same box).  Bit-identical to the single-stream step
# (tests/test_gpu_model.py) -- once the weight-gradient GEMM waited for ALL of its transposed LDS reads: the first overlapped steps
# differed from run to run and went NaN one time in four, and the cause was a kernel, not a missing dependency (train_gemm.hip,
# wgrad_dma_kernel; scripts/dev/debug_overlap2.py and debug_wgrad_concurrent.py are the reproductions).
OVERLAP_TRAINING = True
_side_streams = {}


def _side_stream(device):
    key = str(device)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


class _StreamEdge(torch.autograd.Function):
    """Identity on an edge that crosses streams (the fork into / the join out of a guide branch): the backward marks the
    gradient travelling the other way as used by the stream that will read it.  The caching allocator hands a freed block
    back to the pool of the stream it was allocated on at once; a gradient allocated on one stream and consumed on the other
    must therefore carry a record_stream for the consumer, or the block is re-used under the consumer's still-queued kernel
    (r04, models.OVERLAP_TRAINING; the autograd engine does the same for the edges it routes itself)."""

    @staticmethod
    def forward(ctx, x, consumer_of_grad):
        ctx.consumer = consumer_of_grad
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if g is not None and g.is_cuda:
            # the stream this node runs on has just been handed g by its producer; the consumer must come after it
            here = torch.cuda.current_stream(g.device)
            if here != ctx.consumer:
                ev = torch.cuda.Event()
                ev.record(here)
                ctx.consumer.wait_event(ev)
            g.record_stream(ctx.consumer)
            if _DEBUG_STREAMS is not None:
                _DEBUG_STREAMS.append((here.cuda_stream, ctx.consumer.cuda_stream))
        return g, None


_DEBUG_STREAMS = None
_OVERLAP_TRAINING_STAGES = None
_OVERLAP_INFER_STAGES = None      # (the same for the inference forward: scripts/dev/ab_overlap_stages.py)


def _bias_scalar(conv: nn.Conv2d) -> float:
    """The single bias of a C -> 1 head as a Python float, cached per parameter version (a .item() per forward
    would synchronise the stream and cannot be captured into a graph)."""
    b = conv.bias
    if b is None:
        return 0.0
    key = (b._version, b.data_ptr())
    hit = conv.__dict__.get("_tramba_bias_scalar")
    if hit is None or hit[0] != key:
        hit = conv.__dict__["_tramba_bias_scalar"] = (key, float(b.detach().float().item()))
    return hit[1]


class VSSMDecoder(nn.Module):
    """Trambav6.py:13-139 and Trambav6_enc.py:27-159 (they differ only in the in-features of
    ``concat_back_dim``: 2*skip vs below//2 + skip, selected by ``concat_from_below``)."""

    def __init__(self, deep_supervision, features_per_stage=None, drop_path_rate=0.0, depths=None, img_size=384,
                 channel_first=True, concat_from_below=False, scan=CrossScan_Line, merge=CrossMerge_Line):
        super().__init__()
        assert channel_first
        chans = list(features_per_stage)
        self.deep_supervision = deep_supervision
        n_stages = len(chans)
        dpr = [x.item() for x in torch.linspace(drop_path_rate, 0, (n_stages - 1) * 2)]
        depths = [2, 2, 2, 2] if depths is None else depths
        res0 = img_size // 2 ** len(depths)
        self.channel_first = True
        self.stage_layers = nn.ModuleList()
        self.expand_layers = nn.ModuleList()
        self.guide_layers = nn.ModuleList()
        self.seg_layers = nn.ModuleList()
        self.concat_back_dim = nn.ModuleList()
        skip = chans[0]
        for stage in range(1, n_stages):
            below, skip = chans[-stage], chans[-(stage + 1)]
            self.expand_layers.append(PatchExpand(dim=below, dim_scale=2, norm_layer=LayerNorm2d, channel_first=True))
            r = res0 * (2 ** (stage - 1))
            self.guide_layers.append(FreqBlockv6(dim=skip, num_heads=4, input_resolution=(r, r), mlp_ratio=4.0,
                                                 drop_path=0.0, norm_layer=LayerNorm2d))
            blocks = [MultiScaleDecoderBlock(hidden_dim=skip, drop_path=dp, norm_layer=LayerNorm2d, channel_first=True,
                                             scan=scan, merge=merge)
                      for dp in dpr[sum(depths[:stage - 1]):sum(depths[:stage])]]
            self.stage_layers.append(nn.Sequential(OrderedDict(blocks=nn.Sequential(*blocks))))
            self.seg_layers.append(nn.Conv2d(skip, 1, 1, 1, 0, bias=True))
            cat_in = (below // 2 + skip) if concat_from_below else 2 * skip
            self.concat_back_dim.append(Linear2d(cat_in, skip))
        self.expand_layers.append(FinalPatchExpand_X4(dim=chans[0], dim_scale=4, norm_layer=LayerNorm2d, channel_first=True))
        self.stage_layers.append(nn.Identity())
        self.seg_layers.append(nn.Conv2d(skip, 1, 1, 1, 0, bias=True))
        self.apply(_init_weights)

    @staticmethod
    def _seg_cl(conv: nn.Conv2d, x):
        """1x1 conv C -> 1 on a channels-last map; returns NCHW logits (B,1,H,W) (fp32 from the HIP kernel)."""
        if _infer(x, conv.weight) and x.shape[-1] % 8 == 0:
            y = hip.rowdot_cl(x, hip._f32(conv.weight).view(-1), _bias_scalar(conv))
            return y.unsqueeze(1)
        from .modules import _RowDotCL
        return _RowDotCL.apply(x, conv.weight, conv.bias).unsqueeze(1)

    def _final_cl(self, x_low):
        """Last decoder stage: FinalPatchExpand_X4 -> (Identity) -> seg head.  Inference fuses pixel-shuffle +
        LayerNorm + the 1x1 head, so the (B, 4H, 4W, C) map is never materialised (Trambav6.py:132-137)."""
        fin, conv = self.expand_layers[-1], self.seg_layers[-1]
        if _infer(x_low, conv.weight) and isinstance(self.stage_layers[-1], nn.Identity) and fin.output_dim % 8 == 0:
            if (x_low.dtype != torch.float32 and fin.output_dim == 128 and x_low.shape[-1] % 64 == 0
                    and fin.expand.bias is None):
                y = hip.expand_norm_head_cl(x_low, fin.expand.weight.to(x_low.dtype), hip._f32(fin.norm.weight),
                                            hip._f32(fin.norm.bias), hip._f32(conv.weight).view(-1), _bias_scalar(conv),
                                            fin.scale, fin.norm.eps)
                return y.unsqueeze(1)
            xe = fin.expand._forward_cl(x_low)
            y = hip.shuffle_norm_head_cl(xe, hip._f32(fin.norm.weight), hip._f32(fin.norm.bias),
                                         hip._f32(conv.weight).view(-1), _bias_scalar(conv), fin.scale, fin.norm.eps)
            return y.unsqueeze(1)
        if isinstance(self.stage_layers[-1], nn.Identity) and x_low.is_cuda and fin.output_dim % 8 == 0:
            # training: LayerNorm over the C channels of an output pixel = over one contiguous C-group of the expanded row,
            # and the C -> 1 head is a dot product per group: both run on the UN-shuffled (M * P * P, C) view, and only the
            # scalar logits are rearranged -- the (B, 4H, 4W, C) map (302 MB at batch 8) is never permuted, forward or
            # backward.  'b (p1 p2 c) h w -> b c (h p1) (w p2)' (modules.py:246-250): group g = p1 * P + p2.
            from .modules import _RowDotCL, _ShuffleNormHeadCL
            b, h, w, _ = x_low.shape
            p, c = fin.scale, fin.output_dim
            xe = fin.expand._forward_cl(x_low)                               # (B, H, W, P*P*C)
            if hip.shuffle_norm_head_ok(xe, c) and conv.weight.numel() == c:
                # norm + head as ONE op each way: the normalised map is not stored, its gradient never formed
                lg = _ShuffleNormHeadCL.apply(xe, fin.norm.weight, fin.norm.bias, conv.weight, p, fin.norm.eps)
                if conv.bias is not None:
                    lg = lg + conv.bias.float().view(1, 1, 1)
                return lg.unsqueeze(1)
            yn = fin.norm._forward_cl(xe.view(b, h, w * p * p, c))           # LayerNorm per C-group
            lg = _RowDotCL.apply(yn, conv.weight, conv.bias)                 # (B, H, W*P*P) f32
            lg = lg.view(b, h, w, p, p).permute(0, 1, 3, 2, 4).reshape(b, h * p, w * p)
            return lg.unsqueeze(1)
        return self._seg_cl(conv, fin._forward_cl(x_low))

    def _forward_cl(self, skips_cl, guides=None):
        """skips_cl: [image, s1..sn] with s* channels-last.  Trambav6.py:114-139.
        guides: optional {decoder stage: (tensor, event)} of guide-branch outputs already computed on a side
        stream (BaseUMamba._forward_overlapped); the event is waited for right before the first use."""
        x_low = skips_cl[-1]
        outs = []
        n = len(self.stage_layers)
        for s in range(n):
            if s == n - 1:
                outs.append(self._final_cl(x_low))
                break
            x = self.expand_layers[s]._forward_cl(x_low)
            if s < n - 1:
                if guides is not None and s in guides:
                    mid, ready = guides[s]
                    torch.cuda.current_stream().wait_event(ready)
                    mid.record_stream(torch.cuda.current_stream())
                    if torch.is_grad_enabled() and mid.requires_grad:   # its gradient is formed here and read on the side stream
                        mid = _StreamEdge.apply(mid, _side_stream(mid.device))
                else:
                    mid = self.guide_layers[s]._forward_cl(skips_cl[-(s + 2)])
                x = self.concat_back_dim[s]._forward_cat_cl(x, mid)
                x = _run_blocks(self.stage_layers[s].blocks, x)
            if self.deep_supervision or s == n - 1:
                outs.append(self._seg_cl(self.seg_layers[s], x))
            x_low = x
        return outs if self.deep_supervision else outs[0]

    def forward(self, skips):
        return self._forward_cl([skips[0]] + [to_cl(s) for s in skips[1:]])


class BaseUMamba(nn.Module):
    """Trambav6.py:142-165 (Tramba-V)."""

    def __init__(self, vss_args, decoder_args, use_pretrain=True, pretrained_path=""):
        super().__init__()
        self.vssm_encoder = VSSMEncoder(**vss_args)
        self.decoder = VSSMDecoder(**decoder_args)
        self.compute_dtype = None
        if use_pretrain:
            load_pretrained_Base(self.vssm_encoder, ckpt_path=pretrained_path)

    def _forward_overlapped(self, x):
        """The decoder's guide branch (FreqBlockv6 on an encoder skip) depends on nothing but
        that skip: it is issued on a side stream the moment the encoder stage finishes, so its large 96x96 /
        48x48 kernels fill the CUs that the encoder's small 24x24 / 12x12 launches leave idle.  Captured into a
        hipGraph the fork/join become graph edges."""
        main = torch.cuda.current_stream()
        side = _side_stream(x.device)
        dec = self.decoder
        n = len(dec.stage_layers)
        guides = {}

        def on_stage(i, feat):          # encoder stage i -> skips[i + 1] -> decoder stage n - 2 - i
            s = n - 2 - i
            if s < 0 or s >= len(dec.guide_layers):
                return
            if (torch.is_grad_enabled() and _OVERLAP_TRAINING_STAGES is not None and s not in _OVERLAP_TRAINING_STAGES) or (
                    not torch.is_grad_enabled() and _OVERLAP_INFER_STAGES is not None and s not in _OVERLAP_INFER_STAGES):
                return                  # (debugging aid of scripts/dev/debug_overlap2.py: only these guide branches fork)
            side.wait_stream(main)
            feat.record_stream(side)
            with torch.cuda.stream(side):
                # (under autograd: the gradient of `feat` leaves the branch on the side stream and is summed on the main one)
                fin = _StreamEdge.apply(feat, main) if torch.is_grad_enabled() and feat.requires_grad else feat
                mid = dec.guide_layers[s]._forward_cl(fin)
                ready = torch.cuda.Event()
                ready.record(side)
            guides[s] = (mid, ready)

        skips = self.vssm_encoder._forward_cl(x, on_stage=on_stage)
        out = dec._forward_cl(skips, guides=guides)
        main.wait_stream(side)          # join (a no-op in time: every guide was already waited for)
        return out

    def forward(self, x):
        _need_device(x)
        if self.training:
            model_mask_pool(self).begin_step()        # one stochastic-depth draw per forward (modules._MaskPool)
        if self.compute_dtype is not None:
            x = x.to(self.compute_dtype)
        if OVERLAP_BRANCHES and (OVERLAP_TRAINING or not torch.is_grad_enabled()) and type(self.decoder) is VSSMDecoder:
            out = self._forward_overlapped(x)
        else:
            skips = self.vssm_encoder._forward_cl(x)
            out = self.decoder._forward_cl(skips)
        if self.compute_dtype is not None:
            out = [o.float() for o in out] if isinstance(out, list) else out.float()
        return out

    @torch.no_grad()
    def freeze_encoder(self):
        for name, p in self.vssm_encoder.named_parameters():
            if "patch_embed" not in name:
                p.requires_grad = False

    @torch.no_grad()
    def unfreeze_encoder(self):
        for p in self.vssm_encoder.parameters():
            p.requires_grad = True


def bulid_model(deep_supervision=True, use_pretrain=True, img_size=384, dims=128, depths=[2, 2, 2, 2],
                pretrained_path=""):
    """Trambav6.py:168-200 (the reference's spelling is kept: callers import ``bulid_model``)."""
    vss_args = dict(patch_size=4, in_chans=3, depths=[2, 2, 15, 2], dims=dims, drop_path_rate=0.6, patch_norm=True,
                    norm_layer="LN2D", posembed=False, imgsize=img_size)
    decoder_args = dict(deep_supervision=deep_supervision, features_per_stage=[dims, dims * 2, dims * 4, dims * 8],
                        depths=depths, img_size=img_size, drop_path_rate=0.2)
    return BaseUMamba(vss_args, decoder_args, use_pretrain=use_pretrain, pretrained_path=pretrained_path)


# ----------------------------------------------------------------------------- Tramba-R (ResNet50 encoder)
class Bottleneck(nn.Module):
    """resnet_encoder.py:62-79.  Stock conv/BN: runs on MIOpen (out of kernel scope)."""

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=(3 * dilation - 1) // 2,
                               bias=False, dilation=dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)), inplace=True)
        out = F.relu(self.bn2(self.conv2(out)), inplace=True)
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            x = self.downsample(x)
        return F.relu(out + x, inplace=True)


class ResNet(nn.Module):
    """resnet_encoder.py:81-110 (ResNet50 trunk; returns out5..out1 like the reference).  The
    reference loads a hard-coded checkpoint path in __init__ (:113); here weights come from
    ``load_state_dict`` / ``pretrained_path`` instead."""

    def __init__(self, cfg=None, pretrained_path=None):
        super().__init__()
        self.cfg = cfg
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self.make_layer(64, 3, stride=1, dilation=1)
        self.layer2 = self.make_layer(128, 4, stride=2, dilation=1)
        self.layer3 = self.make_layer(256, 6, stride=2, dilation=1)
        self.layer4 = self.make_layer(512, 3, stride=2, dilation=1)
        if pretrained_path:
            self.load_state_dict(torch.load(pretrained_path, map_location="cpu"), strict=False)

    def make_layer(self, planes, blocks, stride, dilation):
        downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                   nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, dilation=dilation)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, dilation=dilation))
        return nn.Sequential(*layers)

    def forward(self, x):
        out1 = F.relu(self.bn1(self.conv1(x)), inplace=True)
        out1 = F.max_pool2d(out1, kernel_size=3, stride=2, padding=1)
        out2 = self.layer1(out1)
        out3 = self.layer2(out2)
        out4 = self.layer3(out3)
        out5 = self.layer4(out4)
        return out5, out4, out3, out2, out1


class BaseUMambaEnc(nn.Module):
    """Trambav6_enc.py:162-231: Tramba-R (ResNet50), Tramba-S (Swin-B, window 12, 384) and Tramba-P (PVTv2-b4) in
    front of the same decoder.  The reference reads the encoders' ImageNet checkpoints from hard-coded paths at
    construction; here `pretrained_path` is optional (a checkpoint of the whole model is loaded by the caller,
    test_TSOD.py:38)."""

    def __init__(self, enc_type, decoder_args, pretrained_path=None):
        super().__init__()
        self.enc_type = enc_type
        self.compute_dtype = None
        kind = enc_type.split("-")[1] if enc_type.count("-") == 2 else ""
        if kind == "R" and enc_type.startswith("Tramba-"):
            self.encoder = ResNet(pretrained_path=pretrained_path)
            feats, depths = [256, 512, 1024], [2, 2, 2]
        elif kind == "S" and enc_type.startswith("Tramba-"):
            from .encoders import SwinTransformer
            self.encoder = SwinTransformer(img_size=384, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32),
                                           window_size=12)
            _load_encoder(self.encoder, pretrained_path, key="model")
            feats, depths = [128, 256, 512, 1024], [2, 2, 2, 2]
        elif kind == "P" and enc_type.startswith("Tramba-"):
            from .encoders import pvt_v2_b4
            self.encoder = pvt_v2_b4()
            _load_encoder(self.encoder, pretrained_path)
            feats, depths = [64, 128, 320, 512], [2, 2, 2, 2]
        else:
            raise ValueError(f"Unsupported encoder type: {enc_type}")
        self.kind = kind
        self.decoder = VSSMDecoder(**dict(decoder_args, features_per_stage=feats, depths=depths, concat_from_below=True))

    def forward(self, x):
        _need_device(x)
        if self.training:
            model_mask_pool(self).begin_step()
        if self.compute_dtype is not None:
            x = x.to(self.compute_dtype)
        if self.kind == "R":
            outs = self.encoder(x.contiguous(memory_format=torch.channels_last))
            feats = [to_cl(o) for o in outs[1:-1][::-1]]              # Trambav6_enc.py:212-213
        elif self.kind == "S":
            feats = self.encoder.features_cl(x, last=False)          # :210-211 (the last stage's output is never read)
        else:
            feats = self.encoder.features_cl(x)                      # :214-215
        out = self.decoder._forward_cl([x] + feats)
        if self.compute_dtype is not None:
            out = [o.float() for o in out]
        return out

    @torch.no_grad()
    def freeze_encoder(self):
        for name, p in self.encoder.named_parameters():
            if "patch_embed" not in name:
                p.requires_grad = False

    @torch.no_grad()
    def unfreeze_encoder(self):
        for p in self.encoder.parameters():
            p.requires_grad = True


def _load_encoder(encoder, path, key=None):
    """Trambav6_enc.py:177-179, 188-190: keep the checkpoint entries the encoder has, load them strictly by name."""
    if not path:
        return
    import os
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    sd = torch.load(path, map_location="cpu")
    sd = sd[key] if key and key in sd else sd
    own = encoder.state_dict()
    own.update({k: v for k, v in sd.items() if k in own})
    encoder.load_state_dict(own)


def bulid_model_enc(enc_type, deep_supervision=True, img_size=384, pretrained_path=None):
    """Trambav6_enc.py:233-248."""
    decoder_args = dict(deep_supervision=deep_supervision, features_per_stage=None, depths=None, img_size=img_size,
                        drop_path_rate=0.2)
    return BaseUMambaEnc(enc_type, decoder_args, pretrained_path=pretrained_path)


def build(model_name, args):
    """get_model.py:2-31."""
    if model_name in ("Tramba-V-TSOD", "Tramba-V-SOD"):
        path = getattr(args, "pretrained_path", "") or ""
        return bulid_model(deep_supervision=True, use_pretrain=bool(path), img_size=args.img_size, dims=128,
                           depths=[2, 2, 2, 2], pretrained_path=path)
    if model_name in ("Tramba-S-TSOD", "Tramba-P-TSOD", "Tramba-R-TSOD", "Tramba-S-SOD", "Tramba-P-SOD", "Tramba-R-SOD"):
        return bulid_model_enc(enc_type=model_name, deep_supervision=True, img_size=args.img_size)
    if model_name == "BaseUMamba-SOD":
        raise NotImplementedError("BaseUMamba-SOD is the reference's ablation baseline (out of scope, SURVEY 2 #10)")
    return None


# ----------------------------------------------------------------------------- precision policy
@torch.no_grad()
def prepare_inference(model: nn.Module, dtype=torch.bfloat16):
    """Put the model in eval mode with `dtype` activations.

    Policy (the reference is fp32-only; this is the MI355X bf16/fp16 policy):
    GEMM / dense-conv weights are stored in `dtype`; everything numerically delicate stays fp32 --
    LayerNorm affine, depth-wise stencil taps, dt_proj, dt bias, A_logs, Ds, DCT tables, all biases,
    and every accumulation / scan state inside the kernels.
    """
    model.eval()
    if dtype == torch.float32:
        model.compute_dtype = None
        return model
    for m in model.modules():
        if isinstance(m, Linear2d):
            m.weight.data = m.weight.data.to(dtype)
        elif isinstance(m, nn.Conv2d) and m.groups == 1:
            m.weight.data = m.weight.data.to(dtype)
            if m.bias is not None:
                m.bias.data = m.bias.data.to(dtype)
        elif isinstance(m, nn.BatchNorm2d):
            m.to(dtype)
    from .modules import SS2D
    for m in model.modules():
        if isinstance(m, SS2D):
            m.x_proj_weight.data = m.x_proj_weight.data.to(dtype)
    model.compute_dtype = dtype
    return model
