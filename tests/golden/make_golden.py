#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING the Python reference.

Run in the build container only (the reference tree never travels to the GPU box):

    python tests/golden/make_golden.py            # TRAMBA_REFERENCE=/root/reference

What is imported: the reference's own modules (Models/*, Trambav6, Trambav6_enc,
utils/loss, Evaluation/metrics) so every golden array is an OUTPUT OF THE REFERENCE CODE.
What is stubbed, because the image lacks it (none of it is arithmetic on the path):
  * ``timm.models.layers`` {DropPath, trunc_normal_, to_2tuple}, ``timm.models.registry``,
    ``timm.models.vision_transformer`` (decorators / init helpers only);
  * ``dataset`` (imported but unused by resnet_encoder.py:8);
  * ``Tensor.cuda`` / ``Module.cuda`` become identities (tables are built with .cuda() at
    import time, csms6s.py:58-62,107-111,157-158); missing checkpoint files load as {}.
  * ``selective_scan_cuda_oflex``: the third-party CUDA extension is ABSENT from the
    reference tree.  A plain sequential torch-fp64 recurrence defined HERE stands in for it
    so the graph around it can run.  Consequently the fixtures pin everything EXCEPT the
    scan arithmetic itself (for which parity stays "unpinned", see oracle/__init__.py).

Weights follow tests/golden/synth.py (closed-form by parameter name), inputs likewise, so
the fixtures hold only small outputs.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402

REF = os.environ.get("TRAMBA_REFERENCE", "/root/reference")


# --------------------------------------------------------------------------- shims
def _standin_scan(u, delta, A, B, C, D, delta_bias, delta_softplus):
    """Sequential recurrence in fp64 (stand-in for the absent CUDA extension)."""
    nb, kd, L = u.shape
    K, N = B.shape[1], B.shape[2]
    rep = kd // K
    dt = delta.double()
    if delta_bias is not None:
        dt = dt + delta_bias.double()[None, :, None]
    if delta_softplus:
        dt = torch.nn.functional.softplus(dt, threshold=20)
    Bx = B.double().repeat_interleave(rep, 1)
    Cx = C.double().repeat_interleave(rep, 1)
    ud = u.double()
    Ad = A.double()
    h = torch.zeros(nb, kd, N, dtype=torch.float64)
    ys = []
    for l in range(L):
        h = torch.exp(dt[:, :, l, None] * Ad[None]) * h + dt[:, :, l, None] * Bx[:, :, :, l] * ud[:, :, l, None]
        ys.append((Cx[:, :, :, l] * h).sum(-1))
    y = torch.stack(ys, -1)
    if D is not None:
        y = y + D.double()[None, :, None] * ud
    return y


def install_shims():
    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * mask / keep

    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")
    layers.DropPath = DropPath
    layers.trunc_normal_ = lambda t, mean=0.0, std=1.0, a=-2.0, b=2.0: nn.init.trunc_normal_(t, mean, std, a, b)
    layers.to_2tuple = lambda v: v if isinstance(v, tuple) else (v, v)
    registry = types.ModuleType("timm.models.registry")
    registry.register_model = lambda f: f
    vit = types.ModuleType("timm.models.vision_transformer")
    vit._cfg = lambda **kw: dict(kw)
    for name, mod in (("timm", timm), ("timm.models", models), ("timm.models.layers", layers),
                      ("timm.models.registry", registry), ("timm.models.vision_transformer", vit),
                      ("dataset", types.ModuleType("dataset"))):
        sys.modules[name] = mod

    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self

    _real_load = torch.load

    def _load(path, *a, **k):
        if isinstance(path, str) and not os.path.exists(path):
            return {}
        return _real_load(path, *a, **k)

    torch.load = _load
    import torch.utils.model_zoo as model_zoo
    model_zoo.load_url = lambda *a, **k: {}

    ext = types.ModuleType("selective_scan_cuda_oflex")

    def fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows, oflex):
        y = _standin_scan(u, delta, A, B, C, D, delta_bias, delta_softplus)
        return y.float() if oflex else y.to(u.dtype), torch.zeros(1)

    def bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows):
        with torch.enable_grad():
            ins = [t.detach().double().requires_grad_() for t in (u, delta, A, B, C, D, delta_bias)]
            y = _standin_scan(*ins, delta_softplus)
            grads = torch.autograd.grad(y, ins, dout.double())
        return [g.to(t.dtype) for g, t in zip(grads, (u, delta, A, B, C, D, delta_bias))]

    ext.fwd, ext.bwd = fwd, bwd
    sys.modules["selective_scan_cuda_oflex"] = ext
    sys.path.insert(0, REF)


def sha16(arr):
    return hashlib.sha256(np.ascontiguousarray(arr, dtype="<i8").tobytes()).hexdigest()[:16]


def manifest_of(model):
    return [(k, list(v.shape)) for k, v in model.state_dict().items()]


def load_synth(model):
    sd = model.state_dict()
    new = synth.synth_state_dict(((k, v.shape) for k, v in sd.items()), keep=synth.DCT_KEYS)
    for k in sd:
        if k not in new:
            new[k] = sd[k]
    model.load_state_dict(new, strict=True)
    return model


def pooled(t, k):
    return torch.nn.functional.avg_pool2d(t, k).numpy()


# --------------------------------------------------------------------------- main
def main():
    install_shims()
    torch.manual_seed(0)
    torch.set_grad_enabled(True)
    import Models.SS2D.csms6s as cs
    from Models.SS2D import SpiralLine, Window, Dilation
    from Models.DCT_2D import DCT2D
    from Models.vmamba import SS2D, VSSBlock, MultiScaleDecoderBlock
    from Models.freq_mamba import FreqBlockv6
    from Models.modules import PatchExpand, FinalPatchExpand_X4, FreqExpand2D, LayerNorm2d
    from utils.loss import iou_loss
    import Evaluation.metrics as M

    out = {}
    meta = {"reference": "mj129/Tramba", "generator": "tests/golden/make_golden.py",
            "torch": torch.__version__, "numpy": np.__version__}

    # ---- G1: index tables (flat gather index per direction) ----
    g1 = {}
    for h in (12, 24, 48, 96):
        line = cs.spiral_line_index[str(h)]
        g1[f"line_{h}"] = [sha16((t[:, 0] + t[:, 1] * h).numpy()) for t in line]
        dil = cs.dilation_index[str(h)]
        g1[f"dilation_{h}"] = [sha16((t[:, 0] * h + t[:, 1]).numpy()) for t in dil]
        win = cs.window_index[str(h)]
        g1[f"window_{h}"] = [sha16((t[:, 0] * h + t[:, 1]).numpy()) for t in win]
    for h in (7, 14, 28, 56):
        line = cs.spiral_line_index[str(h)]
        g1[f"line_{h}"] = [sha16((t[:, 0] + t[:, 1] * h).numpy()) for t in line]
    # off-table sizes straight from the reference *generators* (size-generic code)
    for h in (16, 32, 64, 192):
        g1[f"line_{h}"] = [sha16((t[:, 0] + t[:, 1] * h).numpy()) for t in SpiralLine.generate_indices(h, h)]
        g1[f"dilation_{h}"] = [sha16((t[:, 0] * h + t[:, 1]).numpy())
                               for t in Dilation.generate_dilation_indices(h, h, dilation_rate=4)]
    for h, ws in ((16, 4), (16, 8), (32, 8), (64, 16), (192, 16)):
        g1[f"window_{h}_ws{ws}"] = [sha16((t[:, 0] * h + t[:, 1]).numpy())
                                    for t in Window.generate_window_indices(h, h, window_size=ws)]
    line12 = cs.spiral_line_index["12"]
    g1["line_12_dir0_flat"] = (line12[0][:, 0] + line12[0][:, 1] * 12).tolist()
    mult = np.zeros(144, dtype=np.int64)
    for t in line12:
        np.add.at(mult, (t[:, 0] + t[:, 1] * 12).numpy(), 1)
    g1["line_12_multiplicity"] = mult.tolist()
    meta["G1"] = g1

    # ---- G2: scan / merge Functions on seeded data, 12x12 ----
    x = synth.synth_input("g2_x", (2, 3, 12, 12))
    out["g2_x"] = x.numpy()
    for tag, scan, merge, k in (("raster", cs.CrossScan, cs.CrossMerge, 4),
                                ("helix", cs.CrossScan_Line, cs.CrossMerge_Line, 8),
                                ("window", cs.CrossScan_Window, cs.CrossMerge_Window, 4),
                                ("dilation", cs.CrossScan_Dilation, cs.CrossMerge_Dilation, 4)):
        xs = scan.apply(x)
        out[f"g2_scan_{tag}"] = xs.numpy()
        ys = synth.synth_input("g2_y_" + tag, (2, k, 3, 12, 12))
        out[f"g2_merge_{tag}"] = merge.apply(ys).numpy()
        # backward of scan == merge, backward of merge == scan (adjoints)
        xr = x.clone().requires_grad_()
        scan.apply(xr).backward(ys.view(2, k, 3, 144))
        out[f"g2_scan_bwd_{tag}"] = xr.grad.numpy()

    # ---- G3: DCT2D ----
    dct = DCT2D(12, 12)
    xd = (torch.arange(144, dtype=torch.float32) / 144).view(1, 1, 12, 12)
    high, low = dct(xd)
    out["g3_arange_low"], out["g3_arange_high"] = low.numpy(), high.numpy()
    xd2 = synth.synth_input("g3_x", (2, 3, 24, 24))
    high, low = DCT2D(24, 24)(xd2)
    out["g3_low_24"], out["g3_high_24"] = low.numpy(), high.numpy()
    out["g3_weight_12"] = dct.dct_x.weight.numpy()
    assert torch.equal(dct.dct_x.weight, dct.dct_y.weight)

    # ---- G4: blocks with synthetic weights, forward + input gradient ----
    def run_block(tag, block, shape):
        block = load_synth(block).eval()
        meta.setdefault("G4_manifest", {})[tag] = manifest_of(block)
        xb = synth.synth_input("g4_" + tag, shape).requires_grad_()
        yb = block(xb)
        gy = synth.synth_input("g4_gy_" + tag, tuple(yb.shape))
        grads = torch.autograd.grad(yb, [xb] + list(block.parameters()), gy)
        out[f"g4_{tag}_y"] = yb.detach().numpy()
        out[f"g4_{tag}_dx"] = grads[0].numpy()
        # a few parameter-gradient digests (sum and abs-sum) keep the fixture small
        meta.setdefault("G4_param_grads", {})[tag] = {
            n: [float(g.double().sum()), float(g.double().abs().sum())]
            for (n, _), g in zip(block.named_parameters(), grads[1:])}

    run_block("ss2d_raster", SS2D(d_model=16, d_state=1, ssm_ratio=2.0, dt_rank="auto", d_conv=3,
                                  conv_bias=False, channel_first=True), (2, 16, 12, 12))
    run_block("vssblock", VSSBlock(hidden_dim=16, drop_path=0.0, norm_layer=LayerNorm2d,
                                   channel_first=True), (2, 16, 12, 12))
    run_block("freqblock", FreqBlockv6(dim=16, input_resolution=(12, 12)), (2, 16, 12, 12))
    run_block("helixblock", MultiScaleDecoderBlock(hidden_dim=16, drop_path=0.0, norm_layer=LayerNorm2d,
                                                   channel_first=True), (2, 16, 12, 12))
    run_block("freqblock24", FreqBlockv6(dim=32, input_resolution=(24, 24)), (1, 32, 24, 24))
    run_block("helixblock24", MultiScaleDecoderBlock(hidden_dim=32, drop_path=0.0, norm_layer=LayerNorm2d,
                                                     channel_first=True), (1, 32, 24, 24))
    run_block("patchexpand", PatchExpand(dim=32, dim_scale=2), (2, 32, 6, 6))
    run_block("finalexpand", FinalPatchExpand_X4(dim=8, dim_scale=4), (2, 8, 6, 6))
    run_block("freqexpand", FreqExpand2D(dim=8), (2, 8, 6, 6))

    # ---- G7: loss / metric known answers ----
    pred = synth.synth_input("g7_pred", (2, 1, 24, 24), scale=2.0)
    mask = (synth.synth_input("g7_mask", (2, 1, 24, 24)) > 0.3).float()
    meta["G7"] = {
        "iou_loss": float(iou_loss(pred, mask)),
        "bce": float(torch.nn.functional.binary_cross_entropy_with_logits(pred, mask)),
    }
    mae = M.MAE()
    for i in range(2):
        mae.step(pred=torch.sigmoid(pred[i, 0]).numpy(), gt=mask[i, 0].numpy())
    meta["G7"]["mae"] = float(mae.get_results()["mae"])

    # ---- G5/G6: full models ----
    if os.environ.get("TRAMBA_GOLDEN_SKIP_FULL") != "1":
        torch.set_grad_enabled(False)
        import Trambav6
        import Trambav6_enc
        mv = Trambav6.bulid_model(deep_supervision=True, use_pretrain=False, img_size=384, dims=128,
                                  depths=[2, 2, 2, 2])
        meta["G6_tramba_v"] = manifest_of(mv)
        meta["G6_tramba_v_params"] = int(sum(p.numel() for p in mv.parameters()))
        load_synth(mv).eval()
        xi = synth.synth_input("g5_v", (1, 3, 384, 384))
        feats = mv.vssm_encoder(xi)
        for i, f in enumerate(feats[1:]):
            out[f"g5_v_enc{i}_pool"] = pooled(f, f.shape[-1] // 6)
        outs = mv(xi)
        for i, o in enumerate(outs[:3]):
            out[f"g5_v_out{i}"] = o.numpy()
        out["g5_v_out3_pool8"] = pooled(outs[3], 8)
        out["g5_v_out3_crop"] = outs[3][:, :, 160:224, 160:224].numpy()
        p = torch.sigmoid(outs[3])[0, 0].numpy()
        gt = (synth.synth_input("g5_gt", (384, 384)) > 0.5).numpy()
        mm = M.MAE()
        mm.step(pred=p, gt=gt)
        meta["G5_tramba_v_mae"] = float(mm.get_results()["mae"])
        del mv

        mr = Trambav6_enc.bulid_model(enc_type="Tramba-R-TSOD", deep_supervision=True, img_size=384)
        meta["G6_tramba_r"] = manifest_of(mr)
        meta["G6_tramba_r_params"] = int(sum(p.numel() for p in mr.parameters()))
        load_synth(mr).eval()
        xi = synth.synth_input("g5_r", (1, 3, 384, 384))
        outs = mr(xi)
        for i, o in enumerate(outs[:2]):
            out[f"g5_r_out{i}"] = o.numpy()
        out["g5_r_out2_pool8"] = pooled(outs[2], 8)
        out["g5_r_out2_crop"] = outs[2][:, :, 160:224, 160:224].numpy()

    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=0, sort_keys=True)
    print("wrote", len(out), "arrays;", os.path.getsize(os.path.join(HERE, "golden.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
