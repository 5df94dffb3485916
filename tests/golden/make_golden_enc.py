#!/usr/bin/env python3
"""Generate tests/golden/golden_enc.npz + golden_enc_meta.json: Tramba-S (Swin-B) and Tramba-P (PVTv2-b4) of the
reference (Trambav6_enc.py) run on CPU with the closed-form synthetic weights of synth.py.

    python tests/golden/make_golden_enc.py            # TRAMBA_REFERENCE=/root/reference

Same shims as make_golden.py (imported from it: timm helpers, .cuda(), the stand-in sequential scan for the absent
selective_scan_cuda_oflex); in addition the encoders' ImageNet checkpoints, which the reference reads from hard-coded
paths at construction (Trambav6_enc.py:177, 188), load as empty dictionaries (and loading an empty dictionary is
made a no-op).  Fixture = state_dict manifests,
parameter counts, pooled encoder features, the three low-resolution outputs in full and a digest of the 384x384 one."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import synth  # noqa: E402


class _Empty(dict):
    def __missing__(self, key):
        return {}


def main():
    mg.install_shims()
    real_load = torch.load
    torch.load = lambda path, *a, **k: _Empty() if isinstance(path, str) and not os.path.exists(path) else real_load(path, *a, **k)
    real_lsd = torch.nn.Module.load_state_dict
    # the (now empty) ImageNet checkpoint is loaded strictly at construction: loading nothing is a no-op here
    torch.nn.Module.load_state_dict = lambda self, sd, *a, **k: None if len(sd) == 0 else real_lsd(self, sd, *a, **k)
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    import Trambav6_enc
    out, meta = {}, {}
    for tag, name in (("s", "Tramba-S-TSOD"), ("p", "Tramba-P-TSOD")):
        m = Trambav6_enc.bulid_model(enc_type=name, deep_supervision=True, img_size=384)
        meta[f"G6_tramba_{tag}"] = mg.manifest_of(m)
        meta[f"G6_tramba_{tag}_params"] = int(sum(p.numel() for p in m.parameters()))
        sd = m.state_dict()
        new = synth.synth_state_dict(((k, v.shape) for k, v in sd.items() if v.is_floating_point()
                                      and not k.endswith("attn_mask")), keep=synth.DCT_KEYS)
        m.load_state_dict({k: new.get(k, v) for k, v in sd.items()}, strict=True)
        m.eval()
        x = synth.synth_input(f"g5_{tag}", (1, 3, 384, 384))
        feats = m.encoder(x)
        for i, f in enumerate(feats):
            out[f"g5_{tag}_enc{i}_pool"] = mg.pooled(f, f.shape[-1] // 6)
        outs = m(x)
        for i, o in enumerate(outs[:3]):
            out[f"g5_{tag}_out{i}"] = o.numpy()
        out[f"g5_{tag}_out3_pool8"] = mg.pooled(outs[3], 8)
        out[f"g5_{tag}_out3_crop"] = outs[3][:, :, 160:224, 160:224].numpy()
        print(name, meta[f"G6_tramba_{tag}_params"], [tuple(o.shape) for o in outs], float(outs[3].abs().mean()), flush=True)
        del m
    np.savez_compressed(os.path.join(HERE, "golden_enc.npz"), **out)
    with open(os.path.join(HERE, "golden_enc_meta.json"), "w") as f:
        json.dump(meta, f)
    print("wrote", sum(v.nbytes for v in out.values()), "bytes of arrays")


if __name__ == "__main__":
    main()
