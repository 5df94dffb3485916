#!/usr/bin/env python3
"""Element-wise error of the bf16 / fp16 Tramba-V forward at the benchmarked configuration (384x384, batch 4) against the
fp32 CPU oracle on the same inputs and closed-form weights: prints, per output map, max-abs and RMS logit error, the RMS
of the logits themselves and the fraction of pixels whose sigmoid > 0.5 decision differs.  The numbers printed here are
what tests/test_gpu_model.py::test_tramba_v_batch4_low_precision_elementwise asserts."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import synth  # noqa: E402
from oracle import model as om  # noqa: E402


def inputs(batch=4):
    xs = [synth.synth_input("g5_v", (1, 3, 384, 384))]
    for i in range(1, batch):
        xs.append(synth.synth_input(f"g5_v_b{i}", (1, 3, 384, 384)))
    return torch.cat(xs, 0)


def stats(got, want):
    d = (got.double() - want.double())
    return {"max_abs": float(d.abs().max()), "rms": float(d.square().mean().sqrt()),
            "ref_rms": float(want.double().square().mean().sqrt()),
            "flips": float(((got > 0) != (want > 0)).double().mean())}


def main():
    import tramba_amd as ta
    dev = "cuda"

    def build():
        m = ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=384)
        sd = m.state_dict()
        new = synth.synth_state_dict(((k, v.shape) for k, v in sd.items()), keep=synth.CONST_KEYS)
        for k in sd:
            new.setdefault(k, sd[k])
        m.load_state_dict(new, strict=True)
        return m.to(dev).eval()

    m = build()
    x = inputs()
    with torch.no_grad():
        want = om.tramba_v({k: v.detach().cpu() for k, v in m.state_dict().items()}, x)
        res = {}
        got32 = [o.float().cpu() for o in m(x.to(dev))]
        res["fp32"] = [stats(g, w) for g, w in zip(got32, want)]
        for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            mm = ta.prepare_inference(build(), dt)   # (prepare_inference rounds the weights in place)
            got = [o.float().cpu() for o in mm(x.to(dev))]
            one = [o.float().cpu() for o in mm(x[2:3].to(dev))]
            res[name] = [stats(g, w) for g, w in zip(got, want)]
            res[name + "_b4_vs_b1"] = [stats(g[2:3], o) for g, o in zip(got, one)]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
