#!/usr/bin/env python3
"""Generate tests/golden/metrics_golden.json by IMPORTING the reference's Evaluation/metrics.py (numpy + scipy only,
nothing stubbed) and running its five metric classes on the closed-form cases of synth.metric_cases().

    python tests/golden/make_golden_metrics.py            # TRAMBA_REFERENCE=/root/reference

Run in the build container only; the fixture (numbers only) is what travels."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402

REF = os.environ.get("TRAMBA_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
import Evaluation.metrics as M  # noqa: E402


def main():
    out = {"cases": {}}
    agg = dict(FM=M.Fmeasure_and_FNR(), WFM=M.WeightedFmeasure(), SM=M.Smeasure(), EM=M.Emeasure(), MAE=M.MAE())
    for name, pred, gt in synth.metric_cases():
        one = dict(FM=M.Fmeasure_and_FNR(), WFM=M.WeightedFmeasure(), SM=M.Smeasure(), EM=M.Emeasure(), MAE=M.MAE())
        for grp in (one, agg):
            for m in grp.values():
                m.step(pred=pred.copy(), gt=gt.copy())
        out["cases"][name] = summarize(one)
        out["cases"][name]["shape"] = list(pred.shape)
    out["all"] = summarize(agg)
    with open(os.path.join(HERE, "metrics_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote metrics_golden.json:", {k: round(v["mae"], 6) for k, v in out["cases"].items()})


def summarize(ms):
    fm, fnr = ms["FM"].get_results()
    em = ms["EM"].get_results()["em"]
    return {
        "mae": float(ms["MAE"].get_results()["mae"]),
        "sm": float(ms["SM"].get_results()["sm"]),
        "wfm": float(ms["WFM"].get_results()["wfm"]),
        "fm_adp": float(fm["fm"]["adp"]),
        "fm_curve": [float(v) for v in fm["fm"]["curve"]],
        "precision": [float(v) for v in fm["pr"]["p"]],
        "recall": [float(v) for v in fm["pr"]["r"]],
        "fnr": float(fnr),
        "em_adp": float(em["adp"]),
        "em_curve": [float(v) for v in em["curve"]],
    }


if __name__ == "__main__":
    main()
