#!/usr/bin/env python3
"""Same-box A/B of the training step (Tramba-V 384x384, batch 8, bf16, stochastic depth on; BASELINE configs[2]) between
TRAMBA_TUNE_GEMM_TILE settings: every variant is its own model + GraphedTrainStep captured with that knob value, and the graphs
are replayed alternately in ONE process.  usage: python scripts/ab_train.py [tune values, default: 18 0]
A value >= 200 means: tune value - 200 with the guide branches on a side stream under autograd (models.OVERLAP_TRAINING, the default
since the end of round 4); a value < 200 captures the single-stream step.  (scripts/ab_knobs.py is the general form of this script.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta
from tramba_amd import hip, models, train

vals = [int(v) for v in sys.argv[1:]] or [18, 0]
b = 8
x = torch.randn(b, 3, 384, 384, generator=torch.Generator().manual_seed(0)).cuda()
y = (torch.rand(b, 1, 384, 384, generator=torch.Generator().manual_seed(1)) > 0.7).float().cuda()
steps = {}
for v in vals:
    torch.manual_seed(1026)
    m = ta.bulid_model(use_pretrain=False, img_size=384).cuda().train()
    m.compute_dtype = torch.bfloat16
    st = ta.GraphedTrainStep(m, train.get_opt(1e-4, m, capturable=True))
    hip.tune_set(hip.TUNE_GEMM_TILE, v % 200)
    models.OVERLAP_TRAINING = v >= 200
    try:
        for _ in range(3):
            loss = st(x, y)
    finally:
        hip.tune_set(hip.TUNE_GEMM_TILE, 0)
        models.OVERLAP_TRAINING = True
    torch.cuda.synchronize()
    steps[v] = st
    print(f"tune {v}: captured, loss {float(loss):.4f}", flush=True)
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = {v: [] for v in vals}
for _ in range(5):
    for v in vals:
        st = steps[v]
        st(x, y)
        torch.cuda.synchronize()
        a.record()
        for _ in range(10):
            st(x, y)
        e.record()
        torch.cuda.synchronize()
        tot[v].append(a.elapsed_time(e) / 10)
for v in vals:
    t = sorted(tot[v])
    print(f"train step, GEMM tune {v}: median {t[len(t) // 2]:.3f} ms  min {t[0]:.3f}  max {t[-1]:.3f}  ({b * 1000.0 / t[len(t) // 2]:.1f} img/s)")
