#!/usr/bin/env python3
"""Same-box A/B of the whole model between settings of the library's tune knobs (include/tramba_hip.h): every variant is captured
as a hipGraph with its knobs set -- the batch-4 bf16 inference forward (BASELINE configs[1]) and the batch-8 training step
(configs[2], GraphedTrainStep) -- and the graphs are replayed alternately in ONE process (box-to-box spread is 3-5 %, more than most
single changes).  A variant is a comma-separated list of knob=value, knobs: gemm, dw, dwrows, wgrad, scan, merge; "base" = all 0.
usage: python scripts/ab_knobs.py [fwd|train|both] variant [variant ...]     e.g.  ab_knobs.py both dw=1 base"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta
from tramba_amd import hip, train

KNOBS = {"gemm": hip.TUNE_GEMM_TILE, "dw": hip.TUNE_DW_FORM, "dwrows": hip.TUNE_DW_ROWS, "wgrad": hip.TUNE_WGRAD_FORM,
         "scan": hip.TUNE_SCAN_FORM, "merge": hip.TUNE_MERGE_FORM}
legs = sys.argv[1] if len(sys.argv) > 1 else "both"
variants = sys.argv[2:] or ["base"]


class knobs:
    def __init__(self, spec):
        self.kv = [] if spec == "base" else [(KNOBS[k], int(v)) for k, v in (p.split("=") for p in spec.split(","))]

    def __enter__(self):
        for k, v in self.kv:
            hip.tune_set(k, v)

    def __exit__(self, *exc):
        for k, _ in self.kv:
            hip.tune_set(k, 0)


def alternate(graphs, reps, rounds=6):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = {v: [] for v in graphs}
    for _ in range(rounds):
        for v, run in graphs.items():
            run()
            torch.cuda.synchronize()
            a.record()
            for _ in range(reps):
                run()
            e.record()
            torch.cuda.synchronize()
            tot[v].append(a.elapsed_time(e) / reps)
    return {v: sorted(t) for v, t in tot.items()}


if legs in ("fwd", "both"):
    torch.manual_seed(0)
    m = ta.prepare_inference(ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=384).cuda().eval(), torch.bfloat16)
    x = torch.randn(4, 3, 384, 384, device="cuda")
    graphs = {}
    for v in variants:
        with knobs(v), torch.no_grad():
            for _ in range(3):
                m(x)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                m(x)
        graphs[v] = g.replay
    for v, t in alternate(graphs, 20).items():
        print(f"forward batch 4, {v:24s}: median {t[len(t) // 2]:.4f} ms  min {t[0]:.4f}  max {t[-1]:.4f}  ({4000.0 / t[len(t) // 2]:.1f} img/s)", flush=True)
    del graphs, m

if legs in ("train", "both"):
    b = 8
    x = torch.randn(b, 3, 384, 384, generator=torch.Generator().manual_seed(0)).cuda()
    y = (torch.rand(b, 1, 384, 384, generator=torch.Generator().manual_seed(1)) > 0.7).float().cuda()
    steps = {}
    for v in variants:
        torch.manual_seed(1026)
        m = ta.bulid_model(use_pretrain=False, img_size=384).cuda().train()
        m.compute_dtype = torch.bfloat16
        st = ta.GraphedTrainStep(m, train.get_opt(1e-4, m, capturable=True))
        with knobs(v):
            for _ in range(3):
                loss = st(x, y)
        torch.cuda.synchronize()
        steps[v] = (lambda s=st: s(x, y))
        print(f"train {v}: captured, loss {float(loss):.4f}", flush=True)
    for v, t in alternate(steps, 10, rounds=5).items():
        print(f"train step batch 8, {v:24s}: median {t[len(t) // 2]:.3f} ms  min {t[0]:.3f}  max {t[-1]:.3f}  ({b * 1000.0 / t[len(t) // 2]:.1f} img/s)", flush=True)
