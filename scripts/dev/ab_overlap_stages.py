#!/usr/bin/env python3
"""which of the decoder's guide branches are worth forking onto the side stream in the captured inference forward (batch 4)?
variants = subsets of decoder stages (0 = the deepest guide, 2 = the 96x96 one), graphs replayed alternately"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tramba_amd as ta
from tramba_amd import models
torch.manual_seed(0)
m = ta.prepare_inference(ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=384).cuda().eval(), torch.bfloat16)
x = torch.randn(4, 3, 384, 384, device="cuda")
variants = {"all": None, "none": (), "0": (0,), "1": (1,), "2": (2,), "01": (0, 1), "12": (1, 2), "02": (0, 2)}
graphs = {}
for name, st in variants.items():
    models._OVERLAP_INFER_STAGES = st
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            m(x)
    graphs[name] = g
models._OVERLAP_INFER_STAGES = None
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = {k: [] for k in graphs}
for _ in range(6):
    for k, g in graphs.items():
        g.replay(); torch.cuda.synchronize()
        a.record()
        for _ in range(20):
            g.replay()
        e.record(); torch.cuda.synchronize()
        tot[k].append(a.elapsed_time(e) / 20)
for k, t in tot.items():
    t = sorted(t)
    print(f"guide branches on the side stream: {k:5s} median {t[len(t) // 2]:.4f} ms  min {t[0]:.4f}", flush=True)
