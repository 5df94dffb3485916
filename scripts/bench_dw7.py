#!/usr/bin/env python3
"""The folded 7x7 depth-wise stencil of DWMSMlp at the decoder's three shapes: r03 kernels (TRAMBA_TUNE_DW_FORM 1: one output row per
thread; weight gradient with the tap row outermost) against the marching kernels of r04 (a lane walks a band of rows; every input row
is loaded and converted once), each launch captured in a hipGraph of 10 and replayed, forms interleaved.
usage: python scripts/bench_dw7.py [rows ...]   (extra band heights to try for the marching kernels)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
rows_extra = [int(a) for a in sys.argv[1:]]


def graph_us(fn, reps=10, replays=5):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(reps):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(replays):
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


tot = {}
for b in (4, 8):
    for h, c in ((96, 512), (48, 1024), (24, 2048)):
        x = torch.randn(b, h, h, c, generator=g).bfloat16().to(dev)
        gy = torch.randn(b, h, h, c, generator=g).bfloat16().to(dev)
        wt = (torch.randn(49, c, generator=g) * 0.1).to(dev)
        bt = torch.randn(c, generator=g).to(dev)
        px = b * h * h * c / 2 / 64 * 49 * 4 / 1024 / 2.4e9 * 1e6     # packed-FMA floor, us (49 v_pk_fma_f32 per channel pair and pixel)
        line = f"B={b} {h}x{h} C={c:5d} fma-floor {px:5.1f} us |"
        launches = {"fwd": lambda: hip.dwconv_cl(x, wt, bt, 2), "fwd+pre": lambda: hip.dwconv_dual_cl(x, wt, bt, 2, True, False),
                    "dgrad": lambda: hip.dwconv_dual_cl(gy, wt, bt, 0, False, True), "wgrad": lambda: hip.dwconv_wgrad_cl(x, gy, 7)}
        for name, fn in launches.items():
            if b == 4 and name != "fwd":
                continue
            cells = []
            for form, rows in [(1, 0), (0, 0)] + [(0, r) for r in rows_extra]:
                hip.tune_set(hip.TUNE_DW_FORM, form)
                hip.tune_set(hip.TUNE_DW_ROWS, rows)
                try:
                    us = graph_us(fn)
                finally:
                    hip.tune_set(hip.TUNE_DW_FORM, 0)
                    hip.tune_set(hip.TUNE_DW_ROWS, 0)
                key = (b, name, "r03" if form else f"march{rows or ''}")
                tot[key] = tot.get(key, 0.0) + us
                cells.append(f"{'r03' if form else 'march' + (str(rows) if rows else '')} {us:6.1f}")
            line += f"  {name}: " + " ".join(cells)
        print(line, flush=True)
for k, v in sorted(tot.items()):
    print(k, round(v, 1))
