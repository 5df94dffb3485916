#!/usr/bin/env python3
"""LayerNorm backward (tramba_layernorm_bwd_res_cl + its partial-sum reduction) on the row counts / widths of a batch-8
training step, hipGraph-timed.  usage: bench_ln_bwd.py [lib path ...]  (other libraries run in child processes)"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SHAPES = [(73728, 128), (73728, 256), (18432, 256), (18432, 512), (4608, 512), (4608, 1024), (1152, 1024), (1152, 2048)]


def child(path):
    sys.path.insert(0, ROOT)
    import torch
    from tramba_amd import hip
    if path != "product":
        hip.LIB_PATH = path
        hip.SIGNATURES.pop("tramba_device_error", None)     # (older libraries: ABI 5)
    dev = torch.device("cuda")
    tot = 0.0
    for rows, c in SHAPES:
        x = torch.randn(rows, c, device=dev).bfloat16()
        dy = torch.randn(rows, c, device=dev).bfloat16()
        gres = torch.randn(rows, c, device=dev).bfloat16()
        w = torch.ones(c, device=dev)
        fn = lambda: hip.layernorm_bwd_res_cl(x, dy, w, 1e-5, gres=gres)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                keep = fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 60 * 1e3
        tot += us
        print(f"  rows={rows:6d} C={c:5d}: {us:6.1f} us  parts={hip.lib().tramba_layernorm_bwd_parts(rows, c, hip.dt(x))}", flush=True)
    print(f"  sum {tot:.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for path in ["product"] + sys.argv[1:]:
            print(path, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], check=False)
