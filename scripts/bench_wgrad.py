#!/usr/bin/env python3
"""Weight-gradient TN GEMM (tramba_wgrad_cl) on the shapes of a batch-8 Tramba-V training step: time per call (TN kernel +
slab sum) and TFLOP/s, per staging form (TRAMBA_TUNE_GEMM_TILE: 0 = LDS-DMA, 3 stages, ~256 workgroups (the library's choice); 9 =
4 stages; 8 = register-staged, one tile in flight; 10 / 11 / 12 = 384 / 512 / 768 workgroups wanted).  usage: bench_wgrad.py [lib path ...]  (other libraries: child processes)"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SHAPES = [  # tokens, N (channels of gy), K (channels of x)
    (73728, 256, 128), (73728, 128, 256), (73728, 512, 128), (73728, 128, 512),
    (18432, 512, 256), (18432, 256, 512), (18432, 1024, 256), (18432, 256, 1024),
    (4608, 1024, 512), (4608, 512, 1024), (4608, 2048, 512), (4608, 512, 2048),
    (1152, 2048, 1024), (1152, 1024, 2048), (1152, 4096, 1024), (1152, 1024, 4096),
]


def child(path):
    sys.path.insert(0, ROOT)
    import torch
    from tramba_amd import hip
    if path != "product":
        hip.LIB_PATH = path
    dev = torch.device("cuda")
    tot = {0: 0.0, 9: 0.0, 8: 0.0, 10: 0.0, 11: 0.0, 12: 0.0}
    for (m, n, k) in SHAPES:
        gy = torch.randn(m, n, device=dev).bfloat16()
        x = torch.randn(m, k, device=dev).bfloat16()
        line, ref = f"  M={m:6d} N={n:5d} K={k:5d}:", None
        for form in (0, 9, 8, 10, 11, 12):
            hip.tune_set(hip.TUNE_GEMM_TILE, form)
            for _ in range(3):
                gw, gb = hip.wgrad_cl(gy, x, True)
            torch.cuda.synchronize()
            # 20 calls captured into one hipGraph: launched from Python a call costs ~25 us of host time, more than its kernels
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(20):
                    keep = hip.wgrad_cl(gy, x, True)
            graph.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 60 * 1e3
            del graph, keep
            tot[form] += us
            ref = (gw, gb) if ref is None else ref
            same = form >= 10 or (torch.equal(gw, ref[0]) and torch.equal(gb, ref[1]))   # (another split: another order)
            line += f"  f{form}: {us:5.1f} us {2.0 * m * n * k / us / 1e6:5.0f} TF{'' if same else ' DIFFERS'}"
        hip.tune_set(hip.TUNE_GEMM_TILE, 0)
        print(line, flush=True)
    print("  sums: " + "  ".join(f"form {f}: {t:.1f} us" for f, t in tot.items()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for path in ["product"] + sys.argv[1:]:
            print(path, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], check=False)
