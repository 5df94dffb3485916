#!/usr/bin/env python3
"""Same-box A/B of the Tramba-V 384x384 batch-4 bf16 forward (BASELINE config 2) between variants that differ in the LIBRARY:
  * `r03`  tramba_amd/_lib_r03/libtramba_hip.so, the end-of-round-3 library built from its commit (scripts only; never shipped)
  * `r04`  the current library with TRAMBA_TUNE_GEMM_TILE 18 (the r03 GEMM kernels) -- isolates everything but the GEMMs
  * `new`  the current library, its own choices
Every variant is captured as a hipGraph in ONE process and the graphs are replayed alternately (box-to-box spread is 3-5 %).
usage: python scripts/ab_lib.py [rounds]   (profiles/r04_ab_lib.txt)"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tramba_amd as ta
from tramba_amd import hip


def load(path):
    l = ctypes.CDLL(path)
    for name, (res, args) in hip.SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError:
            continue
        fn.restype, fn.argtypes = res, args
    return l


def capture(m, x):
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            m(x)
    return g


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    new = hip.lib()
    old_path = os.path.join(ROOT, "tramba_amd", "_lib_r03", "libtramba_hip.so")
    libs = {"new": (new, 0), "r04": (new, 18)}
    if os.path.exists(old_path):
        libs["r03"] = (load(old_path), 0)
    torch.manual_seed(0)
    m = ta.prepare_inference(ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=384).cuda().eval(), torch.bfloat16)
    x = torch.randn(4, 3, 384, 384, device="cuda")
    graphs = {}
    for name, (l, tune) in libs.items():
        hip._lib = l
        l.tramba_tune_set(hip.TUNE_GEMM_TILE, tune)
        graphs[name] = capture(m, x)
        l.tramba_tune_set(hip.TUNE_GEMM_TILE, 0)
    hip._lib = new
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = {k: [] for k in graphs}
    for _ in range(rounds):
        for name, g in graphs.items():
            g.replay()
            torch.cuda.synchronize()
            a.record()
            for _ in range(20):
                g.replay()
            e.record()
            torch.cuda.synchronize()
            tot[name].append(a.elapsed_time(e) / 20)
    for name in graphs:
        t = sorted(tot[name])
        print(f"forward {name}: median {t[len(t) // 2]:.4f} ms  min {t[0]:.4f}  max {t[-1]:.4f}  ({4000.0 / t[len(t) // 2]:.1f} img/s)")


if __name__ == "__main__":
    main()
