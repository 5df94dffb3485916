"""In-process A/B of one module-level switch on the Tramba-V 384x384 batch-4 bf16 forward (BASELINE config 2).
Box-to-box and run-to-run spread is ~2 %, larger than most single optimisations, so both variants are captured as
hipGraphs in ONE process and replayed alternately.

    python scripts/ab_forward.py tramba_amd.models.OVERLAP_BRANCHES
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta  # noqa: E402


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
    modname, attr = sys.argv[1].rsplit(".", 1)
    mod = importlib.import_module(modname)
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    torch.manual_seed(0)
    m = ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=384).cuda().eval()
    m = ta.prepare_inference(m, torch.bfloat16)
    x = torch.randn(4, 3, 384, 384, device="cuda")
    graphs = {}
    for val in (False, True):
        setattr(mod, attr, val)
        graphs[val] = capture(m, x)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = {False: [], True: []}
    for _ in range(rounds):
        for val in (False, True):
            g = graphs[val]
            g.replay()
            torch.cuda.synchronize()
            a.record()
            for _ in range(20):
                g.replay()
            e.record()
            torch.cuda.synchronize()
            tot[val].append(a.elapsed_time(e) / 20)
    for val in (False, True):
        t = sorted(tot[val])
        print(f"{attr}={val}: median {t[len(t) // 2]:.4f} ms  min {t[0]:.4f}  max {t[-1]:.4f}")


if __name__ == "__main__":
    main()
