"""In-process A/B of a library tuning knob (tramba_tune_set) on the Tramba-V 384x384 batch-4 bf16 forward: one hipGraph per
knob value captured in ONE process, replayed alternately.

    python scripts/ab_tune.py <knob index> <value> [<value> ...]      (0 = the library's own choice)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta  # noqa: E402
from tramba_amd import hip  # noqa: E402
from ab_forward import capture  # noqa: E402


def main():
    knob = int(sys.argv[1])
    values = [int(v) for v in sys.argv[2:]] or [0, 1]
    torch.manual_seed(0)
    m = ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=384).cuda().eval()
    m = ta.prepare_inference(m, torch.bfloat16)
    x = torch.randn(4, 3, 384, 384, device="cuda")
    graphs = {}
    for v in values:
        hip.tune_set(knob, v)
        graphs[v] = capture(m, x)
    hip.tune_set(knob, 0)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = {v: [] for v in values}
    for _ in range(8):
        for v in values:
            g = graphs[v]
            g.replay()
            torch.cuda.synchronize()
            a.record()
            for _ in range(20):
                g.replay()
            e.record()
            torch.cuda.synchronize()
            tot[v].append(a.elapsed_time(e) / 20)
    for v in values:
        t = sorted(tot[v])
        print(f"knob {knob} = {v}: median {t[len(t) // 2]:.4f} ms  min {t[0]:.4f}  max {t[-1]:.4f}")


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
