"""Would a fused Mlp kernel (LayerNorm-folded fc1 -> GELU -> fc2 + residual in ONE launch, hidden tile kept on chip) beat the two
launches it replaces on the 24x24 / 12x12 stages?  A lower bound for it from the library's own GEMM kernel: a workgroup of a
fused Mlp owns BM rows and does BOTH contractions for them -- 2 * BM * C * 4C * 2 flop -- streaming all of fc1's and fc2's
weights through its CU; without a cross-CU reduction the launch has only M / BM workgroups.  The same arithmetic per workgroup
and the same number of workgroups is a plain GEMM with M rows, ONE 64-column tile and K = 2 * C * 4C / 64: its time is what the
64-row tile engine needs for the fused kernel's MFMA + staging work, before GELU, the second operand's LDS round trip or the
residual.  Printed next to the two launches the model runs today (fc1 + GELU with LayerNorm folded in, fc2 + residual)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip
dev = torch.device("cuda")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (m, c, what) in ((2304, 512, "24x24 stage, batch 4"), (576, 1024, "12x12 stage, batch 4"), (576, 512, "24x24 stage, batch 1")):
    hdim = 4 * c
    x = torch.randn(m, c, device=dev).bfloat16()
    w1 = (torch.randn(hdim, c, device=dev) * c ** -0.5).bfloat16()
    w2 = (torch.randn(c, hdim, device=dev) * hdim ** -0.5).bfloat16()
    b1, b2 = torch.randn(hdim, device=dev), torch.randn(c, device=dev)
    cs = w1.float().sum(1).contiguous()
    h = hip.linear_ln_cl(x, w1, cs, b1, 1e-5, None, 2)
    t1 = timeit(lambda: hip.linear_ln_cl(x, w1, cs, b1, 1e-5, None, 2))
    t2 = timeit(lambda: hip.linear_cl(h, w2, b2, x, 0))
    both = timeit(lambda: hip.linear_cl(hip.linear_ln_cl(x, w1, cs, b1, 1e-5, None, 2), w2, b2, x, 0))
    kb = 2 * c * hdim // 64
    xa = torch.randn(m, kb, device=dev).bfloat16()
    wb = (torch.randn(64, kb, device=dev) * kb ** -0.5).bfloat16()
    tb = timeit(lambda: hip.linear_cl(xa, wb, None, None, 0), n=10)
    print(f"{what}: M={m} C={c}: fc1+GELU {t1:.1f} us, fc2+residual {t2:.1f} us, the pair back to back {both:.1f} us; "
          f"{(m + 63) // 64} workgroups doing a fused Mlp's arithmetic (M={m}, N=64, K={kb}): {tb:.1f} us", flush=True)
