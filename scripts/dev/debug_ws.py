#!/usr/bin/env python3
"""debug aid for linear_ws_kernel: identity weights -> the output must equal the input; prints the structure of any mismatch"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tramba_amd import hip
dev = torch.device("cuda")
torch.manual_seed(0)


def run(m, n, k, form=19, **kw):
    hip.tune_set(hip.TUNE_GEMM_TILE, form)
    try:
        return kw["fn"]()
    finally:
        hip.tune_set(hip.TUNE_GEMM_TILE, 0)


for (m, n, k) in [(64, 128, 128), (4096, 128, 128), (36864, 128, 128), (36864, 256, 256), (36864, 512, 128)]:
    # x[r][c] encodes (r, c) exactly in bf16-safe integers: small ints
    r = torch.arange(m, device=dev)[:, None]
    c = torch.arange(k, device=dev)[None, :]
    x = ((r % 61) * 2 + (c % 2) + (c // 2 % 32) * 0.0).float()          # value identifies the row mod 61
    xc = c.float().expand(m, k).contiguous()                            # value identifies the column
    w = torch.zeros(n, k, device=dev)
    idx = torch.arange(n, device=dev)
    w[idx, idx % k] = 1.0
    for name, xin in (("rowcode", x), ("colcode", xc)):
        xb, wb = xin.bfloat16().contiguous(), w.bfloat16().contiguous()
        got = run(m, n, k, fn=lambda: hip.linear_cl(xb, wb, None, None, 0)).float()
        want = xb.float()[:, idx % k]
        bad = (got != want)
        print(f"M={m} N={n} K={k} {name}: mismatches {int(bad.sum())} of {bad.numel()}")
        if bad.any():
            rows = bad.any(dim=1).nonzero().flatten()
            cols = bad.any(dim=0).nonzero().flatten()
            print("   bad rows (first 40):", rows[:40].tolist(), "count", len(rows))
            print("   bad rows mod 32 histogram:", torch.bincount(rows % 32, minlength=32).tolist())
            print("   bad tiles (row // 32) first 20:", torch.unique(rows // 32)[:20].tolist(), "n", len(torch.unique(rows // 32)))
            print("   bad cols:", cols[:64].tolist(), "count", len(cols))
            rr = int(rows[0])
            print("   row", rr, "got ", got[rr, :32].tolist())
            print("   row", rr, "want", want[rr, :32].tolist())
torch.cuda.synchronize()
hip.device_error()

print("---- random data, form 19 against form 18, by epilogue")
for (m, n, k) in [(4096, 128, 128), (36864, 512, 128), (36864, 128, 256)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(m, k, generator=g).bfloat16().to(dev)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).bfloat16().to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    res = torch.randn(m, n, generator=g).bfloat16().to(dev)
    for name, fn in (("none", lambda: hip.linear_cl(x, w, None, None, 0)), ("bias", lambda: hip.linear_cl(x, w, bias, None, 0)),
                     ("gelu", lambda: hip.linear_cl(x, w, None, None, 2)), ("res", lambda: hip.linear_cl(x, w, None, res, 0)),
                     ("bias+gelu+res", lambda: hip.linear_cl(x, w, bias, res, 2))):
        a = run(m, n, k, 19, fn=fn).float()
        b = run(m, n, k, 18, fn=fn).float()
        bad = a != b
        print(f"M={m} N={n} K={k} {name:14s}: mismatches {int(bad.sum()):9d} of {bad.numel()}  max |d| {float((a - b).abs().max()):.4f}")
        if bad.any() and name in ("none", "bias", "res"):
            rows = bad.any(dim=1).nonzero().flatten()
            cols = bad.any(dim=0).nonzero().flatten()
            print("   bad rows mod 32 histogram:", torch.bincount(rows % 32, minlength=32).tolist())
            print("   bad cols (first 64):", cols[:64].tolist(), "count", len(cols))
            rr, = rows[:1].tolist()
            cc = bad[rr].nonzero().flatten()[:8].tolist()
            print("   row", rr, "cols", cc, "got", a[rr, cc].tolist(), "want", b[rr, cc].tolist())
