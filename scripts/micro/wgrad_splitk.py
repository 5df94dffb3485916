"""Weight-gradient GEMM gy^T @ x with K = tokens (tall): one aten::mm against a chunked bmm + sum (split-K in Python)."""
import torch

dev = "cuda"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for m, n, k in ((73728, 512, 128), (73728, 128, 512), (73728, 128, 128), (73728, 256, 128), (18432, 1024, 256), (18432, 256, 1024),
                (18432, 512, 256), (4608, 2048, 512), (4608, 512, 2048), (4608, 1024, 512)):
    gy = torch.randn(m, n, device=dev).bfloat16()
    x = torch.randn(m, k, device=dev).bfloat16()
    base = timeit(lambda: torch.mm(gy.t(), x))
    ref = torch.mm(gy.t().float(), x.float())
    out = [f"mm {base:7.1f} us"]
    for s in (4, 8, 16, 32):
        if m % s:
            continue
        fn = lambda: torch.bmm(gy.view(s, m // s, n).transpose(1, 2), x.view(s, m // s, k)).float().sum(0)
        t = timeit(fn)
        err = float((fn() - ref).abs().max() / ref.abs().max())
        out.append(f"S={s}: {t:7.1f} us (err {err:.1e})")
    e0 = float((torch.mm(gy.t(), x).float() - ref).abs().max() / ref.abs().max())
    print(f"M={m} N={n} K={k}: ", " | ".join(out), f"| mm err {e0:.1e}", flush=True)
