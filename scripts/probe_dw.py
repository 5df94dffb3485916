import torch, time, torch.nn.functional as F
dev = "cuda"
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for (b, c, h, ks) in ((8, 256, 96, 3), (8, 512, 96, 7), (8, 1024, 24, 3)):
    for cl in (False, True):
        for cudnn in (True, False):
            x = torch.randn(b, c, h, h, device=dev, dtype=torch.bfloat16, requires_grad=True)
            w = torch.randn(c, 1, ks, ks, device=dev, dtype=torch.bfloat16, requires_grad=True)
            if cl:
                x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_()
            def run():
                with torch.backends.cudnn.flags(enabled=cudnn):
                    y = F.conv2d(x, w, None, padding=ks // 2, groups=c)
                    y.sum().backward()
            try:
                print(f"b{b} c{c} h{h} ks{ks} channels_last={cl} miopen={cudnn}: {t(run):8.2f} ms", flush=True)
            except Exception as e:
                print("fail", e)
