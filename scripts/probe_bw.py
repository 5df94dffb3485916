import torch
dev="cuda"
def t(fn, nbytes, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/20*1e3
    print(f"{name:40s} {us:8.1f} us {nbytes/us/1e3:8.1f} GB/s")
n=4*1024*9216
u=torch.randn(n,device=dev).bfloat16(); d=torch.randn(n,device=dev).bfloat16(); o=torch.empty(n,device=dev)
f=torch.randn(n,device=dev); g=torch.empty_like(f)
t(lambda: g.copy_(f), n*8, "fp32 copy (4B in, 4B out)")
t(lambda: o.copy_(u), n*6, "bf16->fp32 cast (2B in, 4B out)")
t(lambda: torch.add(u,d,out=o), n*8, "bf16+bf16->fp32 (4B in, 4B out)")
big=torch.randn(4*n,device=dev); big2=torch.empty_like(big)
t(lambda: big2.copy_(big), 4*n*8, "fp32 copy 1.2GB")
t(lambda: big.mul_(1.0001), 4*n*8, "fp32 inplace scale 0.6GB rw")
