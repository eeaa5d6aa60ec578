import torch, time
dev='cuda:0'
x=torch.randn(8_000_000,128,device=dev)
y=torch.empty(8_000_000,384,device=dev)
def t(fn,n=10):
    fn(); torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
ms=t(lambda: torch.cat([x,x,x],1,out=y)); print("read 4.1GB write 12.3GB (cat): %.3f ms %.0f GB/s"%(ms,16.4e9/ms/1e6))
z=torch.empty_like(y)
ms=t(lambda: z.copy_(y)); print("copy 12.3->12.3: %.3f ms %.0f GB/s"%(ms,24.6e9/ms/1e6))
ms=t(lambda: y.fill_(1.0)); print("fill 12.3GB: %.3f ms %.0f GB/s"%(ms,12.3e9/ms/1e6))
ms=t(lambda: x.sum()); print("read 4.1GB: %.3f ms %.0f GB/s"%(ms,4.1e9/ms/1e6))
