import torch, numpy as np
n=512; N=n**3
a=torch.rand(N+512*n, dtype=torch.float64, device='cuda'); b=torch.empty_like(a)
f=torch.zeros(N, dtype=torch.uint8, device='cuda')
def t(fn,k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    ts=[]
    for _ in range(k):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
ms=t(lambda: b.copy_(a)); print('flat copy 16 B/cell: %.3f ms %.0f GB/s'%(ms, 16*a.numel()/ms/1e6))
ms=t(lambda: torch.add(a, 1.0, out=b)); print('add 16 B/cell: %.3f ms %.0f GB/s'%(ms, 16*a.numel()/ms/1e6))
c=torch.empty_like(a)
ms=t(lambda: torch.add(a, c, out=b)); print('add3 24 B/cell: %.3f ms %.0f GB/s'%(ms, 24*a.numel()/ms/1e6))
ms=t(lambda: a.sum()); print('read-only sum 8 B/cell: %.3f ms %.0f GB/s'%(ms, 8*a.numel()/ms/1e6))
ms=t(lambda: b.fill_(1.0)); print('write-only 8 B/cell: %.3f ms %.0f GB/s'%(ms, 8*a.numel()/ms/1e6))
