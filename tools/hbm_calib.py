#!/usr/bin/env python3
"""Development (GPU box): what this box's HBM delivers for plain torch fill / copy /
read kernels at the bench's buffer sizes -- the practical ceiling next to the 8 TB/s spec."""
import torch
dev = torch.device("cuda:0")
x = torch.empty(65536 * 200 * 24, dtype=torch.float32, device=dev).uniform_()
y = torch.empty(65536 * 200 * 42, dtype=torch.float32, device=dev)
z = torch.empty_like(y)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
s = t(lambda: y.fill_(1.0)); print(f"fill  2.2 GB write          : {y.numel()*4/s/1e9:8.0f} GB/s")
s = t(lambda: z.copy_(y)); print(f"copy  2.2 GB read + write   : {2*y.numel()*4/s/1e9:8.0f} GB/s")
s = t(lambda: x.sum()); print(f"sum   1.26 GB read          : {x.numel()*4/s/1e9:8.0f} GB/s")
s = t(lambda: (y.fill_(1.0), x.sum())); print(f"fill+sum 3.46 GB (serial)   : {(x.numel()+y.numel())*4/s/1e9:8.0f} GB/s")
