#!/usr/bin/env python
"""Does the 256 MiB Infinity Cache (MALL) serve write -> read hand-offs between kernels on MI355X?
For buffer sizes S: (a) fill S bytes then immediately read them back (sum) -- warm; (b) the same with a 2 GiB flush
buffer touched in between -- cold; (c) repeated fill of the same buffer (does a write-back cache absorb rewrites?).
Reports GB/s of each kernel from HIP events."""
import torch

def t(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]

flush = torch.empty(2 << 28, device='cuda', dtype=torch.float32)   # 2 GiB
src = torch.empty(1 << 28, device='cuda', dtype=torch.float32)     # 1 GiB source for copies
src.normal_()
print('S MiB | fill GB/s | read warm GB/s | read cold GB/s | copy-in (read src, write S) GB/s | add (read S, write S) warm GB/s')
for mib in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024):
    n = mib * (1 << 20) // 4
    a = torch.empty(n, device='cuda', dtype=torch.float32)
    b = torch.empty(n, device='cuda', dtype=torch.float32)
    S = n * 4 / 1e9
    a.fill_(1.0); torch.cuda.synchronize()
    tf = t(lambda: a.fill_(2.0))
    def warm():
        a.fill_(3.0)
    # warm read: fill then sum, time only the sum
    def read_after(prep):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            prep()
            e0.record(); r = a.sum(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[2]
    tw = read_after(lambda: a.fill_(3.0))
    tc = read_after(lambda: (a.fill_(3.0), flush.fill_(0.0)))
    tcp = t(lambda: a.copy_(src[:n]))
    def add_after():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            a.fill_(1.0)
            e0.record(); torch.add(a, 1.0, out=b); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[2]
    ta = add_after()
    print(f'{mib:5d} | {S / tf * 1e3:8.0f} | {S / tw * 1e3:8.0f} | {S / tc * 1e3:8.0f} | {2 * S / tcp * 1e3:8.0f} | {2 * S / ta * 1e3:8.0f}', flush=True)
