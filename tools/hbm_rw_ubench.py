#!/usr/bin/env python
"""HBM rates by direction on this box: write-only (fill), read-only (a max-reduction: ivx_amax-style kernel via torch), copy, and a
read-heavy / write-heavy mix (torch.add of two / out-of-place mul) over buffers well past the 256 MiB Infinity Cache."""
import torch
n = 1 << 28           # 1 GiB of fp32
a = torch.empty(n, device='cuda'); b = torch.empty(n, device='cuda'); c = torch.empty(n, device='cuda')
a.normal_(); b.normal_()
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
GB = n * 4 / 1e9
for name, f, nbytes in (('fill (write only)', lambda: c.fill_(1.5), GB), ('amax (read only)', lambda: a.abs().max() if False else torch.amax(a), GB),
                        ('sum (read only)', lambda: a.sum(), GB), ('copy (1R + 1W)', lambda: c.copy_(a), 2 * GB),
                        ('add (2R + 1W)', lambda: torch.add(a, b, out=c), 3 * GB), ('mul scalar (1R + 1W)', lambda: torch.mul(a, 2.0, out=c), 2 * GB)):
    ms = t(f)
    print(f'{name:24s} {ms:7.3f} ms  {nbytes / ms:7.1f} GB/s')
