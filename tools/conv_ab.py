#!/usr/bin/env python
"""Interleaved A/B of tile configs on layers of the 2-D trunk at 50 views (ScanNet): every config is timed `reps` times in
round-robin order inside one process (box-to-box spread is +-10 %), the median per config is reported with the algorithmic
GB/s (input + output + residual + weights) and TFLOP/s.
  python tools/conv_ab.py [--dtype bf16|f32] [--reps 7] [--iters 3] [--views 50]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imvoxelnet_amd import _lib  # noqa: E402
from imvoxelnet_amd.conv import FusedConv  # noqa: E402

# name, Cin, Cout, k, stride, (H, W), residual
CASES = [('64->256 1x1 /4 res', 64, 256, 1, 1, (120, 160), True),
         ('256->64 1x1 /4', 256, 64, 1, 1, (120, 160), False),
         ('64->64 3x3 /4', 64, 64, 3, 1, (120, 160), False),
         ('128->512 1x1 /8 res', 128, 512, 1, 1, (60, 80), True),
         ('512->128 1x1 /8', 512, 128, 1, 1, (60, 80), False),
         ('128->128 3x3 /8', 128, 128, 3, 1, (60, 80), False),
         ('256->1024 1x1 /16 res', 256, 1024, 1, 1, (30, 40), True),
         ('1024->256 1x1 /16', 1024, 256, 1, 1, (30, 40), False),
         ('256->256 3x3 /16', 256, 256, 3, 1, (30, 40), False),
         ('512->2048 1x1 /32 res', 512, 2048, 1, 1, (15, 20), True),
         ('512->512 3x3 /32', 512, 512, 3, 1, (15, 20), False)]
CFGS = {'bf16': [0, 61, 63, 66, 67, 71, 73, 74, 72, 81, 82], 'f32': [0, 54, 49, 47, 46, 55, 51]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--reps', type=int, default=7)
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--views', type=int, default=50)
    ap.add_argument('--cfgs', default='')
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    L = _lib.lib()
    dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    cfgs = [int(c) for c in a.cfgs.split(',')] if a.cfgs else CFGS[a.dtype]
    g = torch.Generator().manual_seed(0)
    for name, ci, co, k, st, (H, W), has_res in CASES:
        if a.only and a.only not in name:
            continue
        w = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
        bn = (torch.rand(co, generator=g) + .5, torch.randn(co, generator=g) * .1, torch.randn(co, generator=g) * .1, torch.rand(co, generator=g) + .5)
        FusedConv.winograd = False            # tile A/B of the direct kernel
        fc = FusedConv(w, bn=bn, stride=st, padding=k // 2, relu=True, dims=2, dtype=dt, out_dtype=dt).to('cuda')
        x = torch.randn(a.views, 1, H, W, ci, generator=g).to(dt).cuda()
        res = torch.randn(a.views, 1, H, W, co, generator=g).to(dt).cuda() if has_res else None
        es = x.element_size()
        nbytes = (x.numel() + a.views * H * W * co * (2 if has_res else 1) + w.numel()) * es
        flops = 2.0 * a.views * H * W * co * ci * k * k
        times = {c: [] for c in cfgs}
        ref = None
        ok = []
        for c in cfgs:
            L.ivx_conv_set_tile_override(c)
            try:
                y = fc(x, res=res)
                torch.cuda.synchronize()
            except Exception as e:              # a config this dtype / shape does not take
                print(f'{name}: cfg {c} refused ({str(e)[:60]})')
                continue
            ok.append(c)
            if ref is None:
                ref = y
            elif not torch.allclose(y.float(), ref.float(), rtol=2e-2, atol=2e-2):
                print(f'{name}: cfg {c} differs from cfg {ok[0]}: max {(y.float() - ref.float()).abs().max().item():.3e}')
        for rep in range(a.reps + 1):
            for c in ok:
                L.ivx_conv_set_tile_override(c)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fc(x, res=res)
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    times[c].append(e0.elapsed_time(e1) / a.iters)
        L.ivx_conv_set_tile_override(0)
        line = f'{name:24s} {flops / 1e9:6.1f} GFLOP {nbytes / 1e6:7.1f} MB |'
        best = min(ok, key=lambda c: statistics.median(times[c]))
        for c in ok:
            med = statistics.median(times[c])
            line += f' {"*" if c == best else ""}{c}: {med:.3f} ms {nbytes / med / 1e6:5.0f} GB/s {flops / med / 1e9:4.0f} TF |'
        print(line, flush=True)


if __name__ == '__main__':
    main()
