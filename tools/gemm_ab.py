#!/usr/bin/env python
"""Interleaved A/B of grouped-GEMM tile configs on the Winograd-domain GEMMs of the KITTI neck: every config is timed
`reps` times in round-robin order inside one process (box-to-box and run-to-run spread is +-10 %, larger than most tile
effects), the median per config is reported.
  python tools/gemm_ab.py [--reps 7] [--iters 3]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imvoxelnet_amd import ops, _lib  # noqa: E402

CASES = [('64->64 z12', (216, 248, 12), 64, 64, 1, (1, 1, 1), [49, 56, 57, 52, 53, 43, 47, 46]),
         ('64->128 s112', (216, 248, 12), 64, 128, 2, (1, 1, 1), [54, 55, 51]),
         ('128->128 z6', (216, 248, 6), 128, 128, 1, (1, 1, 1), [54, 55, 51, 41, 58, 59]),
         ('128->256 s112', (216, 248, 6), 128, 256, 2, (1, 1, 1), [54, 55, 58]),
         ('256->256 z3', (216, 248, 3), 256, 256, 1, (1, 1, 1), [54, 55, 58])]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=7)
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--batch', type=int, default=4)
    a = ap.parse_args()
    L = _lib.lib()
    g = torch.Generator(device='cuda').manual_seed(0)
    for name, (X, Y, Z), ci, co, sz, pd, cfgs in CASES:
        plan = ops.WinogradLayerPlan((a.batch, X, Y, Z, ci), co, 3, sz, pd, True, 1, 6)
        x = torch.randn(a.batch, X, Y, Z, ci, device='cuda', generator=g)
        u = ops.conv_winograd_weights(torch.randn(co, 3, 3, 3, ci, device='cuda', generator=g) * 0.02, 1, 6)
        ws = torch.empty((plan.ws_bytes,), device='cuda', dtype=torch.uint8)
        plan.input(x, ws)
        times = {c: [] for c in cfgs}
        ref = None
        for c in cfgs:      # every config must produce the bits of the first one (same k order)
            L.ivx_conv_set_tile_override(c)
            ws.zero_()
            plan.input(x, ws)
            plan.gemm(u, ws)
            torch.cuda.synchronize()
            out = ws[ws.numel() - int(plan.m_bytes) - 512:].clone()
            if ref is None:
                ref = out
            elif not torch.equal(out, ref):
                print(f'{name}: cfg {c} differs from cfg {cfgs[0]} in {(out != ref).sum().item()} bytes')
        for rep in range(a.reps + 1):
            for c in cfgs:
                L.ivx_conv_set_tile_override(c)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    plan.gemm(u, ws)
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    times[c].append(e0.elapsed_time(e1) / a.iters)
        L.ivx_conv_set_tile_override(0)
        line = f'{name:14s} {plan.gemm_flops / 1e9:6.0f} GFLOP |'
        for c in cfgs:
            med = statistics.median(times[c])
            line += f' cfg {c}: {med:6.3f} ms ({plan.gemm_flops / med / 1e9:5.1f} TF, spread {min(times[c]):.3f}-{max(times[c]):.3f}) |'
        print(line, flush=True)


if __name__ == '__main__':
    main()
