#!/usr/bin/env python
"""Interleaved A/B of the Winograd transform kernels (csrc/winograd.hip) on the KITTI neck layers at batch 4: input / output kernel
variants (ivx_conv_winograd_set_variant), every combination timed round-robin inside one process; reports the median ms and the
algorithmic GB/s of each, and checks every variant's output against the first combination.
  python tools/wino_ab.py [--reps 5] [--iters 3] [--out 0,1,2,3] [--inp 0,1]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imvoxelnet_amd import ops, _lib  # noqa: E402

CASES = [('64->64 z12 +res', (216, 248, 12), 64, 64, 1, True),
         ('128->128 z6 +res', (216, 248, 6), 128, 128, 1, True),
         ('256->256 z3 +res', (216, 248, 3), 256, 256, 1, True),
         ('64->128 s112', (216, 248, 12), 64, 128, 2, False)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--out', default='0,1,2,3')
    ap.add_argument('--inp', default='0,1')
    a = ap.parse_args()
    outs, inps = ([int(v) for v in s.split(',')] for s in (a.out, a.inp))
    L = _lib.lib()
    g = torch.Generator(device='cuda').manual_seed(0)
    for name, (X, Y, Z), ci, co, sz, has_res in CASES:
        x = torch.randn(a.batch, X, Y, Z, ci, device='cuda', generator=g)
        u = ops.conv_winograd_weights(torch.randn(co, 3, 3, 3, ci, device='cuda', generator=g) * 0.02, 1, 6)
        scale, shift = torch.rand(co, device='cuda') + 0.5, torch.randn(co, device='cuda')
        combos = [(i, outs[0]) for i in inps] + [(inps[0], o) for o in outs[1:]]
        state = {}
        ref = None
        for (i, o) in combos:
            L.ivx_conv_winograd_set_variant(o, i)
            plan = ops.WinogradLayerPlan((a.batch, X, Y, Z, ci), co, 3, sz, (1, 1, 1), True, 1, 6, has_res=has_res)
            ws = torch.empty((plan.ws_bytes,), device='cuda', dtype=torch.uint8)
            out = torch.empty(plan.oshape, device='cuda')
            res = torch.randn(plan.oshape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3)) if has_res else None
            plan.input(x, ws)
            plan.gemm(u, ws)
            plan.output(scale, shift, res, out, ws)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            else:
                err = float((out - ref).abs().max() / ref.abs().max())
                if err > 1e-5:
                    print(f'{name}: in {i} out {o}: max err {err:.2e} of the range vs the first combination')
            state[(i, o)] = (plan, ws, out, res)
        t_in, t_out, t_gemm = ({c: [] for c in combos} for _ in range(3))
        for rep in range(a.reps + 1):
            for c in combos:
                plan, ws, out, res = state[c]
                L.ivx_conv_winograd_set_variant(c[1], c[0])
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                ev[0].record()
                for _ in range(a.iters):
                    plan.input(x, ws)
                ev[1].record()
                for _ in range(a.iters):
                    plan.gemm(u, ws)
                ev[2].record()
                for _ in range(a.iters):
                    plan.output(scale, shift, res, out, ws)
                ev[3].record()
                torch.cuda.synchronize()
                if rep:
                    t_in[c].append(ev[0].elapsed_time(ev[1]) / a.iters)
                    t_gemm[c].append(ev[1].elapsed_time(ev[2]) / a.iters)
                    t_out[c].append(ev[2].elapsed_time(ev[3]) / a.iters)
        L.ivx_conv_winograd_set_variant(-1, -1)
        plan = state[combos[0]][0]
        b_in = x.numel() * 4 + plan.v_bytes
        b_out = plan.m_bytes + state[combos[0]][2].numel() * 4 * (2 if has_res else 1)
        print(f'== {name}: input transform {b_in / 1e9:.2f} GB, output transform {b_out / 1e9:.2f} GB (algorithmic)', flush=True)
        for c in combos:
            mi, mg, mo = (statistics.median(t[c]) for t in (t_in, t_gemm, t_out))
            print(f'  in {c[0]} out {c[1]:2d}: input {mi:6.3f} ms {b_in / mi / 1e6:6.0f} GB/s | gemm {mg:6.3f} ms | output {mo:6.3f} ms '
                  f'{b_out / mo / 1e6:6.0f} GB/s | sum {mi + mg + mo:6.3f}', flush=True)
        del state


if __name__ == '__main__':
    main()
