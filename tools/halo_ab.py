#!/usr/bin/env python
"""Interleaved A/B of the z-halo kernel configurations (ivx_conv_set_halo_mode) on the Winograd-domain GEMMs of the KITTI neck layers
(batch 4, fp16 pair operands): per layer and mode the median time of the GEMM stage and of the whole layer, and whether the outputs are
bit-identical to the first mode's.
  python tools/halo_ab.py [--reps 7] [--s1 10,30,13,33] [--s2 22,42]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imvoxelnet_amd import _lib, ops  # noqa: E402
from imvoxelnet_amd.conv import FusedConv  # noqa: E402

NECK = [('64->64 s111', 64, 64, (1, 1, 1), (216, 248, 12)),
        ('64->128 s112', 64, 128, (1, 1, 2), (216, 248, 12)),
        ('128->128 s111', 128, 128, (1, 1, 1), (216, 248, 6)),
        ('128->256 s112', 128, 256, (1, 1, 2), (216, 248, 6)),
        ('256->256 s111', 256, 256, (1, 1, 1), (216, 248, 3))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=7)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--s1', default='10,30,13,33', help='halo modes tried on the stride-1 layers')
    ap.add_argument('--s2', default='22,42', help='... on the z-stride-2 layers')
    a = ap.parse_args()
    L = _lib.lib()
    g = torch.Generator().manual_seed(0)
    FusedConv.winograd, FusedConv.wino_operands = True, 4
    print(f'# z-halo kernel A/B, batch {a.batch}, median of {a.reps} interleaved repetitions: GEMM stage ms (whole layer ms)')
    for name, ci, co, st, (D, H, W) in NECK:
        modes = [int(m) for m in (a.s2 if st[2] == 2 else a.s1).split(',')]
        w = torch.randn(co, ci, 3, 3, 3, generator=g) * (2.0 / (ci * 27)) ** 0.5
        bn = (torch.rand(co, generator=g) + .5, torch.randn(co, generator=g) * .1, torch.randn(co, generator=g) * .1, torch.rand(co, generator=g) + .5)
        x = torch.randn(a.batch, D, H, W, ci, generator=g).abs_().cuda()
        fc = FusedConv(w, bn=bn, stride=st, padding=1, relu=True, dims=3).to('cuda')
        ref, same = None, {}
        for m in list(modes):
            L.ivx_conv_set_halo_mode(m)
            try:
                y = fc(x)
            except ValueError:          # a config built for another column height
                modes.remove(m)
                continue
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            same[m] = bool(torch.equal(y, ref))
        tg, tl = {m: [] for m in modes}, {m: [] for m in modes}
        for rep in range(a.reps + 1):
            for m in modes:
                L.ivx_conv_set_halo_mode(m)
                ops.winograd_trace = []
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fc(x)
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    tl[m].append(e0.elapsed_time(e1))
                    st_ms = [ea.elapsed_time(eb) for _, ea, eb, _ in ops.winograd_trace]
                    tg[m].append(st_ms[1] if len(st_ms) >= 3 else float('nan'))
                ops.winograd_trace = None
        L.ivx_conv_set_halo_mode(-1)
        print(f'{name:16s} | ' + ' | '.join(f'mode {m}: {statistics.median(tg[m]):.3f} ({statistics.median(tl[m]):.3f}){"" if same[m] else " DIFFERS"}' for m in modes), flush=True)


if __name__ == '__main__':
    main()
