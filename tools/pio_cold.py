#!/usr/bin/env python
"""A pair-IO launch with COLD caches: between timed launches a 1 GB copy sweeps the L2s and the Infinity Cache (and, with --icache, a
different conv kernel runs, which also evicts the instruction cache lines of the timed kernel), one event pair per launch.
  python tools/pio_cold.py [--cfgs 174,177,179] [--reps 20]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from imvoxelnet_amd import _lib  # noqa: E402
import pio_scaling as ps  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfgs', default='0,74,174,177,179')
    ap.add_argument('--reps', type=int, default=20)
    a = ap.parse_args()
    L = _lib.lib()
    big = torch.empty(256 << 20, dtype=torch.float32, device='cuda')
    big2 = torch.empty_like(big)
    other, keep_o = ps.build(4, 96, 320, 64, 64, 3)            # a different kernel instance / shape (icache + data)
    shapes = [(48, 160, 512, 128, 1, False), (48, 160, 128, 128, 3, False), (24, 80, 256, 1024, 1, True), (12, 40, 512, 2048, 1, True), (24, 80, 1024, 512, 1, False)]
    cfgs = [int(c) for c in a.cfgs.split(',')]
    for mode in ('warm', 'cold data', 'cold data + other conv kernel in between'):
        print(f'## {mode}: us per launch (median of {a.reps})')
        for (h, w, ci, co, k, res) in shapes:
            args, keep = ps.build(4, h, w, ci, co, k, True, True, res)
            row = []
            for c in cfgs:
                ts = []
                for r in range(a.reps + 3):
                    if mode != 'warm':
                        big2.copy_(big)
                    if mode.endswith('between'):
                        L.ivx_conv_set_tile_override(74)
                        L.ivx_conv_fwd_pio(*other)
                    L.ivx_conv_set_tile_override(c)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    L.ivx_conv_fwd_pio(*args)
                    e1.record()
                    L.ivx_conv_set_tile_override(0)
                    torch.cuda.synchronize()
                    if r >= 3:
                        ts.append(e0.elapsed_time(e1) * 1e3)
                row.append(f'cfg {c}: {statistics.median(ts):.1f}')
            print(f'{ci}->{co} k{k} {h}x{w} res={res}: ' + ' | '.join(row), flush=True)
            del args, keep


if __name__ == '__main__':
    main()
