#!/usr/bin/env python
"""Workgroup timeline of the one-launch bottleneck (csrc/bottleneck.hip, debug build with -DIVX_CONV_TIMELINE: tools/build_timeline_lib.sh).
Every workgroup stamps s_memrealtime (100 MHz) at entry, after conv1's K loop, after conv1's epilogue (mid1 in LDS), after conv2's K loop, after
conv2's epilogue, after conv3's last K slab and at its end; prints the distribution of each segment per map size.
  python tools/bottleneck_timeline.py [--md out.md]"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from imvoxelnet_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'bin', 'libimvoxel_hip_tl.so')


def q(t, f):
    return float(torch.quantile(t.double(), f))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--md', default=None)
    a = ap.parse_args()
    from imvoxelnet_amd import ops
    from test_gpu_bottleneck import _block
    from test_gpu_pair_chain import make_pair
    L = _lib.lib()
    L.ivx_bottleneck_set_timeline.argtypes = [C.c_void_p]
    lines = ['| map | workgroups | span us | conv1 K loop p50 / p90 | conv1 epilogue | conv2 K loop | conv2 epilogue | conv3 K loop (+ epilogues of earlier units) | last epilogue | whole p50 / p90 |', '|---|---|---|---|---|---|---|---|---|---|']
    for name, P, B, H, W in [('kitti s1', 64, 4, 96, 320), ('kitti s2', 128, 4, 48, 160), ('scannet x50 s1', 64, 50, 120, 160), ('scannet x50 s2', 128, 50, 60, 80)]:
        (f1, f2, f3), _, _ = _block(P, 1)
        x = torch.relu(torch.randn(B, 1, H, W, 4 * P, generator=torch.Generator().manual_seed(1))).cuda()
        xp = make_pair(x)
        for _ in range(3):
            ops.bottleneck_fwd_pio(xp, f1, f2, f3)
        torch.cuda.synchronize()
        buf = torch.zeros(1 << 16, 8, dtype=torch.int64, device='cuda')
        L.ivx_bottleneck_set_timeline(C.c_void_p(buf.data_ptr()))
        try:
            ops.bottleneck_fwd_pio(xp, f1, f2, f3)
            torch.cuda.synchronize()
        finally:
            L.ivx_bottleneck_set_timeline(None)
        t = buf.cpu()
        t = t[t[:, 6] > 0]
        t0 = int(t[:, 0].min())
        span = (int(t[:, 6].max()) - t0) / 100.0
        seg = [(t[:, i + 1] - t[:, i]) / 100.0 for i in range(6)]
        whole = (t[:, 6] - t[:, 0]) / 100.0
        cells = ' | '.join(f'{q(s_, .5):.1f} / {q(s_, .9):.1f}' for s_ in seg)
        lines.append(f'| {name} | {len(t)} | {span:.1f} | {cells} | {q(whole, .5):.1f} / {q(whole, .9):.1f} |')
        print(lines[-1], flush=True)
    if a.md:
        with open(a.md, 'w') as f:
            f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
