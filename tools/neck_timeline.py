#!/usr/bin/env python
"""Workgroup timelines of every launch of the KITTI 3-D neck IN SEQUENCE (the nine layers as the layer-by-layer host runs them: input transform ->
Winograd-domain GEMM -> output transform per layer; debug build tools/bin/libimvoxel_hip_tl.so).  One forward pass per launch: the n-th stage
call of the pass gets the timeline buffer, every other call runs plain.  Stamps: s_memrealtime (100 MHz) at a workgroup's entry and end; the GEMM
kernels (conv_wino_halo_kernel, conv_wino_zblk_kernel, conv_igemm_v4_kernel) also stamp "first group landed" and "K loop done".
  python tools/neck_timeline.py [--md out.md] [--batch 4]"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imvoxelnet_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'bin', 'libimvoxel_hip_tl.so')
import imvoxelnet_amd as ia  # noqa: E402
from imvoxelnet_amd import workloads as kc  # noqa: E402
from imvoxelnet_amd.conv import FusedConv  # noqa: E402


def q(t, f):
    t = t.double().flatten()
    if t.numel() > 1000000:
        t = t[torch.randperm(t.numel())[:1000000]]
    return float(torch.quantile(t, f))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--md', default=None)
    ap.add_argument('--batch', type=int, default=4)
    a = ap.parse_args()
    L = _lib.lib()
    L.ivx_conv_set_timeline.argtypes = [C.c_void_p]
    L.ivx_wino_set_timeline.argtypes = [C.c_void_p]
    model = ia.build_detector(kc.kitti_model_cfg(), test_cfg=dict(kc.KITTI_TEST_CFG))
    ia.randomize_(model, 0)
    dev = torch.device('cuda')
    model.neck_3d.prepare(dev)
    vol = torch.randn(a.batch, 216, 248, 12, 64, generator=torch.Generator().manual_seed(1)).abs_().to(dev)
    state = {'n': 0, 'target': -1, 'log': []}
    buf = torch.zeros(1 << 21, 8, dtype=torch.int64, device='cuda')
    # (with FusedConv.trace set the layer-by-layer host issues the three stages as separate calls: ops.conv_winograd_fwd)
    names = {'ivx_conv_winograd_input': ('input transform', L.ivx_wino_set_timeline), 'ivx_conv_winograd_gemm': ('GEMM', L.ivx_conv_set_timeline),
             'ivx_conv_winograd_output': ('output transform', L.ivx_wino_set_timeline)}

    def wrap(fname):
        real = getattr(L, fname)
        kind, setter = names[fname]

        def w(*args):
            d = args[0]._obj
            i = state['n']
            state['n'] += 1
            state['log'].append((kind, d.Cin, d.Cout, d.sw, d.W))
            if i == state['target']:
                setter(C.c_void_p(buf.data_ptr()))
                try:
                    return real(*args)
                finally:
                    setter(None)
            return real(*args)
        setattr(L, fname, w)
    for f in names:
        wrap(f)
    for _ in range(2):
        state['n'], state['log'] = 0, []
        FusedConv.trace = []
        model.neck_3d.forward_cl(vol)
        FusedConv.trace = None
    torch.cuda.synchronize()
    log = list(state['log'])
    lines = [f'# KITTI neck, batch {a.batch}: {len(log)} stage launches per pass (tools/neck_timeline.py; us; stamps cost a few % of a launch)', '',
             '| # | layer (Cin -> Cout, z stride, slices) | stage | workgroups | span | resident per CU (mean) | start p50 / p90 | prologue p50 / p90 | K loop p50 / p90 | epilogue p50 / p90 | life p50 / p90 / p99 | last end per XCC (min .. max) |',
             '|---|---|---|---|---|---|---|---|---|---|---|---|']
    for tgt, (kind, ci, co, sw, zz) in enumerate(log):
        buf.zero_()
        state['n'], state['target'] = 0, tgt
        FusedConv.trace = []
        model.neck_3d.forward_cl(vol)
        FusedConv.trace = None
        torch.cuda.synchronize()
        state['target'] = -1
        t = buf.cpu()
        t = t[t[:, 3] > 0]
        if len(t) == 0:
            lines.append(f'| {tgt} | {ci} -> {co}, s{sw}, Z {zz} | {kind} | no stamps | | | | | | | | |')
            continue
        t0 = int(t[:, 0].min())
        st = (t[:, 0] - t0) / 100.0
        end = (t[:, 3] - t0) / 100.0
        life = (t[:, 3] - t[:, 0]) / 100.0
        span = float(end.max())
        conc = float(life.sum()) / span / 256.0
        xcc = (t[:, 5] & 0xf).long()
        ends = [float(end[xcc == i].max()) for i in range(8) if (xcc == i).any()]
        if kind == 'GEMM':
            pro, kl, ep = (t[:, 1] - t[:, 0]) / 100.0, (t[:, 2] - t[:, 1]) / 100.0, (t[:, 3] - t[:, 2]) / 100.0
            ph = f'{q(pro, .5):.1f} / {q(pro, .9):.1f} | {q(kl, .5):.1f} / {q(kl, .9):.1f} | {q(ep, .5):.1f} / {q(ep, .9):.1f}'
        else:
            ph = '- | - | -'
        lines.append(f'| {tgt} | {ci} -> {co}, s{sw}, Z {zz} | {kind} | {len(t)} | {span:.0f} | {conc:.2f} | {q(st, .5):.0f} / {q(st, .9):.0f} | {ph} | '
                     f'{q(life, .5):.1f} / {q(life, .9):.1f} / {q(life, .99):.1f} | {min(ends):.0f} .. {max(ends):.0f} |')
        print(lines[-1], flush=True)
    if a.md:
        with open(a.md, 'w') as f:
            f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
