#!/usr/bin/env python
"""Workgroup timelines of the z-halo GEMM kernel on the KITTI neck layers (debug build: bash tools/build_timeline_lib.sh with TU 5 instead
of TU 4 -- see the script).  One forward pass of the layer with the timeline buffer set; only conv_wino_halo_kernel stamps.
  python tools/halo_timeline.py [--modes 30,33,42]"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imvoxelnet_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'bin', 'libimvoxel_hip_tl.so')
from imvoxelnet_amd import ops  # noqa: E402
from imvoxelnet_amd.conv import FusedConv  # noqa: E402

NECK = [('64->64 s111', 64, 64, (1, 1, 1), (216, 248, 12), 30), ('64->128 s112', 64, 128, (1, 1, 2), (216, 248, 12), 42),
        ('128->128 s111', 128, 128, (1, 1, 1), (216, 248, 6), 33), ('128->256 s112', 128, 256, (1, 1, 2), (216, 248, 6), 42)]


def q(t, f):
    t = t.double().flatten()
    if t.numel() > 1000000:
        t = t[torch.randperm(t.numel())[:1000000]]
    return float(torch.quantile(t, f))


def main():
    L = _lib.lib()
    L.ivx_conv_set_timeline.argtypes = [C.c_void_p]
    g = torch.Generator().manual_seed(0)
    FusedConv.winograd, FusedConv.wino_operands = True, 4
    for name, ci, co, st, (D, H, W), mode in NECK:
        w = torch.randn(co, ci, 3, 3, 3, generator=g) * (2.0 / (ci * 27)) ** 0.5
        bn = (torch.rand(co, generator=g) + .5, torch.randn(co, generator=g) * .1, torch.randn(co, generator=g) * .1, torch.rand(co, generator=g) + .5)
        x = torch.randn(4, D, H, W, ci, generator=g).abs_().cuda()
        fc = FusedConv(w, bn=bn, stride=st, padding=1, relu=True, dims=3).to('cuda')
        L.ivx_conv_set_halo_mode(mode)
        for _ in range(2):
            fc(x)
        buf = torch.zeros(1 << 18, 8, dtype=torch.int64, device='cuda')
        torch.cuda.synchronize()
        L.ivx_conv_set_timeline(C.c_void_p(buf.data_ptr()))
        fc(x)
        torch.cuda.synchronize()
        L.ivx_conv_set_timeline(None)
        L.ivx_conv_set_halo_mode(-1)
        t = buf.cpu()
        t = t[t[:, 3] > 0]
        t0 = int(t[:, 0].min())
        st_, pro, kl, ep = (t[:, 0] - t0) / 100.0, (t[:, 1] - t[:, 0]) / 100.0, (t[:, 2] - t[:, 1]) / 100.0, (t[:, 3] - t[:, 2]) / 100.0
        end = (t[:, 3] - t0) / 100.0
        span = float(end.max())
        xcc = (t[:, 5] & 0xf).long()
        per_xcc_end = [float(end[xcc == i].max()) if (xcc == i).any() else 0.0 for i in range(8)]
        life = (t[:, 3] - t[:, 0]) / 100.0
        # concurrency: workgroup-microseconds / span / 256 CUs
        conc = float(life.sum()) / span / 256.0
        print(f'{name} mode {mode}: {len(t)} workgroups, span {span:.0f} us, mean resident workgroups per CU {conc:.2f}; per workgroup: prologue p50 {q(pro, .5):.1f} p90 {q(pro, .9):.1f} | '
              f'K loop p50 {q(kl, .5):.1f} p90 {q(kl, .9):.1f} | epilogue p50 {q(ep, .5):.1f} p90 {q(ep, .9):.1f} | life p50 {q(life, .5):.1f} p99 {q(life, .99):.1f}; '
              f'last end per XCC {[round(v) for v in per_xcc_end]}', flush=True)


if __name__ == '__main__':
    main()
