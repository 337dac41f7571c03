#!/usr/bin/env python
"""Workgroup timeline of one pair-IO launch of conv_igemm_v4_kernel (debug build with -DIVX_CONV_TIMELINE: tools/bin/libimvoxel_hip_tl.so).
Every workgroup records s_memrealtime (100 MHz) at entry, after the prologue barrier (first slabs landed), after the K loop and at its end;
prints, per launch: the span, and the distribution of start times and of the three phases.
  python tools/conv_timeline.py [--cfgs 0,74,475]"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imvoxelnet_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'bin', 'libimvoxel_hip_tl.so')
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import pio_scaling as ps  # noqa: E402


def q(t, f):
    return float(torch.quantile(t.double(), f))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfgs', default='0,74')
    ap.add_argument('--views50', action='store_true', help='the expanding 1x1 + residual layers and their neighbours at 50 ScanNet views (batch 50, 480 x 640 image)')
    a = ap.parse_args()
    L = _lib.lib()
    L.ivx_conv_set_timeline.argtypes = [C.c_void_p]
    shapes = [(96, 320, 64, 256, 1, True), (96, 320, 64, 256, 1, False), (96, 320, 256, 64, 1, False), (96, 320, 64, 64, 3, False),
              (48, 160, 128, 512, 1, True), (48, 160, 512, 128, 1, False), (48, 160, 128, 128, 3, False),
              (24, 80, 256, 1024, 1, True), (24, 80, 1024, 256, 1, False), (24, 80, 256, 256, 3, False),
              (12, 40, 512, 2048, 1, True), (12, 40, 2048, 512, 1, False), (12, 40, 512, 512, 3, False)]
    batch = 4
    if a.views50:
        batch = 50
        shapes = [(60, 80, 128, 512, 1, True), (30, 40, 256, 1024, 1, True), (30, 40, 1024, 256, 1, False), (30, 40, 256, 256, 3, False), (15, 20, 512, 2048, 1, True)]
    for (h, w, ci, co, k, res) in shapes:
        args, keep = ps.build(batch, h, w, ci, co, k, True, True, res)
        for cfg in [int(c) for c in a.cfgs.split(',')]:
            L.ivx_conv_set_tile_override(cfg)
            buf = torch.zeros(1 << 17, 8, dtype=torch.int64, device='cuda')
            try:
                for _ in range(3):
                    L.ivx_conv_fwd_pio(*args)
                torch.cuda.synchronize()
                L.ivx_conv_set_timeline(C.c_void_p(buf.data_ptr()))
                rc = L.ivx_conv_fwd_pio(*args)
                torch.cuda.synchronize()
            finally:
                L.ivx_conv_set_timeline(None)
                L.ivx_conv_set_tile_override(0)
            if rc:
                print(f'{ci}->{co} k{k} {h}x{w} cfg {cfg}: rc {rc}')
                continue
            t = buf.cpu()
            t = t[t[:, 3] > 0]
            if len(t) == 0:
                print(f'{ci}->{co} k{k} {h}x{w} res={res} cfg {cfg}: no stamps (split-K or a non-pio path)')
                continue
            t0 = int(t[:, 0].min())
            st, pro, kl, ep = (t[:, 0] - t0) / 100.0, (t[:, 1] - t[:, 0]) / 100.0, (t[:, 2] - t[:, 1]) / 100.0, (t[:, 3] - t[:, 2]) / 100.0
            span = (int(t[:, 3].max()) - t0) / 100.0
            xcc = t[:, 5] & 0xf
            print(f'{ci}->{co} k{k} {h}x{w} res={res} cfg {cfg}: {len(t)} workgroups, span {span:.1f} us; start p50 {q(st, .5):.1f} p90 {q(st, .9):.1f} max {float(st.max()):.1f} | '
                  f'prologue p50 {q(pro, .5):.1f} p90 {q(pro, .9):.1f} | K loop p50 {q(kl, .5):.1f} p90 {q(kl, .9):.1f} | epilogue p50 {q(ep, .5):.1f} p90 {q(ep, .9):.1f} | '
                  f'workgroups per XCC {torch.bincount(xcc.long(), minlength=8).tolist()}', flush=True)
        del args, keep


if __name__ == '__main__':
    main()
