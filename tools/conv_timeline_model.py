#!/usr/bin/env python
"""Workgroup timelines of chosen pair-IO launches INSIDE the KITTI trunk as the model runs it (layer-by-layer host; debug build
tools/bin/libimvoxel_hip_tl.so): the n-th ivx_conv_fwd_pio call of a forward pass gets the timeline buffer, every other call runs plain.
  python tools/conv_timeline_model.py [--shapes '512,128,1,48;128,128,3,48;512,2048,1,12']   (Cin,Cout,k,H of the input map)"""
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


def q(t, f):
    return float(torch.quantile(t.double(), f))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='512,128,1,48;128,128,3,48;128,512,1,48;256,1024,1,24;1024,256,1,24;512,2048,1,12;1024,512,1,24;64,256,1,96')
    a = ap.parse_args()
    want = [tuple(int(v) for v in s.split(',')) for s in a.shapes.split(';')]
    L = _lib.lib()
    L.ivx_conv_set_timeline.argtypes = [C.c_void_p]
    model = ia.build_detector(kc.kitti_model_cfg(), test_cfg=dict(nms_pre=100, max_num=50, use_rotate_nms=True, nms_thr=.1, score_thr=.1, iou_thr=.25))
    ia.randomize_(model, 0)
    model.backbone.prepare(torch.device('cuda'))
    model.neck.prepare(torch.device('cuda'))
    img = torch.randn(4, 1, 3, 384, 1280, generator=torch.Generator().manual_seed(1)).cuda()
    real = L.ivx_conv_fwd_pio
    state = {'n': 0, 'target': -1, 'log': []}
    buf = torch.zeros(1 << 16, 8, dtype=torch.int64, device='cuda')

    def wrapper(*args):
        d = args[0]._obj
        i = state['n']
        state['n'] += 1
        state['log'].append((d.Cin, d.Cout, d.KH, d.H))
        if i == state['target']:
            L.ivx_conv_set_timeline(C.c_void_p(buf.data_ptr()))
            try:
                return real(*args)
            finally:
                L.ivx_conv_set_timeline(None)
        return real(*args)
    L.ivx_conv_fwd_pio = wrapper
    for _ in range(3):
        state['n'] = 0
        state['log'] = []
        model.features_2d_cl(img)
    torch.cuda.synchronize()
    log = list(state['log'])
    for shp in want:
        if shp not in log:
            print(f'{shp}: not in the trunk ({sorted(set(log))[:6]} ...)')
            continue
        idx = [i for i, s in enumerate(log) if s == shp]
        tgt = idx[min(1, len(idx) - 1)]          # the second launch of the shape (the first of a stage has a different producer)
        buf.zero_()
        state['n'], state['target'] = 0, tgt
        model.features_2d_cl(img)
        torch.cuda.synchronize()
        state['target'] = -1
        t = buf.cpu()
        t = t[t[:, 3] > 0]
        if len(t) == 0:
            print(f'{shp}: no stamps (split-K or a non-pio path)')
            continue
        t0 = int(t[:, 0].min())
        st, pro, kl, ep = (t[:, 0] - t0) / 100.0, (t[:, 1] - t[:, 0]) / 100.0, (t[:, 2] - t[:, 1]) / 100.0, (t[:, 3] - t[:, 2]) / 100.0
        span = (int(t[:, 3].max()) - t0) / 100.0
        print(f'{shp[0]}->{shp[1]} k{shp[2]} at H={shp[3]} (call {tgt} of {len(log)}, preceded by {log[tgt - 1]}): {len(t)} workgroups, span {span:.1f} us; '
              f'start p50 {q(st, .5):.1f} p90 {q(st, .9):.1f} max {float(st.max()):.1f} | prologue p50 {q(pro, .5):.1f} p90 {q(pro, .9):.1f} max {float(pro.max()):.1f} | '
              f'K loop p50 {q(kl, .5):.1f} p90 {q(kl, .9):.1f} | epilogue p50 {q(ep, .5):.1f} p90 {q(ep, .9):.1f} max {float(ep.max()):.1f}', flush=True)


if __name__ == '__main__':
    main()
