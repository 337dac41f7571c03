#!/usr/bin/env python
"""Winograd-domain GEMM stage of a configuration's whole 3-D neck + head under each z-halo kernel mode (ivx_conv_set_halo_mode; -1 = the rule,
0 = the generic grouped kernel): sum of the 'wino_gemm' stage events per pass, layer-by-layer host.
  python tools/neck_halo_ab.py [--config scannet_v1] [--modes -1,0,30,33,10,13]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import imvoxelnet_amd as ia  # noqa: E402
from imvoxelnet_amd import workloads as kc, _lib  # noqa: E402
from imvoxelnet_amd.conv import FusedConv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='scannet_v1')
    ap.add_argument('--modes', default='-1,0,30,33,10,13')
    ap.add_argument('--reps', type=int, default=5)
    a = ap.parse_args()
    L = _lib.lib()
    cfg, B = {'kitti': (kc.kitti_model_cfg(), 4), 'nuscenes': (kc.nuscenes_model_cfg(), 1), 'scannet_fast': (kc.scannet_fast_model_cfg(), 1),
              'sunrgbd_fast': (kc.sunrgbd_fast_model_cfg(), 1), 'scannet_v1': (kc.scannet_v1_model_cfg(), 1)}[a.config]
    model = ia.build_detector(cfg, test_cfg=dict(nms_pre=100, max_num=50, use_rotate_nms=True, nms_thr=.1, score_thr=.1, iou_thr=.25))
    ia.randomize_(model, 0)
    model.neck_3d.prepare(torch.device('cuda'))
    model.bbox_head.prepare(torch.device('cuda'))
    nv = cfg['n_voxels']
    C = cfg['neck_3d'].get('in_channels') or cfg['neck_3d']['channels'][0]
    vol = torch.randn(B, nv[0], nv[1], nv[2], C, generator=torch.Generator().manual_seed(1)).relu_().cuda()

    def run():
        return model.bbox_head.forward_cl(model.neck_3d.forward_cl(vol))
    modes = [int(m) for m in a.modes.split(',')]
    tot = {m: [] for m in modes}
    per = {}
    try:
        for rep in range(a.reps + 1):
            for m in modes:
                L.ivx_conv_set_halo_mode(m)
                FusedConv.trace = []
                try:
                    run()
                    torch.cuda.synchronize()
                except Exception as e:
                    FusedConv.trace = None
                    tot[m] = None
                    print(f'mode {m}: refused ({str(e)[:100]})')
                    continue
                tr, FusedConv.trace = FusedConv.trace, None
                if rep and tot[m] is not None:
                    g = [(t[6], t[1].elapsed_time(t[2])) for t in tr if t[0] == 'wino_gemm']
                    tot[m].append(sum(x[1] for x in g))
                    for i, (d, ms) in enumerate(g):
                        per.setdefault((i, d), {}).setdefault(m, []).append(ms)
    finally:
        L.ivx_conv_set_halo_mode(-1)
    print(f'# {a.config}: Winograd-domain GEMM launches of neck + head, ms per pass (median of {a.reps})')
    for m in modes:
        if tot[m]:
            print(f'mode {m}: {sorted(tot[m])[len(tot[m]) // 2]:.3f}')
    print('| # | layer | ' + ' | '.join(f'mode {m}' for m in modes if tot[m]) + ' |')
    for (i, d), v in sorted(per.items()):
        print(f'| {i} | {d} | ' + ' | '.join(f'{sorted(v[m])[len(v[m]) // 2]:.4f}' if m in v else '-' for m in modes if tot[m]) + ' |')


if __name__ == '__main__':
    main()
