#!/usr/bin/env python
"""Per-launch table of the 3-D neck + head convolutions as the layer-by-layer host runs them (FusedConv.trace events), with the split-operand
form of the layers the Winograd form does not take switched off and on (default rule of conv.py; --all: every 3-D layer, the round-3 opt-in):
  python tools/neck_layers.py [--config scannet_v1|scannet_fast|sunrgbd_fast|nuscenes|kitti] [--top 40]"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import imvoxelnet_amd as ia  # noqa: E402
from imvoxelnet_amd import workloads as kc  # noqa: E402
from imvoxelnet_amd.conv import FusedConv  # noqa: E402


def table(config, pair_mode, top, reps=5):
    FusedConv.pair_mode = pair_mode
    cfg, B = {'kitti': (kc.kitti_model_cfg(), 4), 'nuscenes': (kc.nuscenes_model_cfg(), 1), 'scannet_fast': (kc.scannet_fast_model_cfg(), 1),
              'sunrgbd_fast': (kc.sunrgbd_fast_model_cfg(), 1), 'scannet_v1': (kc.scannet_v1_model_cfg(), 1)}[config]
    model = ia.build_detector(cfg, test_cfg=dict(nms_pre=100, max_num=50, use_rotate_nms=True, nms_thr=.1, score_thr=.1, iou_thr=.25))
    ia.randomize_(model, 0)
    model.neck_3d.prepare(torch.device('cuda'))
    model.bbox_head.prepare(torch.device('cuda'))
    nv = cfg['n_voxels']
    C = cfg['neck_3d'].get('in_channels', cfg['neck_3d'].get('channels', [64])[0] if isinstance(cfg['neck_3d'].get('channels'), (list, tuple)) else 64)
    vol = torch.randn(B, nv[0], nv[1], nv[2], C, generator=torch.Generator().manual_seed(1)).relu_().cuda()

    def run():
        y = model.neck_3d.forward_cl(vol)
        return model.bbox_head.forward_cl(y)
    for _ in range(2):
        out = run()
    agg = collections.OrderedDict()
    for _ in range(reps):
        FusedConv.trace = []
        run()
        torch.cuda.synchronize()
        tr, FusedConv.trace = FusedConv.trace, None
        for t in tr:
            d = agg.setdefault((t[6], t[0]), [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += t[1].elapsed_time(t[2])
            d[2] += t[3]
            d[3] += t[4]
    rows = [(k, v[0] / reps, v[1] / reps, v[2] / reps, v[3] / reps) for k, v in agg.items()]
    total = sum(r[2] for r in rows)
    other = sum(r[2] for r in rows if not r[0][1].startswith('wino_'))
    form = {0: 'off', 1: 'on (every 3-D layer with Cin % 32 == 0)', -1: 'by the default rule'}[pair_mode]
    print(f'\n# {config}, split-operand form {form}: {total:.3f} ms of conv-stage time per step, {other:.3f} ms outside the Winograd form')
    print('| layer shape | stage | launches | ms/step | TFLOP/s executed | GB/s algorithmic |')
    print('|---|---|---|---|---|---|')
    for (desc, kind), n, ms, fl, by in sorted(rows, key=lambda r: -r[2])[:top]:
        if kind.startswith('wino_'):
            continue
        print(f'| {desc} | {kind} | {n:.0f} | {ms:.3f} | {fl / ms / 1e9 if fl else 0:.1f} | {by / ms / 1e6:.0f} |')
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='scannet_v1')
    ap.add_argument('--top', type=int, default=40)
    ap.add_argument('--min-pos', type=int, default=2000, help='FusedConv.pair_min_pos: fewest positions the split-operand form takes with --all')
    ap.add_argument('--all', action='store_true', help='second table with IVX_CONV_PAIR=1 (every 3-D layer with Cin %% 32 == 0) instead of the default rule')
    a = ap.parse_args()
    FusedConv.pair_min_pos = a.min_pos
    o0 = table(a.config, 0, a.top)
    o1 = table(a.config, 1 if a.all else -1, a.top)
    f0 = o0 if isinstance(o0, (list, tuple)) else [o0]
    f1 = o1 if isinstance(o1, (list, tuple)) else [o1]

    def flat(o):
        for t in o:
            if isinstance(t, (list, tuple)):
                yield from flat(t)
            elif torch.is_tensor(t):
                yield t
    for i, (a0, a1) in enumerate(zip(flat(f0), flat(f1))):
        print(f'# output {i}: max |on - off| / max |off| = {(a1.float() - a0.float()).abs().max().item() / max(a0.float().abs().max().item(), 1e-30):.2e}')


if __name__ == '__main__':
    main()
