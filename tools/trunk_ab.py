#!/usr/bin/env python
"""Interleaved A/B of the direct-convolution tile planner on the whole 2-D trunk (ResNet-50 + FPN level 0):
ivx_conv_set_plan_mode(0) = scored tile choice, (1) = the round-1 rule.  Median of `reps` alternating runs.
  python tools/trunk_ab.py [--reps 7]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import imvoxelnet_amd as ia  # noqa: E402
from imvoxelnet_amd import workloads as kc  # noqa: E402
from imvoxelnet_amd import _lib  # noqa: E402
from imvoxelnet_amd.conv import FusedConv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=7)
    a = ap.parse_args()
    L = _lib.lib()
    for name, cfg, shape in (('kitti 4 x 384x1280, FPN 64', kc.kitti_model_cfg(), (4, 1, 3, 384, 1280)),
                             ('scannet fast 50 x 480x640, FPN 256', kc.scannet_fast_model_cfg(), (1, 50, 3, 480, 640)),
                             ('scannet v1 50 x 480x640, FPN 64', kc.scannet_v1_model_cfg(), (1, 50, 3, 480, 640)),
                             ('sunrgbd 1 x 480x640, FPN 256', kc.sunrgbd_fast_model_cfg(), (1, 1, 3, 480, 640))):
        model = ia.build_detector(cfg, test_cfg=dict(nms_pre=100, max_num=50, use_rotate_nms=True, nms_thr=.1, score_thr=.1, iou_thr=.25))
        ia.randomize_(model, 0)
        model.backbone.prepare(torch.device('cuda'))
        model.neck.prepare(torch.device('cuda'))
        img = torch.randn(*shape, generator=torch.Generator().manual_seed(1)).cuda()
        times = {0: [], 1: []}
        flops = None
        for rep in range(a.reps + 1):
            for mode in (0, 1):
                L.ivx_conv_set_plan_mode(mode)
                if flops is None:
                    FusedConv.flops, FusedConv.exec_flops, FusedConv.count_flops = 0.0, 0.0, True
                    model.features_2d_cl(img)
                    FusedConv.count_flops = False
                    flops = FusedConv.exec_flops
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                model.features_2d_cl(img)
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    times[mode].append(e0.elapsed_time(e1))
        L.ivx_conv_set_plan_mode(0)
        m0, m1 = statistics.median(times[0]), statistics.median(times[1])
        print(f'{name:36s} executed {flops / 1e9:7.1f} GFLOP | scored {m0:7.3f} ms ({flops / m0 / 1e9:6.1f} TF) | round-1 rule {m1:7.3f} ms '
              f'({flops / m1 / 1e9:6.1f} TF) | {100 * (m1 - m0) / m1:+.1f} %', flush=True)
        del model, img
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
