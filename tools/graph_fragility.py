#!/usr/bin/env python
"""Reproducer of a fragility of hipGraph replays of this path on the ROCm 7.2 / torch 2.10 stack of the GPU boxes: after a later FRESH
device allocation of a few hundred MB by the process (torch.empty; never written) a replay can run and return garbage (all-zero /
NaN activations -> no detections), permanently for that graph exec.  Seen with both backends of ImVoxelNet.capture_graph (torch
CUDAGraph of the composed path: [33, 34] -> [0, 0] after torch.empty(700 MB); native hipGraph: [27, 43] -> [0, 0] after 800 MB) and not
in other orders / sizes (native survived 900 MB in another run; a trivial torch graph is not affected).  Eager execution is never
affected.  Narrowed down to the runtime's graph packet capture: with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment before the HIP
runtime initialises the replays are exact (tests/graph_replay_check.py); `import imvoxelnet_amd` sets it by default, so run this
reproducer with DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 to see the failure.
  python tools/graph_fragility.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import imvoxelnet_amd as ia  # noqa: E402
from imvoxelnet_amd.workloads import kitti_model_cfg, KITTI_TEST_CFG, kitti_meta  # noqa: E402


def main():
    model = ia.build_detector(kitti_model_cfg(n_voxels=(104, 120, 12)), test_cfg=KITTI_TEST_CFG)
    ia.randomize_(model, 21)
    with torch.no_grad():
        model.bbox_head.conv_cls.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(5))
        model.bbox_head.conv_cls.bias.fill_(-1.0)
        model.bbox_head.conv_reg.weight.normal_(0, 0.002, generator=torch.Generator().manual_seed(6))
    model.prepare(torch.device('cuda'))
    img = torch.randn(2, 1, 3, 192, 640, generator=torch.Generator().manual_seed(3)).cuda()
    metas = [kitti_meta(img_hw=(192, 640), t=(0.0, 0.01 * b, 0.0), box_type=ia.LiDARInstance3DBoxes) for b in range(2)]
    keep = []
    for backend in ('torch', 'native'):
        g = model.capture_graph(img, metas, backend=backend)
        n0 = [len(r['scores_3d']) for r in g(img, metas)]
        keep.append(torch.empty(((700 + 100 * len(keep)) << 20,), device='cuda', dtype=torch.uint8))     # a FRESH allocation (never written)
        n1 = [len(r['scores_3d']) for r in g(img, metas)]
        print(f'{backend:7s} backend: detections right after capture {n0}, after a later torch.empty of {keep[-1].numel() >> 20} MB {n1}')


if __name__ == '__main__':
    main()
