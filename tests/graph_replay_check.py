#!/usr/bin/env python
"""Body of tests/test_gpu_model.py::test_graphed_simple_test_equals_eager, run in a process of its own (see the test's docstring)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]
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
    g = torch.Generator().manual_seed(3)
    imgs = [torch.randn(2, 1, 3, 192, 640, generator=g).cuda() for _ in range(3)]
    metas = [[kitti_meta(img_hw=(192, 640), t=(0.02 * k, 0.01 * b, 0.0), box_type=ia.LiDARInstance3DBoxes) for b in range(2)] for k in range(3)]
    refs = [model.simple_test(img, meta) for img, meta in zip(imgs, metas)]      # eager references first: nothing is allocated after capture
    torch.cuda.synchronize()
    graphed = model.capture_graph(imgs[0], metas[0])
    total = 0
    for k in (2, 1, 0, 2):
        got = graphed(imgs[k], metas[k])
        for r, o in zip(refs[k], got):
            assert torch.equal(r['scores_3d'], o['scores_3d']) and torch.equal(r['labels_3d'], o['labels_3d']), (k, len(r['scores_3d']), len(o['scores_3d']))
            assert torch.equal(r['boxes_3d'].tensor, o['boxes_3d'].tensor)
            total += len(r['scores_3d'])
    assert total > 0
    try:
        graphed(imgs[0][:1], metas[0][:1])
    except ValueError:
        pass
    else:
        raise AssertionError('a different batch must be refused')
    print('GRAPH_REPLAY_OK total=%d' % total)


if __name__ == '__main__':
    main()
