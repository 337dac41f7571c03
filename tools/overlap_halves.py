#!/usr/bin/env python
"""Experiment (round 6): the KITTI batch of 4 as TWO half batches through two native handles on two HIP streams -- do the latency-bound trunk launches
of one half overlap with the throughput-bound neck launches of the other?  Prints ms per batch of 4 for: one handle / one stream (the product), two
handles concurrently (both start together), two handles with the second delayed behind the first's trunk.
  PYTHONPATH=. python tools/overlap_halves.py [--steps 20]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=4)
    a = ap.parse_args()
    import imvoxelnet_amd as ia
    from imvoxelnet_amd import engine
    from imvoxelnet_amd.conv import FusedConv
    from imvoxelnet_amd.workloads import kitti_model_cfg, KITTI_TEST_CFG, kitti_meta
    dev = torch.device('cuda:0')
    model = ia.build_detector(kitti_model_cfg(), test_cfg=KITTI_TEST_CFG)
    ia.randomize_(model, 0)
    with torch.no_grad():
        model.bbox_head.conv_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(5))
        model.bbox_head.conv_cls.bias.fill_(-2.0)
        model.bbox_head.conv_reg.weight.normal_(0, 0.002, generator=torch.Generator().manual_seed(6))
        model.bbox_head.conv_dir_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(7))
    FusedConv.wino_operands = 4
    model.prepare(dev)
    A = model._native
    Bn = engine.NativeModel(model, dev)
    B = a.batch
    img = torch.randn(B, 1, 3, 384, 1280, generator=torch.Generator().manual_seed(1000)).to(dev)
    metas = [kitti_meta(t=(0.01 * b, 0.0, 0.0), box_type=ia.LiDARInstance3DBoxes) for b in range(B)]
    h = B // 2
    i1, i2 = img[:h].contiguous(), img[h:].contiguous()
    m1, m2 = metas[:h], metas[h:]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def whole():
        out = A.detect(img, metas)
        return out[3].cpu()

    def halves():
        with torch.cuda.stream(s1):
            o1 = A.detect(i1, m1)
        with torch.cuda.stream(s2):
            o2 = Bn.detect(i2, m2)
        s1.synchronize(); s2.synchronize()
        return torch.cat([o1[3].cpu(), o2[3].cpu()])

    def halves_seq():
        o1 = A.detect(i1, m1)
        o2 = Bn.detect(i2, m2)
        return torch.cat([o1[3].cpu(), o2[3].cpu()])

    for name, fn in (('one handle, batch %d' % B, whole), ('two half batches, one stream', halves_seq), ('two half batches, two streams', halves),
                     ('one handle, batch %d (again)' % B, whole)):
        for _ in range(4):
            c = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            c = fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        print(f'{name}: {ms:.3f} ms per batch of {B} = {B / ms * 1e3:.1f} images/s   (kept {c.tolist()})', flush=True)


if __name__ == '__main__':
    main()
