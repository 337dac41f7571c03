#!/usr/bin/env python
"""Sub-batch pipelining experiment (round 5): does the latency-bound 2-D trunk of sub-batch i + 1 hide under the
bandwidth-bound 3-D neck of sub-batch i?  Two native handles (one with the trunk, one that starts from the FPN maps) on two
HIP streams, driven from one host thread; the trunk stream runs ahead, the neck stream waits on one event per sub-batch.

  python tools/overlap_ab.py [--batch 4] [--subs 1,2,4] [--steps 10]
Prints ms per step of ImVoxelNet.simple_test (one stream) and of the pipelined schedule per number of sub-batches, and whether the
detections agree."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import imvoxelnet_amd as ia  # noqa: E402
from imvoxelnet_amd import engine  # noqa: E402
from imvoxelnet_amd.workloads import kitti_model_cfg, KITTI_TEST_CFG, kitti_meta  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--subs', default='1,2,4')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    model = ia.build_detector(kitti_model_cfg(), test_cfg=KITTI_TEST_CFG)
    ia.randomize_(model, 0)
    with torch.no_grad():
        model.bbox_head.conv_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(5))
        model.bbox_head.conv_cls.bias.fill_(-2.0)
        model.bbox_head.conv_reg.weight.normal_(0, 0.002, generator=torch.Generator().manual_seed(6))
        model.bbox_head.conv_dir_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(7))
    model.prepare(dev)
    full = model._native
    tail = engine.NativeModel(model, dev, with_trunk=False)
    B, H, W = a.batch, 384, 1280
    img = torch.randn(B, 1, 3, H, W, generator=torch.Generator().manual_seed(1000)).to(dev)
    metas = [kitti_meta(t=(0.01 * b, 0.0, 0.0), box_type=ia.LiDARInstance3DBoxes) for b in range(B)]
    sT, sN = torch.cuda.Stream(), torch.cuda.Stream()

    def piped(nsub):
        bs = B // nsub
        cur = torch.cuda.current_stream()
        proj, orig, crop = model._camera_setup(metas, 4, dev)
        sT.wait_stream(cur)
        sN.wait_stream(cur)
        outs, keep = [], []
        for i in range(nsub):
            sl = slice(i * bs, (i + 1) * bs)
            with torch.cuda.stream(sT):
                p0 = full.backbone_fpn(img[sl].reshape(bs, 3, H, W))
                ev = torch.cuda.Event()
                ev.record(sT)
            with torch.cuda.stream(sN):
                sN.wait_event(ev)
                outs.append(tail.forward(p0, bs, 1, H, W, proj[sl].contiguous(), orig[sl].contiguous(), crop[sl].contiguous()))
            keep.append(p0)
        cur.wait_stream(sN)
        cur.wait_stream(sT)
        boxes, scores, labels, count = (torch.cat([o[k] for o in outs]) for k in range(4))
        return model._results_one_copy(boxes, scores, labels, count, metas)

    def timed(fn):
        for _ in range(a.warmup):
            out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3, out

    base_ms, ref = timed(lambda: model.simple_test(img, metas))
    print(f'simple_test (one stream, batch {B}): {base_ms:.3f} ms/step = {B / base_ms * 1e3:.1f} img/s', flush=True)
    for nsub in [int(s) for s in a.subs.split(',')]:
        if B % nsub:
            continue
        ms, out = timed(lambda: piped(nsub))
        same = all(len(x['scores_3d']) == len(y['scores_3d']) and torch.equal(x['labels_3d'], y['labels_3d']) and
                   torch.allclose(x['scores_3d'], y['scores_3d'], atol=1e-5) and torch.allclose(x['boxes_3d'].tensor, y['boxes_3d'].tensor, atol=1e-4)
                   for x, y in zip(ref, out))
        print(f'pipelined, {nsub} sub-batch(es) of {B // nsub}: {ms:.3f} ms/step = {B / ms * 1e3:.1f} img/s; same detections: {same}', flush=True)
    # serial reference of the split itself (same two handles, ONE stream): what the split costs without the overlap
    for nsub in [int(s) for s in a.subs.split(',')]:
        if B % nsub:
            continue
        bs = B // nsub

        def serial():
            proj, orig, crop = model._camera_setup(metas, 4, dev)
            outs = []
            for i in range(nsub):
                sl = slice(i * bs, (i + 1) * bs)
                p0 = full.backbone_fpn(img[sl].reshape(bs, 3, H, W))
                outs.append(tail.forward(p0, bs, 1, H, W, proj[sl].contiguous(), orig[sl].contiguous(), crop[sl].contiguous()))
            boxes, scores, labels, count = (torch.cat([o[k] for o in outs]) for k in range(4))
            return model._results_one_copy(boxes, scores, labels, count, metas)
        ms, _ = timed(serial)
        print(f'serial split, {nsub} sub-batch(es) of {bs} on one stream: {ms:.3f} ms/step', flush=True)


if __name__ == '__main__':
    main()
