#!/usr/bin/env python
"""Do the HBM-bound Winograd transform kernels run BESIDE the MFMA-bound grouped GEMM when issued on a second stream?

Part 1 (per layer shape): time N grouped GEMMs alone, N (output + input) transform pairs alone, and both loops issued
concurrently on two streams, for several grid caps of the transform kernels / stream priorities / GEMM tile configs.
Perfect overlap: both ~= max(alone); none: both ~= sum.
Part 2 (whole KITTI neck, batch 4): sequential layer-by-layer execution vs the two-stream pipeline of pipeline.py.

  python tools/overlap_ubench.py [--iters 10] [--part 1|2|all]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from imvoxelnet_amd import ops, _lib, pipeline  # noqa: E402

# (name, (X,Y,Z), Cin, Cout, stride_z, pad)
SHAPES = [('64->64 z12', (216, 248, 12), 64, 64, 1, (1, 1, 1)),
          ('128->128 z6', (216, 248, 6), 128, 128, 1, (1, 1, 1)),
          ('256->256 z3', (216, 248, 3), 256, 256, 1, (1, 1, 1))]


def wall(fn, iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(iters)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def part1(a):
    L = _lib.lib()
    g = torch.Generator(device='cuda').manual_seed(0)
    B = a.batch
    for name, (X, Y, Z), ci, co, sz, pd in SHAPES:
        plan = ops.WinogradLayerPlan((B, X, Y, Z, ci), co, 3, sz, pd, True, 1, 6, has_res=True)
        x = torch.randn(B, X, Y, Z, ci, device='cuda', generator=g)
        res = torch.randn(plan.oshape, device='cuda', generator=g)
        out = torch.empty(plan.oshape, device='cuda')
        w0 = torch.randn(co, 3, 3, 3, ci, device='cuda', generator=g) * 0.02
        u = ops.conv_winograd_weights(w0, 1, 6)
        sc = torch.rand(co, device='cuda', generator=g) + 0.5
        sh = torch.randn(co, device='cuda', generator=g)
        wsa = torch.empty((plan.ws_bytes,), device='cuda', dtype=torch.uint8)
        wsb = torch.empty((plan.ws_bytes,), device='cuda', dtype=torch.uint8)
        plan.input(x, wsa)
        plan.input(x, wsb)
        plan.gemm(u, wsb)
        torch.cuda.synchronize()
        print(f'--- {name}  B={B}  GEMM {plan.gemm_flops / 1e9:.0f} GFLOP, transform pair '
              f'{(2 * plan.m_bytes / 2 + 2 * plan.v_bytes / 2 + 12.0 * out.numel()) / 1e9:.2f} GB', flush=True)
        for tile in [int(v) for v in a.gemm_cfgs.split(',')]:
            for prio in (0, -1):
                s1 = torch.cuda.Stream()
                s2 = torch.cuda.Stream(priority=prio)
                for cap in [int(v) for v in a.caps.split(',')]:
                    def gemm_loop(n):
                        L.ivx_conv_set_tile_override(tile)
                        with torch.cuda.stream(s1):
                            for _ in range(n):
                                plan.gemm(u, wsa)
                        L.ivx_conv_set_tile_override(0)

                    def xf_loop(n):
                        ops.winograd_set_transform_blocks(cap)
                        with torch.cuda.stream(s2):
                            for _ in range(n):
                                plan.output(sc, sh, res, out, wsb)
                                plan.input(out, wsb)
                        ops.winograd_set_transform_blocks(0)

                    def both(n):
                        L.ivx_conv_set_tile_override(tile)
                        ops.winograd_set_transform_blocks(cap)
                        for _ in range(n):
                            with torch.cuda.stream(s1):
                                plan.gemm(u, wsa)
                            with torch.cuda.stream(s2):
                                plan.output(sc, sh, res, out, wsb)
                                plan.input(out, wsb)
                        ops.winograd_set_transform_blocks(0)
                        L.ivx_conv_set_tile_override(0)

                    for f in (gemm_loop, xf_loop, both):
                        f(2)
                    tg, tx, tb = wall(gemm_loop, a.iters), wall(xf_loop, a.iters), wall(both, a.iters)
                    print(f'gemm cfg {tile:2d} prio {prio:2d} xf cap {cap:5d}: gemm {tg:6.3f}  xf {tx:6.3f}  both {tb:6.3f} ms   '
                          f'(sum {tg + tx:6.3f}, max {max(tg, tx):6.3f}; hidden {100 * (tg + tx - tb) / min(tg, tx):5.1f} % of the shorter)',
                          flush=True)


def part2(a):
    import imvoxelnet_amd as ia
    from kitti_cfg import kitti_model_cfg, KITTI_TEST_CFG
    model = ia.build_detector(kitti_model_cfg(), test_cfg=KITTI_TEST_CFG)
    ia.randomize_(model, 0)
    neck = model.neck_3d.prepare(torch.device('cuda'))
    vol = torch.randn(a.batch, 216, 248, 12, 64, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1))

    def run(n):
        for _ in range(n):
            neck.forward_cl(vol)

    pipeline.CHUNKS = 0
    run(2)
    ref = neck.forward_cl(vol).clone()
    t_seq = wall(run, a.iters)
    print(f'neck sequential: {t_seq:7.3f} ms', flush=True)
    for chunks in (2, 4):
        if a.batch % chunks:
            continue
        for prio in (-1, 0):
            for cap in [int(v) for v in a.caps.split(',')]:
                pipeline.CHUNKS, pipeline.XF_BLOCKS, pipeline.XF_PRIORITY = chunks, cap, prio
                run(2)
                y = neck.forward_cl(vol)
                same = bool(torch.equal(y, ref))
                t = wall(run, a.iters)
                print(f'neck pipelined chunks {chunks} prio {prio:2d} xf cap {cap:5d}: {t:7.3f} ms  ({100 * (t_seq - t) / t_seq:5.1f} % faster; '
                      f'bit-identical: {same})', flush=True)
    pipeline.CHUNKS = 0


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--part', default='all')
    ap.add_argument('--caps', default='0,256,512,1024,2048')
    ap.add_argument('--gemm-cfgs', default='0,51')
    a = ap.parse_args()
    if a.part in ('1', 'all'):
        part1(a)
    if a.part in ('2', 'all'):
        a.batch = 4
        part2(a)
