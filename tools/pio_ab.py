#!/usr/bin/env python
"""Interleaved A/B of the 2-D trunk's layers in the chained fp16-pair form (ivx_conv_fwd_pio) against fp32 MFMA, per tile config: every
variant is timed `reps` times round-robin inside one process, the median is reported with the algorithmic GB/s and the TFLOP/s of
fp32 multiply-adds.
  python tools/pio_ab.py [--set kitti|views50] [--reps 7] [--cfgs 0,63,67,...] [--only 64->256]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imvoxelnet_amd import _lib, ops  # noqa: E402
from imvoxelnet_amd.conv import FusedConv  # noqa: E402

# name, Cin, Cout, k, stride, divisor of the image size (input map), residual ('' | 'same' | 'up'), out_pair
LAYERS = [('64->64 1x1 /4', 64, 64, 1, 1, 4, '', True), ('64->64 3x3 /4', 64, 64, 3, 1, 4, '', True), ('64->256 1x1 /4 res', 64, 256, 1, 1, 4, 'same', True),
          ('256->64 1x1 /4', 256, 64, 1, 1, 4, '', True), ('256->128 1x1 /4', 256, 128, 1, 1, 4, '', True), ('128->128 3x3 s2 /4', 128, 128, 3, 2, 4, '', True),
          ('256->512 1x1 s2 /4', 256, 512, 1, 2, 4, '', False), ('128->512 1x1 /8 res', 128, 512, 1, 1, 8, 'same', True), ('512->128 1x1 /8', 512, 128, 1, 1, 8, '', True),
          ('128->128 3x3 /8', 128, 128, 3, 1, 8, '', True), ('512->256 1x1 /8', 512, 256, 1, 1, 8, '', True), ('256->256 3x3 s2 /8', 256, 256, 3, 2, 8, '', True),
          ('512->1024 1x1 s2 /8', 512, 1024, 1, 2, 8, '', False), ('256->1024 1x1 /16 res', 256, 1024, 1, 1, 16, 'same', True),
          ('1024->256 1x1 /16', 1024, 256, 1, 1, 16, '', True), ('256->256 3x3 /16', 256, 256, 3, 1, 16, '', True), ('1024->512 1x1 /16', 1024, 512, 1, 1, 16, '', True),
          ('512->512 3x3 s2 /16', 512, 512, 3, 2, 16, '', True), ('1024->2048 1x1 s2 /16', 1024, 2048, 1, 2, 16, '', False),
          ('512->2048 1x1 /32 res', 512, 2048, 1, 1, 32, 'same', True), ('2048->512 1x1 /32', 2048, 512, 1, 1, 32, '', True), ('512->512 3x3 /32', 512, 512, 3, 1, 32, '', True),
          ('2048->64 1x1 /32 lat', 2048, 64, 1, 1, 32, '', False), ('1024->64 1x1 /16 lat', 1024, 64, 1, 1, 16, 'up', False),
          ('512->64 1x1 /8 lat', 512, 64, 1, 1, 8, 'up', False), ('256->64 1x1 /4 lat', 256, 64, 1, 1, 4, 'up', True), ('64->64 3x3 /4 fpn', 64, 64, 3, 1, 4, '', False),
          # FastIndoor configs: FPN with 256 channels
          ('256->256 1x1 /4 lat256', 256, 256, 1, 1, 4, 'up', True), ('256->256 3x3 /4 fpn256', 256, 256, 3, 1, 4, '', False)]
SETS = {'kitti': (4, 384, 1280), 'views50': (50, 480, 640), 'views20': (20, 480, 640), 'nuscenes': (6, 928, 1600)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--set', default='kitti', choices=sorted(SETS))
    ap.add_argument('--reps', type=int, default=7)
    ap.add_argument('--iters', type=int, default=2)
    ap.add_argument('--cfgs', default='0,67,66,63,73,75,74,61,76,81')
    ap.add_argument('--only', default='')
    ap.add_argument('--epis', default='0', help='epilogue modes to interleave per pair config (ivx_conv_set_epilogue_mode): 0 default, 1 narrow')
    a = ap.parse_args()
    L = _lib.lib()
    B, IH, IW = SETS[a.set]
    cfgs = [int(c) for c in a.cfgs.split(',')]
    epis = [int(e) for e in a.epis.split(',')]
    cfgs = [(c, e) for c in cfgs for e in epis]
    g = torch.Generator().manual_seed(0)
    tot = {'f32': 0.0, 'pair default': 0.0, 'pair best': 0.0}
    print(f'# {a.set}: batch {B}, image {IH}x{IW}; median ms over {a.reps} interleaved repetitions; f32 = FusedConv fp32 path (Winograd where it applies)')
    print('| layer | f32 ms | ' + ' | '.join(f'pair cfg {c}' + (f' epi {e}' if len(epis) > 1 else '') for c, e in cfgs) + ' | best | GB/s (best) | TFLOP/s fp32-equivalent (best) |')
    print('|---|---|' + '---|' * (len(cfgs) + 3))
    for name, ci, co, k, st, div, res_kind, out_pair in LAYERS:
        if a.only and a.only not in name:
            continue
        H, W = IH // div, IW // div
        w = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
        bn = (torch.rand(co, generator=g) + .5, torch.randn(co, generator=g) * .1, torch.randn(co, generator=g) * .1, torch.rand(co, generator=g) + .5)
        fc = FusedConv(w, bn=bn, stride=st, padding=k // 2, relu=True, dims=2, chain=True).to('cuda')
        x = torch.randn(B, 1, H, W, ci, generator=g).cuda().relu_()
        Ho, Wo = (H + 2 * (k // 2) - k) // st + 1, (W + 2 * (k // 2) - k) // st + 1
        res = rp = None
        res_mode = 0
        if res_kind == 'same':
            res = torch.randn(B, 1, Ho, Wo, co, generator=g).cuda()
            rp = ops.pair_from_float(res)
        elif res_kind == 'up':
            res = torch.randn(B, 1, Ho // 2, Wo // 2, co, generator=g).cuda()
            res.ivx_slots = ops.new_slots('cuda')
            res.ivx_slots[:1].view(torch.float32)[0] = float(res.abs().max())
            rp, res_mode = res, 2
        xp = ops.pair_from_float(x)
        nbytes = (x.numel() + B * Ho * Wo * co + (res.numel() if res is not None else 0) + w.numel()) * 4
        flops = 2.0 * B * Ho * Wo * co * ci * k * k
        variants = [('f32', None)] + [(f'pair {c}' + (f' e{e}' if len(epis) > 1 else ''), (c, e)) for c, e in cfgs]
        times = {n: [] for n, _ in variants}
        bad = set()

        def run(n, c):
            if c is None:
                L.ivx_conv_set_tile_override(0)
                return fc(x, res=res, res_mode=res_mode)
            L.ivx_conv_set_tile_override(c[0])
            L.ivx_conv_set_epilogue_mode(c[1])
            try:
                return fc(xp, res=rp, res_mode=res_mode, out_pair=out_pair)
            finally:
                L.ivx_conv_set_tile_override(0)
                L.ivx_conv_set_epilogue_mode(0)
        for n, c in variants:          # warm-up + which configs the layer takes
            try:
                run(n, c)
            except Exception:
                bad.add(n)
        torch.cuda.synchronize()
        for _ in range(a.reps):
            for n, c in variants:
                if n in bad:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run(n, c)
                e1.record()
                torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) / a.iters)
        med = {n: (statistics.median(t) if t else float('nan')) for n, t in times.items()}
        pairs = {n: v for n, v in med.items() if n != 'f32' and v == v}
        best = min(pairs, key=pairs.get)
        tot['f32'] += med['f32']
        tot['pair default'] += med.get('pair 0', med.get(f'pair 0 e{epis[0]}', float('nan')))
        tot['pair best'] += pairs[best]
        print(f'| {name} {H}x{W} | {med["f32"]:.4f} | ' + ' | '.join(f'{med[n]:.4f}' for n, _ in variants[1:]) +
              f' | {best} | {nbytes / pairs[best] / 1e6:.0f} | {flops / pairs[best] / 1e9:.1f} |', flush=True)
    print('# sums over the listed layers (one launch each): ' + ', '.join(f'{k} {v:.3f} ms' for k, v in tot.items()))


if __name__ == '__main__':
    main()
