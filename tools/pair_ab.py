#!/usr/bin/env python
"""Split-operand (bf16 pair) convolution: accuracy against an fp64 reference and an interleaved timing A/B on the conv layers of
the KITTI neck (batch 4): fp32 MFMA direct kernel, fp32 Winograd form, pair form under several tile configs.
  python tools/pair_ab.py [--reps 5] [--batch 4] [--cfgs 0,74,81,82] [--skip-accuracy]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imvoxelnet_amd import _lib, ops  # noqa: E402
from imvoxelnet_amd.conv import FusedConv, pack_pair_weights  # noqa: E402

# name, Cin, Cout, stride, (D, H, W) of the input
NECK = [('64->64 s111', 64, 64, (1, 1, 1), (216, 248, 12)),
        ('64->128 s112', 64, 128, (1, 1, 2), (216, 248, 12)),
        ('128->128 s111', 128, 128, (1, 1, 1), (216, 248, 6)),
        ('128->256 s112', 128, 256, (1, 1, 2), (216, 248, 6)),
        ('256->256 s111', 256, 256, (1, 1, 1), (216, 248, 3))]


def accuracy():
    """small 3-D layer against an fp64 CPU convolution: fp32 MFMA direct, Winograd F(4), F(6) (where they apply) and the pair form"""
    g = torch.Generator().manual_seed(1)
    for ci, co, shape, act in ((64, 64, (2, 24, 26, 8), 1.0), (128, 128, (1, 30, 30, 6), 1.0), (256, 256, (1, 18, 20, 3), 1.0),
                               (64, 64, (2, 24, 26, 8), 1e-3), (64, 64, (2, 24, 26, 8), 300.0)):     # activation scale: the fp16 pair range
        B, D, H, W = shape
        w = torch.randn(co, ci, 3, 3, 3, generator=g) * (2.0 / (ci * 27)) ** 0.5
        x = torch.randn(B, D, H, W, ci, generator=g).abs_() * act          # post-ReLU-like activations
        ref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.double(), padding=1).permute(0, 2, 3, 4, 1)
        scale = ref.abs().max().item()
        rms = ref.pow(2).mean().sqrt().item()
        out = {}
        for mode in ('direct', 'wino4', 'wino6', 'wino4p', 'wino6p', 'pair', 'pair_naive'):
            FusedConv.pair_mode = 1 if mode.startswith('pair') else 0
            FusedConv.winograd = mode.startswith('wino')
            FusedConv.winograd_tile = int(mode[4]) if mode.startswith('wino') else 0
            FusedConv.wino_operands = 4 if mode.endswith('p') else 0
            FusedConv.winograd_min_pos = 0
            FusedConv.pair_min_pos = 0
            fc = FusedConv(w, padding=1, dims=3).to('cuda')
            if mode == 'pair_naive':     # the validation kernel on the same pair operands: plain fp32 products of (hi + lo)
                y = ops.conv_fwd(ops.bf16_pair_split(x.cuda()), fc.wp, None, None, (3, 3, 3), (1, 1, 1), (1, 1, 1), wgt_layout=1, pair=True, naive=True)
            else:
                y = fc(x.cuda())
            torch.cuda.synchronize()
            e = (y.double().cpu() - ref)
            out[mode] = (e.abs().max().item() / scale, e.pow(2).mean().sqrt().item() / rms)
        print(f'accuracy {ci}->{co} K={27 * ci} act x{act}: ' + ' | '.join(f'{m}: max/max|y| {a:.2e} rms/rms {b:.2e}' for m, (a, b) in out.items()), flush=True)
    FusedConv.winograd_tile = 0
    FusedConv.winograd_min_pos = 2000
    FusedConv.pair_min_pos = 2000
    FusedConv.wino_operands = 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--cfgs', default='0,73,75,63,74,81,82,83')
    ap.add_argument('--direct', action='store_true', help='also time the direct-form pair kernel (bf16 pairs)')
    ap.add_argument('--skip-accuracy', action='store_true')
    ap.add_argument('--only', default='')
    ap.add_argument('--halo', type=int, default=-1, help='ivx_conv_set_halo_mode: -1 default rule, 0 generic kernel only, 1 .. 4 force a z-halo config')
    a = ap.parse_args()
    L = _lib.lib()
    L.ivx_conv_set_halo_mode(a.halo)
    if not a.skip_accuracy:
        accuracy()
    cfgs = [int(c) for c in a.cfgs.split(',')]
    g = torch.Generator().manual_seed(0)
    for name, ci, co, st, (D, H, W) in NECK:
        if a.only and a.only not in name:
            continue
        w = torch.randn(co, ci, 3, 3, 3, generator=g) * (2.0 / (ci * 27)) ** 0.5
        bn = (torch.rand(co, generator=g) + .5, torch.randn(co, generator=g) * .1, torch.randn(co, generator=g) * .1, torch.rand(co, generator=g) + .5)
        x = torch.randn(a.batch, D, H, W, ci, generator=g).abs_().cuda()
        FusedConv.pair_mode, FusedConv.winograd = 0, True
        f_w = FusedConv(w, bn=bn, stride=st, padding=1, relu=True, dims=3).to('cuda')
        FusedConv.pair_mode = 1
        f_p = FusedConv(w, bn=bn, stride=st, padding=1, relu=True, dims=3).to('cuda') if a.direct else None
        FusedConv.pair_mode = 0
        # (name, layer, pair_mode, tile override, wino operands)
        runs = [('wino', f_w, 0, 0, 0)] + [(f'winoP{c}', f_w, 0, c, 4) for c in cfgs] + ([(f'pair{c}', f_p, 1, c, 0) for c in cfgs] if a.direct else [])
        ref = None
        ok = []
        for nm, fc, pm, c, wo in runs:
            FusedConv.pair_mode = pm
            FusedConv.wino_operands = wo
            L.ivx_conv_set_tile_override(c if (pm or wo) else 0)
            try:
                y = fc(x)
                torch.cuda.synchronize()
            except Exception as e:
                print(f'{name}: {nm} refused ({str(e)[:80]})')
                continue
            ok.append((nm, fc, pm, c, wo))
            if ref is None:
                ref = y
            else:
                print(f'{name}: {nm} vs wino: max |d| / max |y| = {(y - ref).abs().max().item() / ref.abs().max().item():.2e}')
        # the split pass alone
        xp = ops.bf16_pair_split(x)
        times = {nm: [] for nm, *_ in ok}
        times['split'] = []
        stages = {}
        for rep in range(a.reps + 1):
            for nm, fc, pm, c, wo in ok:
                FusedConv.pair_mode = pm
                FusedConv.wino_operands = wo
                L.ivx_conv_set_tile_override(c if (pm or wo) else 0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if not pm:
                    ops.winograd_trace = []
                e0.record()
                fc(x)
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    times[nm].append(e0.elapsed_time(e1))
                    if not pm:
                        stages.setdefault(nm, []).append([ea.elapsed_time(eb) for _, ea, eb, _ in ops.winograd_trace])
                ops.winograd_trace = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.bf16_pair_split(x, out=xp)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                times['split'].append(e0.elapsed_time(e1))
        L.ivx_conv_set_tile_override(0)
        od, oh, ow = ((n + 2 - 3) // s + 1 for n, s in zip((D, H, W), st))
        gf = 2.0 * a.batch * od * oh * ow * co * ci * 27 / 1e9
        sp = statistics.median(times['split'])
        line = f'{name:16s} {gf:6.0f} GFLOP | split {sp:.3f} ms ({8.0 * x.numel() / sp / 1e6:.0f} GB/s)'
        for nm, *_ in ok:
            t = statistics.median(times[nm])
            tt = t - sp if nm.startswith('pair') else t
            line += f' | {nm}: {t:.3f} ms' + (f' (gemm {tt:.3f} = {3 * gf / tt:.0f} TF bf16)' if nm.startswith('pair') else f' ({gf / t:.0f} TF eff)')
            if nm in stages:
                st3 = [statistics.median(v[i] for v in stages[nm]) for i in range(3)]
                line += ' [in %.3f gemm %.3f out %.3f]' % tuple(st3)
        print(line, flush=True)


if __name__ == '__main__':
    main()
