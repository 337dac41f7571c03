#!/usr/bin/env python
"""Per-layer micro-benchmark of ivx_conv_fwd on the KITTI neck shapes (A/B of tile configs).
  python tools/conv_bench.py [--batch 4] [--cfgs 0,1,2,3] [--layers all|0,1,..] [--iters 5]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imvoxelnet_amd import ops, _lib  # noqa: E402

# name, (X,Y,Z), Cin, Cout, stride, pad
LAYERS = [
    ('b1.conv 64->64 z12', (216, 248, 12), 64, 64, (1, 1, 1), (1, 1, 1)),
    ('down1 64->128 s112', (216, 248, 12), 64, 128, (1, 1, 2), (1, 1, 1)),
    ('b2.conv 128->128 z6', (216, 248, 6), 128, 128, (1, 1, 1), (1, 1, 1)),
    ('down2 128->256 s112', (216, 248, 6), 128, 256, (1, 1, 2), (1, 1, 1)),
    ('b3.conv 256->256 z3', (216, 248, 3), 256, 256, (1, 1, 1), (1, 1, 1)),
    ('last 256->256 p0', (216, 248, 3), 256, 256, (1, 1, 1), (0, 0, 0)),
    # tail experiment: 128->128 at exactly 13 x 768 M-tiles (B=4) vs the KITTI shape's 13.08 rounds
    ('b2-like exact 9984 tiles', (208, 256, 6), 128, 128, (1, 1, 1), (1, 1, 1)),
    ('b2-like 9984+384 tiles', (216, 256, 6), 128, 128, (1, 1, 1), (1, 1, 1)),
]


# ResNet-50 + FPN(64) at 384x1280: name, (H,W), Cin, Cout, k, stride, pad   (2-D: D = 1)
def resnet_layers():
    L = [('stem 7x7 s2', (384, 1280), 4, 64, 7, 2, 3)]
    h, w, cin = 96, 320, 64
    for li, (planes, nb) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3))):
        for bi in range(nb):
            s = 2 if (bi == 0 and li > 0) else 1
            if bi == 0 or bi == 1:
                L.append((f'l{li+1}.{bi}.conv1 1x1', (h, w), cin, planes, 1, 1, 0))
                L.append((f'l{li+1}.{bi}.conv2 3x3 s{s}', (h, w), planes, planes, 3, s, 1))
                L.append((f'l{li+1}.{bi}.conv3 1x1', (h // s, w // s), planes, planes * 4, 1, 1, 0))
                if bi == 0:
                    L.append((f'l{li+1}.{bi}.down 1x1 s{s}', (h, w), cin, planes * 4, 1, s, 0))
            h, w, cin = h // s, w // s, planes * 4
    for i, (c, hh, ww) in enumerate(((256, 96, 320), (512, 48, 160), (1024, 24, 80), (2048, 12, 40))):
        L.append((f'fpn.lat{i} 1x1', (hh, ww), c, 64, 1, 1, 0))
    L.append(('fpn.out0 3x3', (96, 320), 64, 64, 3, 1, 1))
    L.append(('head 1x1 256->20', (214, 246), 256, 20, 1, 1, 0))
    return L


def run2d(a):
    L = _lib.lib()
    g = torch.Generator(device='cuda').manual_seed(0)
    for name, (H, W), ci, co, k, st, pd in resnet_layers():
        x = torch.randn(a.batch, 1, H, W, ci, device='cuda', generator=g)
        w = torch.randn(co, 1, k, k, ci, device='cuda', generator=g) * 0.02
        line = f'{name:26s} {ci:5d}->{co:5d} {H:4d}x{W:4d}'
        for cfg in [int(v) for v in a.cfgs.split(',')]:
            L.ivx_conv_set_tile_override(cfg)
            y = ops.conv_fwd(x, w, None, None, (1, k, k), (1, st, st), (0, pd, pd))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.conv_fwd(x, w, None, None, (1, k, k), (1, st, st), (0, pd, pd), out=y)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            line += f' | c{cfg}: {ms * 1e3:7.1f}us {2.0 * y.numel() * ci * k * k / ms / 1e9:6.1f}TF'
        print(line, flush=True)
    L.ivx_conv_set_tile_override(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--cfgs', default='0')
    ap.add_argument('--layers', default='all')
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--set', default='neck')
    ap.add_argument('--layout', type=int, default=1)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    ap.add_argument('--winograd', action='store_true', help='also time the F(m x m, 3x3) form of each stride-1 layer')
    ap.add_argument('--wcfgs', default='0', help='tile overrides of the grouped GEMM to sweep')
    ap.add_argument('--tile', type=int, default=4, help='m of the Winograd form (2, 4 or 6)')
    ap.add_argument('--plan1', action='store_true', help='A/B: the round-1 tile rule (ivx_conv_set_plan_mode(1))')
    ap.add_argument('--hw', default='384,1280', help='image size of --set resnet')
    ap.add_argument('--narrow', action='store_true', help='A/B: one-channel-per-lane epilogue stores (ivx_conv_set_epilogue_mode(1))')
    a = ap.parse_args()
    _lib.lib().ivx_conv_set_epilogue_mode(1 if a.narrow else 0)
    _lib.lib().ivx_conv_set_plan_mode(1 if a.plan1 else 0)
    if a.set == 'resnet':
        return run2d(a)
    L = _lib.lib()
    layers = range(len(LAYERS)) if a.layers == 'all' else [int(v) for v in a.layers.split(',')]
    g = torch.Generator(device='cuda').manual_seed(0)
    for li in layers:
        name, (X, Y, Z), ci, co, st, pd = LAYERS[li]
        dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
        ck = 64 if a.dtype == 'bf16' else 32
        x = torch.randn(a.batch, X, Y, Z, ci, device='cuda', generator=g).to(dt)
        lay = a.layout if ci % ck == 0 else 0
        w = (torch.randn((co, 3, 3, 3, ci) if lay == 0 else (co, ci // ck, 3, 3, 3, ck), device='cuda', generator=g) * 0.02).to(dt)
        sc = torch.rand(co, device='cuda', generator=g) + 0.5
        sh = torch.randn(co, device='cuda', generator=g)
        for cfg in [int(v) for v in a.cfgs.split(',')]:
            L.ivx_conv_set_tile_override(cfg)
            try:
                y = ops.conv_fwd(x, w, sc, sh, (3, 3, 3), st, pd, relu=True, wgt_layout=lay)
            except Exception as e:  # noqa
                print(f'{name:24s} cfg {cfg}: {e}')
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.conv_fwd(x, w, sc, sh, (3, 3, 3), st, pd, relu=True, out=y, wgt_layout=lay)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            flops = 2.0 * y.numel() * ci * 27
            print(f'{name:24s} layout {lay} cfg {cfg}: {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s  ({flops / 1e9:.0f} GFLOP)', flush=True)
        L.ivx_conv_set_tile_override(0)
        if a.winograd and a.dtype == 'f32' and st[0] == 1 and st[1] == 1:
            # the same layer in the F(m x m, 3x3) form: per-stage times from the staged entry points
            w0 = w if lay == 0 else w.permute(0, 2, 3, 4, 1, 5).reshape(co, 3, 3, 3, ci).contiguous()
            u = ops.conv_winograd_weights(w0, lay, a.tile)
            for wcfg in [int(v) for v in a.wcfgs.split(',')]:      # tile override of the grouped GEMM
                L.ivx_conv_set_tile_override(wcfg)
                yw = ops.conv_winograd_fwd(x, u, sc, sh, 3, st[2], pd, True, wgt_layout=lay)
                torch.cuda.synchronize()
                err = (yw - y).abs().max().item() / max(y.abs().max().item(), 1e-30)
                ops.winograd_trace = []
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    ops.conv_winograd_fwd(x, u, sc, sh, 3, st[2], pd, True, out=yw, wgt_layout=lay)
                e1.record()
                torch.cuda.synchronize()
                tr, ops.winograd_trace = ops.winograd_trace, None
                ms = e0.elapsed_time(e1) / a.iters
                stage = {k: sum(s0.elapsed_time(s1) for n, s0, s1, _ in tr if n == k) / a.iters for k in ('input', 'gemm', 'output')}
                gf = sum(f for n, _, _, f in tr if n == 'gemm') / a.iters
                print(f'{name:24s} winograd F{a.tile} cfg {wcfg}: {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s direct-equivalent | input '
                      f'{stage["input"]:.3f} gemm {stage["gemm"]:.3f} ({gf / stage["gemm"] / 1e9:.1f} TFLOP/s executed) output '
                      f'{stage["output"]:.3f} ms | max rel diff vs direct {err:.2e}', flush=True)
            L.ivx_conv_set_tile_override(0)


if __name__ == '__main__':
    main()
