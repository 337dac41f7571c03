#!/usr/bin/env python
"""Where does a small pair-IO launch spend its time?  ivx_conv_fwd_pio called back to back through the raw C-ABI (prebuilt arguments: ~5 us
of host time per call, so the device stays busy), event-timed over `iters` launches: sweeps of K, of the number of tiles, of the tile
config and of the epilogue's options on 1x1 / 3x3 layers of the KITTI trunk's /16 map.
  python tools/pio_scaling.py [--iters 50]"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imvoxelnet_amd import _lib, ops  # noqa: E402
from imvoxelnet_amd.ops import ConvDesc, IVX_F16_PAIR, _ptr, _scale_ptr, _stream  # noqa: E402


def build(B, H, W, ci, co, k, out_pair=True, amax=True, res=False):
    g = torch.Generator().manual_seed(ci + co)
    x = (torch.randn(B, 1, H, W, ci, generator=g).abs_() * 2.0).cuda()
    w = torch.randn(co, ci, 1, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
    scale, shift = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1
    xp = ops.pair_from_float(x)
    packed, sp, wb, sb = ops.pair_pack_filters(w.permute(0, 2, 3, 4, 1).reshape(co, k * k, ci).contiguous(), scale, shift)
    packed, sp, shift = packed.cuda(), sp.cuda(), shift.cuda()
    d = ConvDesc(B, 1, H, W, ci, co, 1, k, k, 1, 1, 1, 0, k // 2, k // 2, 1, 0, 0, 0, 1, 0, 0, 1.0, IVX_F16_PAIR, IVX_F16_PAIR if out_pair else 0, 1.0)
    io = _lib.PairIO()
    io.in_scale, io.amax_in = _scale_ptr(xp.slots), _ptr(xp.slots)
    slots = ops.new_slots('cuda')
    io.amax_out = _ptr(slots) if amax else None
    io.out_scale = _scale_ptr(slots)
    io.wbound, io.sbound = float(wb), float(sb)
    keep = [x, xp, packed, sp, shift, slots]
    rd = None
    if res:
        r = ops.pair_from_float(torch.randn(B, 1, H, W, co, generator=g).cuda())
        d.res_mode = 1
        io.res_dtype, io.res_scale, io.amax_res = IVX_F16_PAIR, _scale_ptr(r.slots), _ptr(r.slots)
        rd = r.data
        keep.append(r)
    out = torch.empty((B, 1, H, W, 2 * co if out_pair else co), device='cuda', dtype=torch.float16 if out_pair else torch.float32)
    L = _lib.lib()
    wsb = max(int(L.ivx_conv_pio_workspace_bytes(C.byref(d), C.byref(io))), 0)
    ws = torch.empty((max(wsb, 256),), device='cuda', dtype=torch.uint8)
    st = _stream()
    args = (C.byref(d), C.byref(io), _ptr(xp.data), _ptr(packed), _ptr(sp), _ptr(shift), _ptr(rd), _ptr(out), _ptr(ws), wsb, st)
    keep += [out, ws, d, io]
    return args, keep


def timed(L, args, iters):
    for _ in range(5):
        L.ivx_conv_fwd_pio(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        L.ivx_conv_fwd_pio(*args)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--expand', action='store_true', help="only: the four Cout-expanding 1x1 layers with a residual at KITTI's sizes, by epilogue option and tile")
    a = ap.parse_args()
    L = _lib.lib()
    if a.expand:
        cfgs = [int(c) for c in os.environ.get("PIO_CFGS", "0,74,73,81").split(",")]
        print('## 1x1 expansions of the bottlenecks (KITTI batch 4): us per launch (algorithmic GB/s) by epilogue option; tile configs ' + str(cfgs))
        for (h, w, ci, co) in ((96, 320, 64, 256), (48, 160, 128, 512), (24, 80, 256, 1024), (12, 40, 512, 2048)):
            M = 4 * h * w
            for out_pair, amax, res in ((True, True, True), (True, True, False), (False, False, False), (False, False, True)):
                args, keep = build(4, h, w, ci, co, 1, out_pair, amax, res)
                nbytes = (M * ci + M * co * (2 if res else 1) + ci * co) * 4
                row = []
                for c in cfgs:
                    L.ivx_conv_set_tile_override(c)
                    try:
                        t = timed(L, args, a.iters)
                        row.append(f'cfg {c}: {t:.1f} ({nbytes / t / 1e3:.0f})')
                    except Exception:
                        row.append(f'cfg {c}: -')
                    L.ivx_conv_set_tile_override(0)
                print(f'{ci}->{co} at {h}x{w} out_pair={out_pair} amax={amax} residual={res} [{nbytes / 1e6:.0f} MB]: ' + ' | '.join(row), flush=True)
                del args, keep
        return
    B, H, W = 4, 24, 80
    print('## K sweep: 1x1, M = 7680 rows (4 x 24 x 80), Cout 256, pair out; us per launch by tile config')
    cfgs = [int(c) for c in os.environ.get("PIO_CFGS", "0,66,166,74,174,73,81").split(",")]
    print('| Cin | ' + ' | '.join(f'cfg {c}' for c in cfgs) + ' |')
    print('|---|' + '---|' * len(cfgs))
    for ci in (64, 128, 256, 512, 1024, 2048, 4096):
        args, keep = build(B, H, W, ci, 256, 1)
        row = []
        for c in cfgs:
            L.ivx_conv_set_tile_override(c)
            try:
                row.append('%.1f' % timed(L, args, a.iters))
            except Exception:
                row.append('-')
            L.ivx_conv_set_tile_override(0)
        print(f'| {ci} | ' + ' | '.join(row) + ' |', flush=True)
    print('\n## tile-count sweep: 1x1 256 -> 256, pair out; us per launch')
    print('| map | rows | ' + ' | '.join(f'cfg {c}' for c in cfgs) + ' |')
    print('|---|---|' + '---|' * len(cfgs))
    for (h, w) in ((6, 20), (12, 40), (24, 80), (48, 160), (96, 320)):
        args, keep = build(B, h, w, 256, 256, 1)
        row = []
        for c in cfgs:
            L.ivx_conv_set_tile_override(c)
            row.append('%.1f' % timed(L, args, a.iters))
            L.ivx_conv_set_tile_override(0)
        print(f'| {h}x{w} | {B * h * w} | ' + ' | '.join(row) + ' |', flush=True)
    print('\n## epilogue options: 1024 -> 256 1x1 and 256 -> 256 3x3 at 24 x 80 x 4 (default plan); us per launch')
    for (ci, co, k) in ((1024, 256, 1), (256, 256, 3), (256, 1024, 1)):
        for out_pair, amax, res in ((True, True, False), (True, False, False), (False, True, False), (False, False, False), (True, True, True)):
            args, keep = build(B, H, W, ci, co, k, out_pair, amax, res)
            print(f'{ci}->{co} k{k} out_pair={out_pair} amax_out={amax} residual={res}: {timed(L, args, a.iters):.1f} us', flush=True)


if __name__ == '__main__':
    main()
