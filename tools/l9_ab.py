#!/usr/bin/env python
"""Round 5: the GEMM stage of the last KITTI neck layer (256 -> 256, z 3 -> 1, pad (1,1,0): a plain grouped GEMM with K = 3 Cin on the generic
LDS-DMA kernel) per tile config, incl. the deep-ring forms; ms, median of `reps` interleaved repetitions."""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imvoxelnet_amd import _lib, ops  # noqa: E402
from imvoxelnet_amd.ops import _ptr, _stream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=7)
    ap.add_argument('--cfgs', default='0,82,81,181,281,76,176,74,174')
    a = ap.parse_args()
    L = _lib.lib()
    P = ops.IVX_F16_PAIR
    B, (X, Y, Z), ci, co = 4, (216, 248, 3), 256, 256
    g = torch.Generator(device='cuda').manual_seed(9)
    x = torch.randn(B, X, Y, Z, ci, device='cuda', generator=g).clamp_min_(0)
    w = torch.randn(co, 3, 3, 3, ci, device='cuda', generator=g) * (2.0 / (27 * ci)) ** 0.5
    u = ops.conv_winograd_weights(w, 1, 6, operands=P)
    d = ops._wino_desc(B, X, Y, Z, ci, co, 3, 1, (1, 1, 0), True, 1, 0, operands=P)
    wsb = L.ivx_conv_winograd_workspace_bytes(C.byref(d), 6)
    ws = torch.empty((wsb,), device='cuda', dtype=torch.uint8)
    out = torch.empty(B, X, Y, 1, co, device='cuda')
    assert L.ivx_conv_winograd_input(C.byref(d), 6, _ptr(x), _ptr(ws), wsb, _stream()) == 0
    cfgs = [int(c) for c in a.cfgs.split(',')]
    ref = None
    times = {c: [] for c in cfgs}
    for rep in range(a.reps + 1):
        for c in cfgs:
            L.ivx_conv_set_tile_override(c)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = L.ivx_conv_winograd_gemm(C.byref(d), 6, _ptr(u), _ptr(ws), wsb, _stream())
            e1.record()
            L.ivx_conv_set_tile_override(0)
            torch.cuda.synchronize()
            if rc != 0:
                times[c].append(float('nan'))
                continue
            if rep == 0:
                L.ivx_conv_winograd_output(C.byref(d), 6, None, None, None, _ptr(out), _ptr(ws), wsb, _stream())
                torch.cuda.synchronize()
                if ref is None:
                    ref = out.clone()
                elif not torch.equal(out, ref):
                    print(f'cfg {c}: DIFFERS from cfg {cfgs[0]} (max {float((out - ref).abs().max()):.3e})')
            else:
                times[c].append(e0.elapsed_time(e1))
    print('256->256 z3->1 GEMM stage ms: ' + ' | '.join(f'cfg {c}: {statistics.median(t):.3f}' for c, t in times.items()))


if __name__ == '__main__':
    main()
