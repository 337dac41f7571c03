#!/usr/bin/env python
"""Round 5: the fused GEMM + output form of F(4x4,3x3) (ivx_conv_winograd_gemm_output_amax) against the three-stage pipelines F(4x4) and
F(6x6) on the KITTI neck's stride-1 layers, batch 4; per-stage ms (median of `reps` interleaved repetitions), whole layer ms.
  python tools/fused_ab.py [--reps 7] [--layers 64,128]"""
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
    ap.add_argument('--layers', default='64,128')
    ap.add_argument('--batch', type=int, default=4)
    a = ap.parse_args()
    L = _lib.lib()
    P = ops.IVX_F16_PAIR
    shapes = {64: ((216, 248, 12), 64, 64), 128: ((216, 248, 6), 128, 128), 256: ((216, 248, 3), 256, 256)}
    for key in [int(k) for k in a.layers.split(',')]:
        (X, Y, Z), ci, co = shapes[key]
        g = torch.Generator(device='cuda').manual_seed(key)
        x = torch.randn(a.batch, X, Y, Z, ci, device='cuda', generator=g).clamp_min_(0)
        w = torch.randn(co, 3, 3, 3, ci, device='cuda', generator=g) * (2.0 / (27 * ci)) ** 0.5
        sc, sh = torch.rand(co, device='cuda', generator=g) + 0.5, torch.randn(co, device='cuda', generator=g) * 0.1
        for use_res in (False, True):
            res = torch.randn(a.batch, X, Y, Z, co, device='cuda', generator=g) if use_res else None
            out = torch.empty(a.batch, X, Y, Z, co, device='cuda')
            plans = {}
            for tile in (4, 6):
                u = ops.conv_winograd_weights(w, 1, tile, operands=P)
                d = ops._wino_desc(a.batch, X, Y, Z, ci, co, 3, 1, (1, 1, 1), True, 1, 1 if use_res else 0, operands=P)
                wsb = L.ivx_conv_winograd_workspace_bytes(C.byref(d), tile)
                plans[tile] = (u, d, torch.empty((wsb,), device='cuda', dtype=torch.uint8), wsb)
            stages = {
                'in6': lambda: L.ivx_conv_winograd_input(C.byref(plans[6][1]), 6, _ptr(x), _ptr(plans[6][2]), plans[6][3], _stream()),
                'gemm6': lambda: L.ivx_conv_winograd_gemm(C.byref(plans[6][1]), 6, _ptr(plans[6][0]), _ptr(plans[6][2]), plans[6][3], _stream()),
                'out6': lambda: L.ivx_conv_winograd_output(C.byref(plans[6][1]), 6, _ptr(sc), _ptr(sh), _ptr(res), _ptr(out), _ptr(plans[6][2]), plans[6][3], _stream()),
                'in4': lambda: L.ivx_conv_winograd_input(C.byref(plans[4][1]), 4, _ptr(x), _ptr(plans[4][2]), plans[4][3], _stream()),
                'gemm4': lambda: L.ivx_conv_winograd_gemm(C.byref(plans[4][1]), 4, _ptr(plans[4][0]), _ptr(plans[4][2]), plans[4][3], _stream()),
                'out4': lambda: L.ivx_conv_winograd_output(C.byref(plans[4][1]), 4, _ptr(sc), _ptr(sh), _ptr(res), _ptr(out), _ptr(plans[4][2]), plans[4][3], _stream()),
                'fused4': lambda: L.ivx_conv_winograd_gemm_output_amax(C.byref(plans[4][1]), 4, _ptr(plans[4][0]), _ptr(sc), _ptr(sh), _ptr(res), _ptr(out),
                                                                     _ptr(plans[4][2]), plans[4][3], None, _stream()),
            }
            for fn in stages.values():          # warm-up (and V / hdr in place for the gemm / output stages)
                rc = fn()
                assert rc == 0, _lib.lib().ivx_last_error()
            ref = out.clone()
            stages['in4'](); stages['gemm4'](); stages['out4']()
            o3 = out.clone()
            stages['fused4']()
            torch.cuda.synchronize()
            diff = float((out - o3).abs().max()) / float(o3.abs().max())
            times = {k: [] for k in stages}
            for _ in range(a.reps):
                for k, fn in stages.items():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn(); fn()
                    e1.record()
                    torch.cuda.synchronize()
                    times[k].append(e0.elapsed_time(e1) / 2)
            m = {k: statistics.median(v) for k, v in times.items()}
            print(f'{ci}->{co} z{Z} res={use_res}: F(6x6) in {m["in6"]:.3f} + gemm {m["gemm6"]:.3f} + out {m["out6"]:.3f} = {m["in6"] + m["gemm6"] + m["out6"]:.3f} | '
                  f'F(4x4) in {m["in4"]:.3f} + gemm {m["gemm4"]:.3f} + out {m["out4"]:.3f} = {m["in4"] + m["gemm4"] + m["out4"]:.3f} | '
                  f'fused: in {m["in4"]:.3f} + gemm+out {m["fused4"]:.3f} = {m["in4"] + m["fused4"]:.3f}  (fused vs 3-stage F(4x4): max diff {diff:.2e} of range)', flush=True)


if __name__ == '__main__':
    main()
