#!/usr/bin/env python
"""Workgroup timeline of the one-launch stem (csrc/stem.hip, debug build: tools/build_timeline_lib.sh): s_memrealtime (100 MHz) at entry, after the
patch / filter loads, after the GEMM, after the BN + ReLU stage writes, after the pool + stores, at the end."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from imvoxelnet_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'bin', 'libimvoxel_hip_tl.so')


def q(t, f):
    return float(torch.quantile(t.double(), f))


def main():
    from imvoxelnet_amd import ops
    from test_gpu_stem import _stem
    L = _lib.lib()
    L.ivx_stem_set_timeline.argtypes = [C.c_void_p]
    f, fr, sp, _, _ = _stem(1)
    print('| images | workgroups | span us | loads p50 / p90 | GEMM | BN + ReLU -> LDS | pool + stores | amax commit | whole |\n|---|---|---|---|---|---|---|---|---|')
    for name, N, H, W in [('kitti x4', 4, 384, 1280), ('scannet x50', 50, 480, 640)]:
        img = torch.randn(N, 3, H, W, generator=torch.Generator().manual_seed(1)).cuda()
        for _ in range(3):
            ops.stem_pool_pair(img, fr, sp, f.shift, f.wbound, f.sbound)
        torch.cuda.synchronize()
        buf = torch.zeros(1 << 17, 8, dtype=torch.int64, device='cuda')
        L.ivx_stem_set_timeline(C.c_void_p(buf.data_ptr()))
        try:
            ops.stem_pool_pair(img, fr, sp, f.shift, f.wbound, f.sbound)
            torch.cuda.synchronize()
        finally:
            L.ivx_stem_set_timeline(None)
        t = buf.cpu()
        t = t[t[:, 5] > 0]
        t0 = int(t[:, 0].min())
        span = (int(t[:, 5].max()) - t0) / 100.0
        seg = [(t[:, i + 1] - t[:, i]) / 100.0 for i in range(5)]
        whole = (t[:, 5] - t[:, 0]) / 100.0
        st = (t[:, 0] - t0) / 100.0
        print(f'# {name}: workgroups that start within 3 us of the first: {int((st < 3.0).sum())}; start p50 {q(st, .5):.1f} us')
        print(f'| {name} | {len(t)} | {span:.1f} | ' + ' | '.join(f'{q(s_, .5):.1f} / {q(s_, .9):.1f}' for s_ in seg) + f' | {q(whole, .5):.1f} / {q(whole, .9):.1f} |', flush=True)


if __name__ == '__main__':
    main()
