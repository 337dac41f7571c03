"""A/B on the GPU box: the one-launch head of the trunk (ivx_amax_f32 + ivx_stem_pool_fwd_pair) against the three launches it replaces
(ivx_nchw_to_nhwc_amax, the 7x7 stem on fp32 MFMA, ivx_maxpool2d_fwd_pair).  python tools/stem_ab.py [--md out.md]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from bottleneck_ab import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--md', default=None)
    a = ap.parse_args()
    from imvoxelnet_amd import ops
    from test_gpu_stem import _stem
    f, fr, sp, _, _ = _stem(1)
    rows = []
    for name, N, H, W in [('kitti x4', 4, 384, 1280), ('scannet x50', 50, 480, 640), ('nuscenes x6', 6, 928, 1600), ('scannet x20', 20, 480, 640),
                          ('sunrgbd x1', 1, 480, 640)]:
        img = torch.randn(N, 3, H, W, generator=torch.Generator().manual_seed(1)).cuda()

        def old():
            x4 = ops.to_channels_last_amax(img, pad_to=4)
            return ops.maxpool2d_pair(f(x4), ops.slots_of(x4), f.wbound, f.sbound, 3, 2, 1)
        t_new = timed(lambda: ops.stem_pool_pair(img, fr, sp, f.shift, f.wbound, f.sbound))
        t_old = timed(old)
        islots = ops.new_slots(img.device)
        from imvoxelnet_amd import _lib
        import ctypes as C
        t_amax = timed(lambda: _lib.lib().ivx_amax_f32(C.c_void_p(img.data_ptr()), img.numel(), C.c_void_p(islots.data_ptr()),
                                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        by = img.numel() * 4 + N * ((H // 2 - 1) // 2 + 1) * ((W // 2 - 1) // 2 + 1) * 256
        rows.append((name, t_old * 1e3, t_new * 1e3, t_amax * 1e3, by / t_new / 1e6))
        print(f'{name}: three launches {t_old * 1e3:.1f} us  one launch (+ amax pass) {t_new * 1e3:.1f} us  (amax pass alone {t_amax * 1e3:.1f} us; {by / t_new / 1e6:.0f} GB/s image + pooled map)', flush=True)
    if a.md:
        with open(a.md, 'w') as fo:
            fo.write('| images | three launches (us) | amax + one launch (us) | amax pass alone (us) | GB/s (image + pooled pair map once) |\n|---|---|---|---|---|\n')
            for r in rows:
                fo.write(f'| {r[0]} | {r[1]:.1f} | {r[2]:.1f} | {r[3]:.1f} | {r[4]:.0f} |\n')


if __name__ == '__main__':
    main()
