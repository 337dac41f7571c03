"""A/B on the GPU box: the one-launch identity bottleneck (ivx_bottleneck_fwd_pio) against the three-launch pair chain, per map size.
python tools/bottleneck_ab.py [--md out.md]"""
import argparse
import sys
import os
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--md', default=None)
    a = ap.parse_args()
    from imvoxelnet_amd import ops
    from test_gpu_bottleneck import _block
    from test_gpu_pair_chain import make_pair
    rows = []
    for name, P, B, H, W in [('kitti s1', 64, 4, 96, 320), ('kitti s2', 128, 4, 48, 160), ('scannet x50 s1', 64, 50, 120, 160),
                             ('scannet x50 s2', 128, 50, 60, 80), ('nuscenes s1', 64, 6, 232, 400), ('nuscenes s2', 128, 6, 116, 200),
                             ('scannet x20 s1', 64, 20, 120, 160), ('scannet x20 s2', 128, 20, 60, 80)]:
        (f1, f2, f3), _, _ = _block(P, 1)
        x = torch.relu(torch.randn(B, 1, H, W, 4 * P, generator=torch.Generator().manual_seed(1))).cuda()
        xp = make_pair(x)
        t_f = timed(lambda: ops.bottleneck_fwd_pio(xp, f1, f2, f3))
        t_c = timed(lambda: f3(f2(f1(xp, out_pair=True), out_pair=True), res=xp, out_pair=True))
        px = B * H * W
        fl = 2.0 * px * (4 * P * P + 9 * P * P + 4 * P * P) * 3
        by = px * 4 * P * 4 * 2
        rows.append((name, P, px, t_c * 1e3, t_f * 1e3, fl / t_f / 1e9, by / t_f / 1e6))
        print(f'{name}: chain {t_c * 1e3:.1f} us  fused {t_f * 1e3:.1f} us  ({fl / t_f / 1e9:.0f} TFLOP/s of fp16 products, {by / t_f / 1e6:.0f} GB/s in + out)', flush=True)
    # the first block of stage 1 (shortcut conv): one launch against the four-launch chain (the shortcut conv on the same stream)
    from test_gpu_bottleneck import _proj_block, _proj_chain
    prow = []
    for name, B, H, W in [('kitti layer1.0', 4, 96, 320), ('scannet x50 layer1.0', 50, 120, 160), ('nuscenes layer1.0', 6, 232, 400),
                          ('scannet x20 layer1.0', 20, 120, 160)]:
        (f1, f2, f3, fd, bank), _, _ = _proj_block(1)
        xp = make_pair(torch.relu(torch.randn(B, 1, H, W, 64, generator=torch.Generator().manual_seed(1))).cuda())
        t_f = timed(lambda: ops.bottleneck_proj_fwd_pio(xp, f1, f2, bank))
        t_c = timed(lambda: _proj_chain(xp, f1, f2, f3, fd))
        px = B * H * W
        fl = 2.0 * px * (64 * 64 + 9 * 64 * 64 + 2 * 64 * 256) * 3
        by = px * (64 + 256) * 4
        prow.append((name, px, t_c * 1e3, t_f * 1e3, fl / t_f / 1e9, by / t_f / 1e6))
        print(f'{name}: chain {t_c * 1e3:.1f} us  fused {t_f * 1e3:.1f} us  ({fl / t_f / 1e9:.0f} TFLOP/s of fp16 products, {by / t_f / 1e6:.0f} GB/s in + out)', flush=True)
    if a.md:
        with open(a.md, 'w') as f:
            f.write('| map (projection block 64 -> 64 -> 64 -> 256 + shortcut conv) | pixels | four launches (us) | one launch (us) | TFLOP/s (fp16 products) | GB/s (input + output once) |\n|---|---|---|---|---|---|\n')
            for r in prow:
                f.write(f'| {r[0]} | {r[1]} | {r[2]:.1f} | {r[3]:.1f} | {r[4]:.0f} | {r[5]:.0f} |\n')
            f.write('\n')
            f.write('| map | planes | pixels | three launches (us) | one launch (us) | TFLOP/s (fp16 products, no halo recompute) | GB/s (input + output once) |\n|---|---|---|---|---|---|---|\n')
            for r in rows:
                f.write(f'| {r[0]} | {r[1]} | {r[2]} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.0f} | {r[6]:.0f} |\n')


if __name__ == '__main__':
    main()
