#!/usr/bin/env python
"""Where the error of the "bf16 + fp8 2-D conv" mode comes from: a torch-CPU emulation of the ResNet-50 + FPN trunk in fp32
with e4m3 rounding applied to chosen tensors of every bottleneck -- the conv1 output x1 (per-tensor scale amax / 448), the conv2 output
x2, the conv2 / conv3 filters w2 / w3 (one scale per output channel, BN folded) -- and the FPN level-0 error of each choice against the
unquantised run, as a fraction of the signal's rms.  `2t` = a two-term filter: w ~ q1 + q2, both e4m3 on the same scale (q2 = the e4m3
rounding of the first term's residual), i.e. two fp8 MFMAs per product.  No GPU, no library: the numbers are a property of the formats.
  python tools/fp8_noise_budget.py [--views 2] [--seed 41]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import imvoxelnet_amd as ia  # noqa: E402
from imvoxelnet_amd import workloads as kc  # noqa: E402

E4M3_MAX = 448.0


def q8(x, scale):
    """e4m3 rounding of x / scale (saturating), back in fp32"""
    return (x / scale).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float() * scale


def q_act(x):
    return q8(x, x.abs().max().clamp_min(1e-30) / E4M3_MAX)


def q_w(w, terms):
    s = w.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-30) / E4M3_MAX
    q1 = q8(w, s)
    return q1 if terms == 1 else q1 + q8(w - q1, s)


def fold(sd, conv, bn):
    w = sd[conv + '.weight'].float()
    g, b, m, v = (sd[f'{bn}.{k}'].float() for k in ('weight', 'bias', 'running_mean', 'running_var'))
    sc = g / torch.sqrt(v + 1e-5)
    return w * sc.view(-1, 1, 1, 1), b - m * sc


def trunk(sd, img, mode):
    """mode: dict x1, x2 (bool), w2, w3 (0 exact | 1 | 2 terms), stages (set of stage indices where it applies)"""
    w, b = fold(sd, 'backbone.conv1', 'backbone.bn1')
    x = F.max_pool2d(F.relu(F.conv2d(img, w, b, stride=2, padding=3)), 3, 2, 1)
    outs = []
    for si, nb in enumerate((3, 4, 6, 3)):
        on = si in mode['stages']
        for j in range(nb):
            p = f'backbone.layer{si + 1}.{j}'
            st = 2 if (j == 0 and si > 0) else 1
            w1, b1 = fold(sd, p + '.conv1', p + '.bn1')
            w2, b2 = fold(sd, p + '.conv2', p + '.bn2')
            w3, b3 = fold(sd, p + '.conv3', p + '.bn3')
            idt = x
            if (p + '.downsample.0.weight') in sd:
                wd, bd = fold(sd, p + '.downsample.0', p + '.downsample.1')
                idt = F.conv2d(x, wd, bd, stride=st)
            y = F.relu(F.conv2d(x, w1, b1))
            if on and mode['x1']:
                y = q_act(y)
            y = F.relu(F.conv2d(y, q_w(w2, mode['w2']) if (on and mode['w2']) else w2, b2, stride=st, padding=1))
            if on and mode['x2']:
                y = q_act(y)
            y = F.conv2d(y, q_w(w3, mode['w3']) if (on and mode['w3']) else w3, b3)
            x = F.relu(y + idt)
        outs.append(x)
    lat = [F.conv2d(o, sd[f'neck.lateral_convs.{i}.conv.weight'].float(), sd[f'neck.lateral_convs.{i}.conv.bias'].float()) for i, o in enumerate(outs)]
    for i in range(3, 0, -1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
    return F.conv2d(lat[0], sd['neck.fpn_convs.0.conv.weight'].float(), sd['neck.fpn_convs.0.conv.bias'].float(), padding=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--views', type=int, default=2)
    ap.add_argument('--seed', type=int, default=41)
    a = ap.parse_args()
    torch.manual_seed(0)
    m = ia.build_detector(kc.scannet_v1_model_cfg(), test_cfg=dict(kc.SCANNET_V1_TEST_CFG))
    ia.randomize_(m, a.seed)
    sd = m.state_dict()
    img = torch.randn(a.views, 3, 480, 640, generator=torch.Generator().manual_seed(2))
    allst = {0, 1, 2, 3}
    base = dict(x1=False, x2=False, w2=0, w3=0, stages=allst)
    with torch.no_grad():
        ref = trunk(sd, img, base)
        rms = ref.pow(2).mean().sqrt()
        rows = [('x1 only', dict(base, x1=True)), ('x2 only', dict(base, x2=True)), ('w2 only', dict(base, w2=1)), ('w3 only', dict(base, w3=1)),
                ('the mode as built: x1, x2, w2, w3 (one-term filters)', dict(base, x1=True, x2=True, w2=1, w3=1)),
                ('x1, x2 + two-term filters', dict(base, x1=True, x2=True, w2=2, w3=2)),
                ('x2, w3 (conv2 stays bf16)', dict(base, x2=True, w3=1)),
                ('x2 + two-term w3 (conv2 stays bf16)', dict(base, x2=True, w3=2)),
                ('as built, stages 2-4 only', dict(base, x1=True, x2=True, w2=1, w3=1, stages={1, 2, 3})),
                ('as built, stages 3-4 only', dict(base, x1=True, x2=True, w2=1, w3=1, stages={2, 3})),
                ('two-term filters, stages 2-4 only', dict(base, x1=True, x2=True, w2=2, w3=2, stages={1, 2, 3})),
                ('two-term filters, stages 3-4 only', dict(base, x1=True, x2=True, w2=2, w3=2, stages={2, 3})),
                # round 6: one-term candidates that need no second MFMA chain
                ('as built, stage 4 only', dict(base, x1=True, x2=True, w2=1, w3=1, stages={3})),
                ('x2, w3 (conv2 stays bf16), stages 2-4 only', dict(base, x2=True, w3=1, stages={1, 2, 3})),
                ('x2, w3 (conv2 stays bf16), stages 3-4 only', dict(base, x2=True, w3=1, stages={2, 3})),
                ('x1, w2 (conv3 stays bf16), stages 3-4 only', dict(base, x1=True, w2=1, stages={2, 3})),
                ('x2 + two-term w3 (conv2 stays bf16), stages 2-4 only', dict(base, x2=True, w3=2, stages={1, 2, 3})),
                ('x2 + two-term w3 (conv2 stays bf16), stages 3-4 only', dict(base, x2=True, w3=2, stages={2, 3}))]
        print(f'# e4m3 noise budget of the bottleneck interiors: FPN level-0 error / signal rms ({a.views} views 480x640, seed {a.seed}; torch-CPU fp32 emulation)')
        print('| e4m3 tensors | rms error / rms | max error / max |')
        print('|---|---|---|')
        for name, md in rows:
            y = trunk(sd, img, md)
            print(f'| {name} | {float((y - ref).pow(2).mean().sqrt() / rms) * 100:.2f} % | {float((y - ref).abs().max() / ref.abs().max()) * 100:.2f} % |', flush=True)


if __name__ == '__main__':
    main()
