#!/usr/bin/env python
"""How exact is v_mfma_f32_32x32x16_fp8_fp8's accumulation?  e4m3 x e4m3 products are exact in fp32, so an fp32-accumulating
kernel should match a sequential fp32 fma chain (the validation kernel) and a float64 reference to ~K * 2^-24 of the sum of
|products|.  Prints the worst deviations of the MFMA kernel and of the validation kernel from float64, in units of that bound."""
import os
import sys

import torch
import torch.nn.functional as F

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from imvoxelnet_amd.conv import FusedConv, QTensor, FP8, FP8_MAX  # noqa: E402

g = torch.Generator().manual_seed(0)
for cin, cout, hw in ((64, 256, (24, 40)), (512, 128, (12, 20)), (2048, 64, (6, 10))):
    x = torch.randn(2, cin, *hw, generator=g).abs() * 2
    w = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    s_x = float(x.abs().max()) / FP8_MAX
    xq = (x / s_x).to(FP8)
    fc = FusedConv(w, dims=2, dtype=FP8, out_dtype=torch.float32).to('cuda')
    wq = fc._w_host.float().reshape(cout, -1)[:, :cin] if fc.layout == 0 else None
    xin = QTensor(xq.view(torch.uint8).unsqueeze(2).permute(0, 2, 3, 4, 1).contiguous().cuda().view(FP8), s_x)
    y = fc(xin).float().cpu()[:, 0].permute(0, 3, 1, 2)
    yn = fc(xin, naive=True).float().cpu()[:, 0].permute(0, 3, 1, 2)
    # float64 reference on the raw e4m3 values, scaled like the kernel does
    wraw = (w / fc.w_scale.view(-1, 1, 1, 1)).to(FP8).double()
    acc = F.conv2d(xq.double(), wraw)
    mag = F.conv2d(xq.double().abs(), wraw.abs())
    sc = (fc.w_scale.double() * s_x).view(1, -1, 1, 1)
    ref = acc * sc
    bound = mag * sc * 2.0 ** -24
    for nm, t in (('mfma', y), ('naive', yn)):
        d = (t.double() - ref).abs()
        print(f'Cin {cin:5d}: {nm:5s} max |err| / (sum|products| * 2^-24) = {float((d / bound).max()):8.2f}   max |err| / max |ref| = {float(d.max() / ref.abs().max()):.2e}')
