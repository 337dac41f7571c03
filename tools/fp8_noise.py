import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import imvoxelnet_amd as ia
from imvoxelnet_amd import workloads as kc
model = ia.build_detector(kc.scannet_v1_model_cfg(), test_cfg=dict(kc.SCANNET_V1_TEST_CFG))
ia.randomize_(model, 41)
img = torch.randn(1, 4, 3, 480, 640, generator=torch.Generator().manual_seed(2)).cuda()
model.prepare(torch.device('cuda'), dtype=torch.float32)
p32 = model.features_2d_cl(img).float()
model.prepare(torch.device('cuda'), dtype=torch.bfloat16)
pbf = model.features_2d_cl(img).float()
def rel(a, b): return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
print('bf16 vs fp32: %.4f of the signal rms' % rel(pbf, p32))
for st in (None, 3, 2, 1):
    model.prepare(torch.device('cuda'), dtype=torch.bfloat16)
    model.calibrate_fp8(img, stages=st)
    p8 = model.features_2d_cl(img).float()
    print('fp8 stages', st, ': vs bf16 %.4f, vs fp32 %.4f of the signal rms; max err %.4f of max' % (rel(p8, pbf), rel(p8, p32), float((p8 - p32).abs().max() / p32.abs().max())))
