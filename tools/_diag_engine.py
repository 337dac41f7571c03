import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import imvoxelnet_amd as ia
import kitti_cfg as kc
from imvoxelnet_amd import ops

model = ia.build_detector(kc.kitti_model_cfg(n_voxels=(104, 120, 12)), test_cfg=kc.KITTI_TEST_CFG)
ia.randomize_(model, 21)
model.prepare(torch.device('cuda'))
hw, B = (192, 640), 2
img = torch.randn(B, 1, 3, *hw, generator=torch.Generator().manual_seed(3)).cuda()
metas = [kc.kitti_meta(img_hw=hw, t=(0.02 * b, 0.01 * b, 0.0), box_type=ia.LiDARInstance3DBoxes) for b in range(B)]
x = img.reshape(B, 3, *hw).contiguous()
nat = model._native
def d(a, b): return f'equal={torch.equal(a,b)} max|d|={(a-b).abs().max().item():.3e} ndiff={(a!=b).sum().item()} of {a.numel()}'
p0a = model.features_2d_cl(img); p0b = model.features_2d_cl(img)
print('python trunk twice :', d(p0a, p0b))
e0a = nat.backbone_fpn(x).clone(); e0b = nat.backbone_fpn(x).clone()
print('engine trunk twice :', d(e0a, e0b))
print('python vs engine trunk:', d(p0a, e0a))
vol, valid = model.lift_cl(p0a, metas)
ya = model.neck_3d.forward_cl(vol); yb = model.neck_3d.forward_cl(vol)
print('python neck twice  :', d(ya, yb))
na = nat.neck3d(vol).clone(); nb = nat.neck3d(vol).clone()
print('engine neck twice  :', d(na, nb))
print('python vs engine neck:', d(ya, na))
# per-layer of the trunk on the python side with wide vs narrow epilogue
from imvoxelnet_amd import _lib
L = _lib.lib()
L.ivx_conv_set_epilogue_mode(1)
p0n = model.features_2d_cl(img)
yn = model.neck_3d.forward_cl(vol)
L.ivx_conv_set_epilogue_mode(0)
print('python trunk wide vs narrow epilogue:', d(p0a, p0n))
print('python neck wide vs narrow epilogue :', d(ya, yn))
# stage-wise: stem, stage outputs
xcl = ops.to_channels_last(x, pad_to=4)
f = model.backbone.forward_cl(xcl)
for i, t in enumerate(f): print('C%d' % (i + 2), tuple(t.shape), float(t.abs().max()))
