import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, imvoxelnet_amd as ia, kitti_cfg as kc
m = ia.build_detector(kc.kitti_model_cfg(), test_cfg=kc.KITTI_TEST_CFG); ia.randomize_(m, 0); m.prepare(torch.device('cuda'))
L = m._native.L
print('KITTI batch 4 workspace bytes', L.ivx_model_workspace_bytes(m._native.h, 4, 1, 384, 1280), 'trunk', L.ivx_backbone_fpn_workspace_bytes(m._native.h, 4, 384, 1280), 'neck', L.ivx_neck3d_workspace_bytes(m._native.h, 4))
