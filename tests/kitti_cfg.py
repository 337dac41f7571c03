"""The KITTI ImVoxelNet configuration (configs/imvoxelnet/imvoxelnet_kitti.py:1-65 of the reference) as
plain dicts, plus the synthetic camera of SURVEY.md section 8d.  Shared by tests, smoke and bench."""
import numpy as np


def kitti_model_cfg(n_voxels=(216, 248, 12), in_ch=64, out_ch=256):
    return dict(
        type='ImVoxelNet', pretrained=None,
        backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='pytorch'),
        neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=in_ch, num_outs=4),
        neck_3d=dict(type='KittiImVoxelNeck', in_channels=in_ch, out_channels=out_ch),
        bbox_head=dict(type='Anchor3DHead', num_classes=1, in_channels=out_ch, feat_channels=out_ch,
                       use_direction_classifier=True,
                       anchor_generator=dict(type='Anchor3DRangeGenerator',
                                             ranges=[[0, -39.68, -1.78, 69.12 - .32, 39.68 - .32, -1.78]],
                                             sizes=[[1.6, 3.9, 1.56]], rotations=[0, 1.57], reshape_out=True),
                       diff_rad_by_sin=True, bbox_coder=dict(type='DeltaXYZWLHRBBoxCoder'),
                       loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0)),
        n_voxels=n_voxels, voxel_size=(.32, .32, .32))


KITTI_TEST_CFG = dict(use_rotate_nms=True, nms_across_levels=False, nms_thr=0.01, score_thr=0.1, min_bbox_size=0,
                      nms_pre=100, max_num=50)


def kitti_meta(img_hw=(384, 1280), t=(0.0, 0.0, 0.0), box_type=None):
    K = np.array([[721.5377, 0, 609.5593, 0], [0, 721.5377, 172.854, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    E = np.array([[0, -1, 0, t[0]], [0, 0, -1, t[1]], [1, 0, 0, t[2]], [0, 0, 0, 1]], np.float32)
    m = dict(img_shape=(img_hw[0], img_hw[1], 3), ori_shape=(img_hw[0], img_hw[1], 3),
             lidar2img=dict(intrinsic=K, extrinsic=[E], origin=np.array([34.56, 0, -1], np.float32)))
    if box_type is not None:
        m['box_type_3d'] = box_type
    return m
