"""The BASELINE.json workloads as plain dicts: the reference's model configurations (configs/imvoxelnet/imvoxelnet_kitti.py:1-65
and its siblings, restated with the same keys and values) and the synthetic cameras of SURVEY.md section 8d.  Host-only data:
bench.py, __graft_entry__.smoke(), tools/ and tests/ build their models and img_metas from here (tests/kitti_cfg.py re-exports
it under its old name)."""
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


# ---------------------------------------------------------------------------------------------------------------
# The other BASELINE.json configurations (reference configs/imvoxelnet/*.py) with the synthetic cameras of SURVEY 8d.
def _resnet_fpn(cf):
    return dict(backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                              norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='pytorch'),
                neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=cf, num_outs=4))


def nuscenes_model_cfg(n_voxels=(312, 312, 12), dcn=True):
    """imvoxelnet_nuscenes.py:1-68 (ResNet-50 with DCNv2 in stages 3-4, as the reference; dcn=False: plain ResNet-50)."""
    trunk = _resnet_fpn(64)
    if dcn:
        trunk['backbone'].update(dcn=dict(type='DCNv2', deform_groups=1, fallback_on_stride=False),
                                 stage_with_dcn=(False, False, True, True))
    return dict(type='ImVoxelNet', pretrained=None, **trunk,
                neck_3d=dict(type='NuScenesImVoxelNeck', in_channels=64, out_channels=256),
                bbox_head=dict(type='Anchor3DHead', num_classes=1, in_channels=256, feat_channels=256, use_direction_classifier=True,
                               anchor_generator=dict(type='Anchor3DRangeGenerator', sizes=[[1.98, 4.67, 1.74]],
                                                     ranges=[[-49.92, -49.92, -1.0, 49.92 - .32 * 2, 49.92 - .32 * 2, -1.0]]),
                               diff_rad_by_sin=True, dir_offset=0.7854, dir_limit_offset=0, bbox_coder=dict(type='DeltaXYZWLHRBBoxCoder'),
                               loss_cls=dict(type='FocalLoss', use_sigmoid=True)),
                n_voxels=n_voxels, voxel_size=(.32, .32, .32))


NUSCENES_TEST_CFG = dict(use_rotate_nms=True, nms_across_levels=False, nms_pre=1000, nms_thr=0.2, score_thr=0.05,
                         min_bbox_size=0, max_num=500)


def scannet_fast_model_cfg():
    """imvoxelnet_scannet_fast.py:1-38."""
    return dict(type='ImVoxelNet', pretrained=None, **_resnet_fpn(256),
                neck_3d=dict(type='FastIndoorImVoxelNeck', in_channels=256, out_channels=128, n_blocks=[1, 1, 1]),
                bbox_head=dict(type='ScanNetImVoxelHeadV2', loss_bbox=dict(type='AxisAlignedIoULoss', loss_weight=1.0), n_classes=18,
                               n_channels=128, n_reg_outs=6, n_scales=3, limit=27, centerness_topk=18),
                voxel_size=(.16, .16, .16), n_voxels=(40, 40, 16))


SCANNET_FAST_TEST_CFG = dict(nms_pre=1000, iou_thr=.25, score_thr=.01)


def sunrgbd_fast_model_cfg():
    """imvoxelnet_sunrgbd_fast.py:1-38."""
    return dict(type='ImVoxelNet', pretrained=None, **_resnet_fpn(256),
                neck_3d=dict(type='FastIndoorImVoxelNeck', in_channels=256, out_channels=128, n_blocks=[1, 1, 1]),
                bbox_head=dict(type='SunRgbdImVoxelHeadV2', n_classes=10, n_channels=128, n_reg_outs=7, n_scales=3, limit=27,
                               centerness_topk=18),
                voxel_size=(.16, .16, .16), n_voxels=(40, 40, 16))


SUNRGBD_FAST_TEST_CFG = dict(nms_pre=1000, nms_thr=.15, use_rotate_nms=True, score_thr=.0)


def scannet_v1_model_cfg():
    """imvoxelnet_scannet.py:1-38 (Atlas U-Net neck, V1 head with n_convs=0)."""
    return dict(type='ImVoxelNet', pretrained=None, **_resnet_fpn(64),
                neck_3d=dict(type='ImVoxelNeck', channels=[64, 128, 256, 512], out_channels=64, down_layers=[1, 2, 3, 4],
                             up_layers=[3, 2, 1], conditional=False),
                bbox_head=dict(type='ScanNetImVoxelHead', loss_bbox=dict(type='AxisAlignedIoULoss', loss_weight=1.0), n_classes=18,
                               n_channels=64, n_convs=0, n_reg_outs=6),
                voxel_size=(.08, .08, .08), n_voxels=(80, 80, 32))


SCANNET_V1_TEST_CFG = dict(nms_pre=1000, iou_thr=.15, score_thr=.01)


def _look_at(eye, target, up=(0, 0, 1.0)):
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, up)
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    R = np.stack([r, d, f])
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = -R @ eye
    return E.astype(np.float32)


def indoor_meta(n_views, img_hw=(480, 640), origin=(0, 0, .5), radius=2.5, box_type=None):
    """V cameras on a circle of radius 2.5 m looking at the origin; K at the image scale."""
    K = np.array([[577.87, 0, 319.5, 0], [0, 577.87, 239.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    E = [_look_at((radius * np.cos(2 * np.pi * i / n_views), radius * np.sin(2 * np.pi * i / n_views), 1.2), (0, 0, .5))
         for i in range(n_views)]
    m = dict(img_shape=(img_hw[0], img_hw[1], 3), ori_shape=(img_hw[0], img_hw[1], 3),
             lidar2img=dict(intrinsic=K, extrinsic=E, origin=np.array(origin, np.float32)))
    if box_type is not None:
        m['box_type_3d'] = box_type
    return m


def nuscenes_meta(img_hw=(928, 1600), box_type=None):
    """6 cameras at yaw {0, +-55, +-110, 180} deg, f = 1266, K folded into the extrinsics (intrinsic = eye)."""
    K = np.array([[1266., 0, 800, 0], [0, 1266., 450, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)
    E = []
    for yaw in np.deg2rad([0, 55, 110, 180, -110, -55]):
        eye = np.array([0.5 * np.cos(yaw), 0.5 * np.sin(yaw), 0.6])
        E.append((K @ _look_at(eye, eye + np.array([np.cos(yaw), np.sin(yaw), 0.0])).astype(np.float64)).astype(np.float32))
    m = dict(img_shape=(img_hw[0], img_hw[1], 3), ori_shape=(img_hw[0], img_hw[1], 3),
             lidar2img=dict(intrinsic=np.eye(4, dtype=np.float32), extrinsic=E, origin=np.array([0, 0, -1.0], np.float32)))
    if box_type is not None:
        m['box_type_3d'] = box_type
    return m
