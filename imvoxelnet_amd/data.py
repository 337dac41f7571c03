"""Input side of the path: checkpoint loading, the test-time image pipeline and the
on-disk calibration -> `lidar2img` adapters.  Host-side numpy/torch; nothing here touches the kernels.

Reference: configs/imvoxelnet/imvoxelnet_kitti.py:66,94-105 (Resize keep_ratio -> Normalize -> Pad(size_divisor=32)),
mmdet3d/datasets/{kitti,nuscenes,scannet,sunrgbd}_monocular_dataset.py (lidar2img), pipelines/multi_view.py:45-53
(KittiSetOrigin).  mmcv/cv2 are not available here: the resize is torch bilinear (half-pixel centres, no anti-alias),
which matches cv2.INTER_LINEAR up to its fixed-point rounding -- parity of the resize is UNPINNED.
"""
import numpy as np
import torch
import torch.nn.functional as F

IMG_NORM_CFG = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)


# ------------------------------------------------------------------ checkpoints
def torch_load_trusted(filename, map_location='cpu'):
    """torch.load for mmcv-style .pth files.  They are pickled dicts with 'meta' (strings, config text) and possibly
    optimizer state next to 'state_dict', which torch >= 2.6's default weights_only=True refuses with an opaque
    unpickling error; the caller vouches for the file, as mmcv.runner.load_checkpoint does."""
    try:
        return torch.load(filename, map_location=map_location, weights_only=False)
    except TypeError:           # torch < 1.13: no weights_only argument
        return torch.load(filename, map_location=map_location)


def load_checkpoint(model, filename, map_location='cpu', strict=False):
    """mmcv.runner.load_checkpoint for the released ImVoxelNet .pth files: a dict with 'state_dict' (and 'meta'),
    keys optionally prefixed with 'module.'.  Returns the checkpoint dict.  A model that was already prepared is re-packed for
    the device by its load_state_dict hook; otherwise call model.prepare(device) (or just run it)."""
    ckpt = torch_load_trusted(filename, map_location=map_location)
    sd = ckpt.get('state_dict', ckpt) if isinstance(ckpt, dict) else ckpt
    sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
    res = model.load_state_dict(sd, strict=False)
    missing = [k for k in res.missing_keys if not k.endswith('num_batches_tracked')]
    if strict and (missing or res.unexpected_keys):
        raise RuntimeError(f'checkpoint mismatch: missing {missing[:8]}, unexpected {list(res.unexpected_keys)[:8]}')
    if isinstance(ckpt, dict) and 'meta' in ckpt and 'CLASSES' in ckpt['meta']:
        model.CLASSES = ckpt['meta']['CLASSES']          # tools/test.py:122-125
    ckpt['_missing_keys'], ckpt['_unexpected_keys'] = missing, list(res.unexpected_keys)
    return ckpt


# ------------------------------------------------------------------ image pipeline
def rescale_size(old_hw, scale):
    """mmcv.rescale_size for a (w, h) tuple scale: largest size keeping the aspect ratio inside `scale`."""
    h, w = old_hw
    max_long, max_short = max(scale), min(scale)
    f = min(max_long / max(h, w), max_short / min(h, w))
    return int(h * float(f) + 0.5), int(w * float(f) + 0.5)


def prepare_image(img_bgr_u8, img_scale, img_norm_cfg=IMG_NORM_CFG, size_divisor=32, keep_ratio=True):
    """LoadImageFromFile -> Resize -> Normalize -> Pad for ONE image (H,W,3 uint8, BGR as cv2 loads it).
    Returns (tensor [3,Hp,Wp] fp32, meta dict with img_shape / ori_shape / pad_shape as the mmdet pipeline sets them)."""
    img = torch.from_numpy(np.ascontiguousarray(img_bgr_u8)).float().permute(2, 0, 1)[None]
    ori_shape = tuple(img_bgr_u8.shape)
    if keep_ratio:
        nh, nw = rescale_size(ori_shape[:2], img_scale)
    else:
        nw, nh = img_scale
    if (nh, nw) != ori_shape[:2]:
        img = F.interpolate(img, size=(nh, nw), mode='bilinear', align_corners=False)
    mean = torch.tensor(img_norm_cfg['mean'], dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(img_norm_cfg['std'], dtype=torch.float32).view(1, 3, 1, 1)
    if img_norm_cfg.get('to_rgb', True):
        img = img[:, [2, 1, 0]]
    img = (img - mean) / std
    ph = (nh + size_divisor - 1) // size_divisor * size_divisor
    pw = (nw + size_divisor - 1) // size_divisor * size_divisor
    img = F.pad(img, (0, pw - nw, 0, ph - nh))
    return img[0], dict(img_shape=(nh, nw, 3), ori_shape=ori_shape, pad_shape=(ph, pw, 3))


# ------------------------------------------------------------------ calibration -> lidar2img
def kitti_lidar2img(P2, R0_rect, Tr_velo_to_cam, point_cloud_range=(0, -39.68, -3, 69.12, 39.68, 1)):
    """kitti_monocular_dataset.py:16-22 + KittiSetOrigin (multi_view.py:45-53).  4x4 calibration matrices."""
    rect, trv2c, p2 = (np.asarray(m).astype(np.float32) for m in (R0_rect, Tr_velo_to_cam, P2))
    extrinsic = rect @ trv2c
    extrinsic[:3, 3] += np.linalg.inv(p2[:3, :3]) @ p2[:3, 3]
    intrinsic = np.copy(p2)
    intrinsic[:3, 3] = 0
    pcr = np.array(point_cloud_range, dtype=np.float32)
    return dict(extrinsic=[extrinsic], intrinsic=intrinsic, origin=(pcr[:3] + pcr[3:]) / 2.)


def nuscenes_lidar2img(lidar2img_per_camera, point_cloud_range=(-49.92, -49.92, -2.92, 49.92, 49.92, 0.92)):
    """nuscenes_monocular_dataset.py:15-22: K is already folded into every camera's lidar2img, intrinsic = eye(4)."""
    pcr = np.array(point_cloud_range, dtype=np.float32)
    return dict(extrinsic=[np.asarray(x).astype(np.float32) for x in lidar2img_per_camera], intrinsic=np.eye(4, dtype=np.float32),
                origin=(pcr[:3] + pcr[3:]) / 2.)


def scannet_lidar2img(axis_align_matrix, camera_extrinsics, intrinsics):
    """scannet_monocular_dataset.py:19-32: extrinsic_v = inv(axis_align @ cam2world_v), origin (0, 0, .5)."""
    aam = np.asarray(axis_align_matrix).astype(np.float32)
    ext = [np.linalg.inv(aam @ np.asarray(e)).astype(np.float32) for e in camera_extrinsics]
    return dict(extrinsic=ext, intrinsic=np.asarray(intrinsics).astype(np.float32), origin=np.array([.0, .0, .5], np.float32))


def sunrgbd_lidar2img(K, Rt):
    """sunrgbd_monocular_dataset.py:29-73: K stored column-major (reshape(3,3).T), Rt with y/z swapped and y negated,
    extrinsic rotation = Rt^T, origin (0, 3, -1)."""
    intrinsic = np.eye(4)
    intrinsic[:3, :3] = np.asarray(K).copy().reshape(3, 3).T
    rt = np.asarray(Rt).copy()
    rt[:, [1, 2]] = rt[:, [2, 1]]
    rt[:, 1] = -1 * rt[:, 1]
    extrinsic = np.eye(4)
    extrinsic[:3, :3] = rt.T
    return dict(extrinsic=[extrinsic.astype(np.float32)], intrinsic=intrinsic.astype(np.float32), origin=np.array([0, 3, -1], np.float32))
