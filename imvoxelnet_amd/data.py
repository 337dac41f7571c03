"""Input side of the path: checkpoint loading, the test-time image pipeline and the
on-disk calibration -> `lidar2img` adapters.  Host-side numpy/torch; nothing here touches the kernels.

Reference: configs/imvoxelnet/imvoxelnet_kitti.py:66,94-105 (Resize keep_ratio -> Normalize -> Pad(size_divisor=32)),
mmdet3d/datasets/{kitti,nuscenes,scannet,sunrgbd}_monocular_dataset.py (lidar2img), pipelines/multi_view.py:45-53
(KittiSetOrigin, SunRgbdSetOrigin), pipelines/multi_view.py:7-31 (MultiViewPipeline view sampling).
mmcv / cv2 are not installed here: the resize restates cv2.resize(INTER_LINEAR) on uint8 images -- the 11-bit fixed-point
coefficient tables, the int32 horizontal pass, the `>> 4 ... >> 16 ... + 2 >> 2` vertical pass and the exact-half
INTER_AREA shortcut -- from OpenCV's published algorithm (imgproc/resize.cpp); it is checked against hand-derived vectors
(tests/test_host_cpu.py), not against cv2 itself: parity of the resize stays UNPINNED.
"""
import numpy as np
import torch
import torch.nn.functional as F

IMG_NORM_CFG = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)


# ------------------------------------------------------------------ checkpoints
def torch_load_trusted(filename, map_location='cpu', trusted=None):
    """torch.load for mmcv-style .pth files: `weights_only=True` first (tensors, dicts, strings and numbers -- what a released
    ImVoxelNet checkpoint holds -- load under it; nothing in the file is executed).  Files that carry other pickled objects
    (an optimizer's param-group classes, numpy scalars in 'meta') need the full unpickler, which can run arbitrary code from
    the file: that fallback is taken only when the caller vouches for the file -- `trusted=True`, or IVX_TRUST_CHECKPOINT=1
    in the environment (what mmcv.runner.load_checkpoint does unconditionally)."""
    import os
    import pickle
    try:
        return torch.load(filename, map_location=map_location, weights_only=True)
    except TypeError:           # torch < 1.13: no weights_only argument
        return torch.load(filename, map_location=map_location)
    except (pickle.UnpicklingError, RuntimeError) as e:
        if trusted is None:
            trusted = os.environ.get('IVX_TRUST_CHECKPOINT', '0') == '1'
        if not trusted:
            raise RuntimeError(f'{filename}: not loadable with weights_only=True ({str(e).splitlines()[0][:160]}); if you trust the '
                               'file, pass trusted=True (or set IVX_TRUST_CHECKPOINT=1) to unpickle it fully') from e
        return torch.load(filename, map_location=map_location, weights_only=False)


def load_checkpoint(model, filename, map_location='cpu', strict=False, trusted=None):
    """mmcv.runner.load_checkpoint for the released ImVoxelNet .pth files: a dict with 'state_dict' (and 'meta'),
    keys optionally prefixed with 'module.'.  Returns the checkpoint dict.  A model that was already prepared is re-packed for
    the device by its load_state_dict hook; otherwise call model.prepare(device) (or just run it).  trusted: see torch_load_trusted."""
    ckpt = torch_load_trusted(filename, map_location=map_location, trusted=trusted)
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


def _linear_tables(src, dst):
    """cv2 resize.cpp, INTER_LINEAR coefficient tables for one axis: source index and the two 11-bit weights per output
    position.  fx = (d + 0.5) * (src / dst) - 0.5 in FLOAT, sx = floor(fx); clamped at both borders; weights =
    round((1 - fx, fx) * 2048) as int16."""
    scale = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = (fx - sx.astype(np.float32)).astype(np.float32)
    lo = sx < 0
    sx[lo], fx[lo] = 0, 0.0
    hi = sx >= src - 1
    sx[hi], fx[hi] = src - 1, 0.0
    a1 = np.rint(fx * np.float32(2048.0)).astype(np.int64)           # cvRound: round half to even
    a0 = np.rint((np.float32(1.0) - fx) * np.float32(2048.0)).astype(np.int64)
    return sx, np.minimum(sx + 1, src - 1), a0, a1


def imresize_cv2_linear(img_u8, size_hw):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for an (H, W, C) uint8 image (what mmcv.imresize runs inside
    mmdet's Resize: imvoxelnet_kitti.py:98), in integer arithmetic:
      horizontal: D[x] = S[sx] * a0 + S[sx+1] * a1                                  (int32, weights sum to 2048)
      vertical:   dst  = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2
    An exact 2x down-scale on both axes takes OpenCV's INTER_AREA shortcut: (a + b + c + d + 2) >> 2."""
    img = np.ascontiguousarray(img_u8)
    if img.dtype != np.uint8 or img.ndim != 3:
        raise TypeError('imresize_cv2_linear expects an (H, W, C) uint8 image')
    h, w = img.shape[:2]
    nh, nw = int(size_hw[0]), int(size_hw[1])
    if (nh, nw) == (h, w):
        return img.copy()
    if h == 2 * nh and w == 2 * nw:
        q = img.astype(np.int64).reshape(nh, 2, nw, 2, img.shape[2])
        return ((q.sum(axis=(1, 3)) + 2) >> 2).astype(np.uint8)
    sx, sx1, a0, a1 = _linear_tables(w, nw)
    sy, sy1, b0, b1 = _linear_tables(h, nh)
    src = img.astype(np.int64)
    rows = src[:, sx] * a0[None, :, None] + src[:, sx1] * a1[None, :, None]        # [h, nw, C], up to 255 * 2048
    d0, d1 = rows[sy] >> 4, rows[sy1] >> 4
    out = (((b0[:, None, None] * d0) >> 16) + ((b1[:, None, None] * d1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def prepare_image(img_bgr_u8, img_scale, img_norm_cfg=IMG_NORM_CFG, size_divisor=32, keep_ratio=True):
    """LoadImageFromFile -> Resize -> Normalize -> Pad for ONE image (H,W,3 uint8, BGR as cv2 loads it).
    Returns (tensor [3,Hp,Wp] fp32, meta dict with img_shape / ori_shape / pad_shape as the mmdet pipeline sets them)."""
    ori_shape = tuple(img_bgr_u8.shape)
    if keep_ratio:
        nh, nw = rescale_size(ori_shape[:2], img_scale)
    else:
        nw, nh = img_scale
    resized = imresize_cv2_linear(img_bgr_u8, (nh, nw))                    # uint8 in, uint8 out, as mmcv.imresize
    img = torch.from_numpy(resized).float().permute(2, 0, 1)[None]
    mean = torch.tensor(img_norm_cfg['mean'], dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(img_norm_cfg['std'], dtype=torch.float32).view(1, 3, 1, 1)
    if img_norm_cfg.get('to_rgb', True):
        img = img[:, [2, 1, 0]]
    img = (img - mean) / std
    ph = (nh + size_divisor - 1) // size_divisor * size_divisor
    pw = (nw + size_divisor - 1) // size_divisor * size_divisor
    img = F.pad(img, (0, pw - nw, 0, ph - nh))
    return img[0], dict(img_shape=(nh, nw, 3), ori_shape=ori_shape, pad_shape=(ph, pw, 3))


# ------------------------------------------------------------------ multi-view pipeline
class MultiViewPipeline:
    """pipelines/multi_view.py:7-31: draw `n_images` of the scene's views with np.random.choice (with replacement only when
    the scene has fewer views), run the per-view transforms on each drawn view, and re-order the extrinsics to the drawn
    order.  `transforms`: one callable or a list of callables dict -> dict (mmdet's Compose of LoadImageFromFile / Resize /
    Normalize / Pad; view_transform() below is that chain for an in-memory image).  Seed numpy's global generator as
    the reference's training / test scripts do to reproduce a draw."""

    def __init__(self, transforms, n_images):
        self.transforms = list(transforms) if isinstance(transforms, (list, tuple)) else [transforms]
        self.n_images = n_images

    def _apply(self, res):
        for t in self.transforms:
            res = t(res)
        return res

    def __call__(self, results):
        ids = np.arange(len(results['img_info']))
        ids = np.random.choice(ids, self.n_images, replace=self.n_images > len(ids))
        imgs, extrinsics, last = [], [], None
        for i in ids.tolist():
            last = self._apply({key: results[key][i] for key in ('img_prefix', 'img_info')})
            imgs.append(last['img'])
            extrinsics.append(results['lidar2img']['extrinsic'][i])
        for key, val in last.items():          # per-view meta of the LAST drawn view becomes the sample's (img_shape, ...)
            if key not in ('img', 'img_prefix', 'img_info'):
                results[key] = val
        results['img'] = imgs
        results['lidar2img']['extrinsic'] = extrinsics
        return results


def view_transform(img_scale, img_norm_cfg=IMG_NORM_CFG, size_divisor=32, keep_ratio=True, loader=None):
    """The per-view transform chain of the reference test pipelines (imvoxelnet_kitti.py:94-105) for MultiViewPipeline:
    img_info['filename'] is loaded with `loader` (default: img_info['array'], an in-memory BGR uint8 image)."""
    def run(res):
        arr = loader(res['img_info']['filename']) if loader is not None else res['img_info']['array']
        img, meta = prepare_image(arr, img_scale, img_norm_cfg, size_divisor, keep_ratio)
        out = dict(res)
        out.update(meta)
        out['img'] = img
        return out
    return run


class KittiSetOrigin:
    """multi_view.py:45-53 (also the nuScenes configs): origin = centre of point_cloud_range, fp32."""

    def __init__(self, point_cloud_range):
        pcr = np.array(point_cloud_range, dtype=np.float32)
        self.origin = (pcr[:3] + pcr[3:]) / 2.

    def __call__(self, results):
        results['lidar2img']['origin'] = self.origin.copy()
        return results


class SunRgbdSetOrigin:
    """multi_view.py:84-94 (Total3D configs): the point 3 m along the ray through the image centre."""

    def __call__(self, results):
        l2i = results['lidar2img']
        projection = l2i['intrinsic'][:3, :3] @ l2i['extrinsic'][0][:3, :3]
        h, w, _ = results['ori_shape']
        centre = np.array([w / 2, h / 2, 1], dtype=np.float32)
        centre *= 3
        l2i['origin'] = np.linalg.inv(projection) @ centre
        return results


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
