"""Box containers and utilities with the reference's names and semantics
(mmdet3d/core/bbox/structures/{base_box3d,lidar_box3d,depth_box3d,utils}.py,
mmdet3d/core/bbox/transforms.py:49-67).  Tiny host/elementwise helpers: plain torch tensor ops.
"""
import numpy as np
import torch


def limit_period(val, offset=0.5, period=np.pi):
    """structures/utils.py:5-18."""
    return val - torch.floor(val / period + offset) * period


def xywhr2xyxyr(boxes_xywhr):
    """structures/utils.py:64-82: (x, y, w, h, r) -> (x - w/2, y - h/2, x + w/2, y + h/2, r)."""
    out = torch.zeros_like(boxes_xywhr)
    hw = boxes_xywhr[:, 2] / 2
    hh = boxes_xywhr[:, 3] / 2
    out[:, 0] = boxes_xywhr[:, 0] - hw
    out[:, 1] = boxes_xywhr[:, 1] - hh
    out[:, 2] = boxes_xywhr[:, 0] + hw
    out[:, 3] = boxes_xywhr[:, 1] + hh
    out[:, 4] = boxes_xywhr[:, 4]
    return out


def rotation_3d_in_axis(points, angles, axis=0):
    """structures/utils.py:21-61: rotate points [N,M,3] by angles [N] about `axis`."""
    s, c = torch.sin(angles), torch.cos(angles)
    one, zero = torch.ones_like(c), torch.zeros_like(c)
    if axis == 1:
        rows = [[c, zero, -s], [zero, one, zero], [s, zero, c]]
    elif axis in (2, -1):
        rows = [[c, -s, zero], [s, c, zero], [zero, zero, one]]
    elif axis == 0:
        rows = [[zero, c, -s], [zero, s, c], [one, zero, zero]]
    else:
        raise ValueError(f'axis should in range [0, 1, 2], got {axis}')
    rot_t = torch.stack([torch.stack(r) for r in rows])       # [3,3,N]
    return torch.einsum('aij,jka->aik', points, rot_t)


class BaseInstance3DBoxes:
    """base_box3d.py:37-66: N x box_dim tensor, bottom-centre origin (0.5, 0.5, 0)."""

    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device('cpu')
        tensor = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, box_dim)).to(dtype=torch.float32, device=device)
        assert tensor.dim() == 2 and tensor.size(-1) == box_dim, tensor.size()
        if tensor.shape[-1] == 6:
            assert box_dim == 6
            tensor = torch.cat((tensor, tensor.new_zeros(tensor.shape[0], 1)), dim=-1)
            self.box_dim = box_dim + 1
            self.with_yaw = False
        else:
            self.box_dim = box_dim
            self.with_yaw = with_yaw
        self.tensor = tensor.clone()
        if tuple(origin) != (0.5, 0.5, 0):
            dst = self.tensor.new_tensor((0.5, 0.5, 0))
            src = self.tensor.new_tensor(origin)
            self.tensor[:, :3] += self.tensor[:, 3:6] * (dst - src)

    volume = property(lambda self: self.tensor[:, 3] * self.tensor[:, 4] * self.tensor[:, 5])
    dims = property(lambda self: self.tensor[:, 3:6])
    yaw = property(lambda self: self.tensor[:, 6])
    height = property(lambda self: self.tensor[:, 5])
    bottom_center = property(lambda self: self.tensor[:, :3])
    center = bottom_center
    bev = property(lambda self: self.tensor[:, [0, 1, 3, 4, 6]])

    @property
    def gravity_center(self):
        bc = self.bottom_center
        gc = torch.zeros_like(bc)
        gc[:, :2] = bc[:, :2]
        gc[:, 2] = bc[:, 2] + self.tensor[:, 5] * 0.5
        return gc

    top_height = property(lambda self: self.tensor[:, 2] + self.tensor[:, 5])
    bottom_height = property(lambda self: self.tensor[:, 2])

    def new_box(self, data):
        t = self.tensor.new_tensor(data) if not isinstance(data, torch.Tensor) else data.to(self.tensor.device)
        return type(self)(t, box_dim=self.box_dim, with_yaw=self.with_yaw)

    @classmethod
    def height_overlaps(cls, boxes1, boxes2, mode='iou'):
        """base_box3d.py:352-381."""
        lowest_top = torch.min(boxes1.top_height.view(-1, 1), boxes2.top_height.view(1, -1))
        highest_bottom = torch.max(boxes1.bottom_height.view(-1, 1), boxes2.bottom_height.view(1, -1))
        return torch.clamp(lowest_top - highest_bottom, min=0)

    # hook so CPU-only tests can substitute the oracle; the product path is the device kernel
    _bev_overlap_fn = None

    @classmethod
    def overlaps(cls, boxes1, boxes2, mode='iou'):
        """base_box3d.py:383-443: 3-D IoU / IoF = rotated-BEV overlap x height overlap / (v1 + v2 - overlap).  Like the
        reference, the BEV overlap runs on the device whatever device the boxes live on."""
        assert type(boxes1) == type(boxes2), f'boxes of different types: {type(boxes1)} and {type(boxes2)}'
        assert mode in ('iou', 'iof')
        rows, cols = len(boxes1), len(boxes2)
        if rows * cols == 0:
            return boxes1.tensor.new(rows, cols)
        overlaps_h = cls.height_overlaps(boxes1, boxes2)
        b1, b2 = xywhr2xyxyr(boxes1.bev), xywhr2xyxyr(boxes2.bev)
        if cls._bev_overlap_fn is not None:
            overlaps_bev = cls._bev_overlap_fn(b1, b2)
        else:
            from .nms import boxes_overlap_bev
            overlaps_bev = boxes_overlap_bev(b1.contiguous().cuda(), b2.contiguous().cuda())
        overlaps_3d = overlaps_bev.to(boxes1.tensor.device) * overlaps_h
        v1, v2 = boxes1.volume.view(-1, 1), boxes2.volume.view(1, -1)
        if mode == 'iou':
            return overlaps_3d / torch.clamp(v1 + v2 - overlaps_3d, min=1e-8)
        return overlaps_3d / torch.clamp(v1, min=1e-8)

    def to(self, device):
        return type(self)(self.tensor.to(device), box_dim=self.box_dim, with_yaw=self.with_yaw)

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        t = self.tensor[item]
        if t.dim() == 1:
            t = t.view(1, -1)
        return type(self)(t, box_dim=self.box_dim, with_yaw=self.with_yaw)

    def __repr__(self):
        return self.__class__.__name__ + '(\n    ' + str(self.tensor) + ')'


class LiDARInstance3DBoxes(BaseInstance3DBoxes):
    """lidar_box3d.py (x front, y left, z up; yaw about z)."""


class DepthInstance3DBoxes(BaseInstance3DBoxes):
    """depth_box3d.py (x right, y front, z up; yaw about z)."""


def get_box_type(box_type):
    t = box_type.lower()
    if t == 'lidar':
        return LiDARInstance3DBoxes
    if t == 'depth':
        return DepthInstance3DBoxes
    raise ValueError(f'box type {box_type} is not built (LiDAR | Depth)')


def bbox3d2result(bboxes, scores, labels):
    """core/bbox/transforms.py:49-67."""
    return dict(boxes_3d=bboxes.to('cpu'), scores_3d=scores.cpu(), labels_3d=labels.cpu())
