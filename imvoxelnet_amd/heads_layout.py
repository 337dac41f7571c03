"""LayoutHead -- the 2-D head of the SUN RGB-D "Total" configs (mmdet3d/models/dense_heads/layout_head.py:8-116,
configs/imvoxelnet/imvoxelnet_total_sunrgbd_fast.py:13-19): global average pool of C5, two 3-layer MLPs that predict the
camera (pitch, roll) and the room layout box.  At test time the predicted angles replace the dataset extrinsics in the
unprojection (detectors/imvoxelnet.py:59-61,121-124,164-187).

The pool and the six Linear layers run on the device (ivx_global_avgpool_fwd, 1x1 ivx_conv_fwd with bias + ReLU in
the epilogue; Dropout is the identity in eval mode); the final limit_period / exp on 9 numbers per sample is done on
the host after the one small D2H the reference also needs (its extrinsics are built from `angles` on the CPU).
"""
import torch
from torch import nn

from . import ops
from .conv import FusedConv
from .params import invalidate_packed_on_load
from .registry import HEADS


class LinearParams(nn.Module):
    """Holds `weight` [out,in] and `bias` [out] like nn.Linear."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)        # nn.Linear's default

    def forward(self, *a, **k):
        raise RuntimeError('LinearParams is a parameter container; the layer runs in libimvoxel_hip.so')


def _mlp_params(cin, hidden, cout):
    # nn.Sequential(Linear, ReLU, Dropout, Linear, ReLU, Dropout, Linear): parameters live at indices 0, 3, 6
    return nn.Sequential(LinearParams(cin, hidden), nn.Identity(), nn.Identity(), LinearParams(hidden, hidden), nn.Identity(),
                         nn.Identity(), LinearParams(hidden, cout))


@HEADS.register_module()
class LayoutHead(nn.Module):
    def __init__(self, n_channels, linear_size, dropout=0.0, loss_angle=None, loss_layout=None, **kwargs):
        super().__init__()
        self.angle_mlp = _mlp_params(n_channels, linear_size, 2)
        self.layout_mlp = _mlp_params(n_channels, linear_size, 7)
        self._device = None
        invalidate_packed_on_load(self)

    def init_weights(self):
        pass

    def prepare(self, device):
        def mlp(seq):
            return [FusedConv(seq[i].weight[:, :, None, None], seq[i].bias, relu=(i != 6), dims=2, dtype=torch.float32).to(device)
                    for i in (0, 3, 6)]
        self.f_angle, self.f_layout = mlp(self.angle_mlp), mlp(self.layout_mlp)
        self._device = device
        return self

    def forward_cl(self, c5, img_metas):
        """c5 [B,1,h,w,C] channels-last -> (angles: list of [2] CPU tensors, layouts: list of [7] CPU tensors)."""
        if self._device is None:
            self.prepare(c5.device)
        x = ops.global_avgpool(c5 if c5.dtype == torch.float32 else c5.float())
        a = l_ = x
        for f in self.f_angle:
            a = f(a)
        for f in self.f_layout:
            l_ = f(l_)
        both = torch.cat([a.reshape(a.shape[0], -1), l_.reshape(l_.shape[0], -1)], 1).cpu()      # [B, 2 + 7]
        angles, layouts = [], []
        for row in both:
            angle, layout = self._forward_single(row[:2], row[2:], None)
            angles.append(angle)
            layouts.append(layout)
        return angles, layouts

    def forward(self, x, img_metas):
        """x [B,C,h,w] (reference layout), as layout_head.py:41-50."""
        return self.forward_cl(ops.to_channels_last(x.contiguous()), img_metas)

    @staticmethod
    def _forward_single(angle, layout, img_meta):
        """layout_head.py:52-74: angle = limit_period(angle); layout = (centre, exp(size), yaw) -- through the library's host
        function (ivx_layout_head_decode: fixed fp32 order, libm expf), so this host and the native model handle return the same
        bits; the reference's torch ops (limit_period, torch.exp) agree with it to 1 ulp."""
        import ctypes as C
        from . import _lib
        a = angle.detach().to('cpu', torch.float32).contiguous()
        l_ = layout.detach().to('cpu', torch.float32).contiguous()
        oa, ol = torch.empty(2), torch.empty(7)
        _lib.check(_lib.lib().ivx_layout_head_decode(C.c_void_p(a.data_ptr()), C.c_void_p(l_.data_ptr()), C.c_void_p(oa.data_ptr()),
                                                     C.c_void_p(ol.data_ptr())), 'ivx_layout_head_decode')
        return oa, ol

    def get_bboxes(self, angles, layouts, img_metas):
        """layout_head.py:106-115: -> (angles, layout boxes of the sample's box type, gravity-centre origin)."""
        out_a, out_l = [], []
        for angle, layout, meta in zip(angles, layouts, img_metas):
            out_a.append(angle.cpu())
            out_l.append(meta['box_type_3d'](layout.unsqueeze(0), origin=(.5, .5, .5)))
        return out_a, out_l


def layout_extrinsics(angles):
    """get_extrinsics through the library's host function (ivx_layout_extrinsics: the same products in a fixed order, libm cosf / sinf):
    what the detector feeds the unprojection, on both hosts of the library.  Equal to get_extrinsics (the reference's torch ops) to 1 ulp
    of the four trigonometric values."""
    import ctypes as C
    from . import _lib
    a = angles.detach().to('cpu', torch.float32).contiguous()
    e = torch.empty(4, 4)
    _lib.check(_lib.lib().ivx_layout_extrinsics(C.c_void_p(a.data_ptr()), C.c_void_p(e.data_ptr())), 'ivx_layout_extrinsics')
    return e


def get_extrinsics(angles):
    """detectors/imvoxelnet.py:164-187: camera extrinsic [4,4] from predicted (pitch, roll), yaw = 0; same torch CPU
    ops in the same order."""
    yaw = angles.new_zeros(())
    pitch, roll = angles
    r = angles.new_zeros((3, 3))
    r[0, 0] = torch.cos(yaw) * torch.cos(pitch)
    r[0, 1] = torch.sin(yaw) * torch.sin(roll) - torch.cos(yaw) * torch.cos(roll) * torch.sin(pitch)
    r[0, 2] = torch.cos(roll) * torch.sin(yaw) + torch.cos(yaw) * torch.sin(pitch) * torch.sin(roll)
    r[1, 0] = torch.sin(pitch)
    r[1, 1] = torch.cos(pitch) * torch.cos(roll)
    r[1, 2] = -torch.cos(pitch) * torch.sin(roll)
    r[2, 0] = -torch.cos(pitch) * torch.sin(yaw)
    r[2, 1] = torch.cos(yaw) * torch.sin(roll) + torch.cos(roll) * torch.sin(yaw) * torch.sin(pitch)
    r[2, 2] = torch.cos(yaw) * torch.cos(roll) - torch.sin(yaw) * torch.sin(pitch) * torch.sin(roll)
    t = angles.new_tensor([[0., 0., 1.], [0., -1., 0.], [-1., 0., 0.]])      # Total3DUnderstanding axes
    r = t @ r.T
    r = r[:, [2, 0, 1]]                                                       # DepthInstance3DBoxes axes
    r[2] *= -1
    extrinsic = angles.new_zeros((4, 4))
    extrinsic[:3, :3] = r
    extrinsic[3, 3] = 1.
    return extrinsic
