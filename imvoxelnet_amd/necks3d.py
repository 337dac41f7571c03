"""3-D necks on the MI355X conv kernel, under the reference's registry names and state-dict keys
(mmdet3d/models/necks/imvoxelnet.py).  forward(x) takes / returns the reference layout
([B,C,X,Y,Z] in; Kitti/NuScenes: one [B,256,Y',X'] out); forward_cl is the channels-last fast path the
detector chains without layout changes.
"""
import torch
from torch import nn

from . import ops
from .conv import FusedConv
from .params import ConvParams, BNParams
from .registry import NECKS


class BasicBlock3d(nn.Module):
    """conv3x3x3-BN-ReLU-conv3x3x3-BN-(+x)-ReLU, bias-free convs (necks/imvoxelnet.py:191-230)."""

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        if stride != 1 or inplanes != planes:
            raise NotImplementedError('the ImVoxelNet necks only use stride-1, same-width BasicBlock3d')
        self.conv1 = ConvParams(inplanes, planes, 3)
        self.bn1 = BNParams(planes)
        self.conv2 = ConvParams(planes, planes, 3)
        self.bn2 = BNParams(planes)

    def prepare(self, device):
        self.f1 = FusedConv(self.conv1.weight, bn=self.bn1.tensors(), padding=1, relu=True).to(device)
        self.f2 = FusedConv(self.conv2.weight, bn=self.bn2.tensors(), padding=1, relu=True).to(device)

    def forward_cl(self, x):
        return self.f2(self.f1(x), res=x)


def _conv_bn_relu_params(cin, cout):
    # nn.Sequential(Conv3d(bias=True), BatchNorm3d, ReLU): parameters live at indices 0 and 1
    return nn.Sequential(ConvParams(cin, cout, 3, bias=True), BNParams(cout))


class _StackNeck(nn.Module):
    """Shared body of KittiImVoxelNeck / NuScenesImVoxelNeck: block, conv, block, conv, block, conv."""
    strides = None
    paddings = None

    def __init__(self, in_channels, out_channels):
        super().__init__()
        c = in_channels
        self.model = nn.Sequential(
            BasicBlock3d(c, c), _conv_bn_relu_params(c, c * 2),
            BasicBlock3d(c * 2, c * 2), _conv_bn_relu_params(c * 2, c * 4),
            BasicBlock3d(c * 4, c * 4), _conv_bn_relu_params(c * 4, out_channels))
        self._device = None

    def init_weights(self):
        pass

    def prepare(self, device):
        self.fconv = []
        for i, m in enumerate(self.model):
            if isinstance(m, BasicBlock3d):
                m.prepare(device)
            else:
                k = len(self.fconv)
                self.fconv.append(FusedConv(m[0].weight, m[0].bias, bn=m[1].tensors(), stride=self.strides[k],
                                            padding=self.paddings[k], relu=True).to(device))
        self._device = device
        return self

    def forward_cl(self, x):
        """x [B,X,Y,Z,C] -> [B,X',Y',1,Cout]."""
        if self._device is None:
            self.prepare(x.device)
        k = 0
        for m in self.model:
            if isinstance(m, BasicBlock3d):
                x = m.forward_cl(x)
            else:
                x = self.fconv[k](x)
                k += 1
        if x.shape[3] != 1:
            raise AssertionError(f'the z axis must collapse to 1 (got {x.shape[3]}); necks/imvoxelnet.py:119,150')
        return x

    def forward(self, x):
        y = ops.from_channels_last(self.forward_cl(ops.to_channels_last(x.contiguous())), 3)
        return [y[..., 0].transpose(-1, -2)]


@NECKS.register_module()
class KittiImVoxelNeck(_StackNeck):
    """necks/imvoxelnet.py:94-123: down-convs stride (1,1,2) pad 1; last conv k3 s1 p0 (all axes)."""
    strides = [(1, 1, 2), (1, 1, 2), (1, 1, 1)]
    paddings = [(1, 1, 1), (1, 1, 1), (0, 0, 0)]


@NECKS.register_module()
class NuScenesImVoxelNeck(_StackNeck):
    """necks/imvoxelnet.py:126-154: first down-conv stride 2 on every axis, last conv pad (1,1,0)."""
    strides = [(2, 2, 2), (1, 1, 2), (1, 1, 1)]
    paddings = [(1, 1, 1), (1, 1, 1), (1, 1, 0)]
