"""3-D necks on the MI355X conv kernel, under the reference's registry names and state-dict keys
(mmdet3d/models/necks/imvoxelnet.py).  forward(x) takes / returns the reference layout
([B,C,X,Y,Z] in; Kitti/NuScenes: one [B,256,Y',X'] out); forward_cl is the channels-last fast path the
detector chains without layout changes.
"""
from torch import nn

from . import ops
from .conv import FusedConv
from .params import ConvParams, BNParams, invalidate_packed_on_load
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
        invalidate_packed_on_load(self)

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


# ---------------------------------------------------------------------------------------------------
# Indoor necks
from .conv import FusedConvTranspose2x  # noqa: E402
from .params import ConvTransposeParams  # noqa: E402


class BasicBlock3dV2(nn.Module):
    """necks/imvoxelnet.py:233-260: conv3(stride)-BN-ReLU-conv3-BN, identity = 1x1x1(stride)-BN when strided."""

    def __init__(self, in_channels, out_channels, stride=1):
        super().__init__()
        self.stride = stride
        self.conv1 = ConvParams(in_channels, out_channels, 3)
        self.norm1 = BNParams(out_channels)
        self.conv2 = ConvParams(out_channels, out_channels, 3)
        self.norm2 = BNParams(out_channels)
        if stride != 1:
            self.downsample = nn.Sequential(ConvParams(in_channels, out_channels, 1), BNParams(out_channels))

    def prepare(self, device):
        self.f1 = FusedConv(self.conv1.weight, bn=self.norm1.tensors(), stride=self.stride, padding=1, relu=True).to(device)
        self.f2 = FusedConv(self.conv2.weight, bn=self.norm2.tensors(), padding=1, relu=True).to(device)
        self.fd = None
        if self.stride != 1:
            self.fd = FusedConv(self.downsample[0].weight, bn=self.downsample[1].tensors(), stride=self.stride).to(device)

    def forward_cl(self, x):
        idt = x if self.fd is None else self.fd(x)
        return self.f2(self.f1(x), res=idt)


@NECKS.register_module()
class FastIndoorImVoxelNeck(nn.Module):
    """necks/imvoxelnet.py:8-67.  Levels: down_layer_i (BasicBlock3dV2 stacks, stride 2 from level 1 on),
    up_block_i = ConvTranspose3d(k2,s2)-BN-ReLU-conv3-BN-ReLU, skip ADD, out_block_i = conv3-BN-ReLU."""

    def __init__(self, in_channels, n_blocks, out_channels):
        super().__init__()
        self.n_scales = len(n_blocks)
        c = in_channels
        for i in range(self.n_scales):
            stride = 1 if i == 0 else 2
            blocks = []
            for j in range(n_blocks[i]):
                if j == 0 and stride != 1:
                    blocks.append(BasicBlock3dV2(c, c * 2, stride))
                    c = c * 2
                else:
                    blocks.append(BasicBlock3dV2(c, c))
            setattr(self, f'down_layer_{i}', nn.Sequential(*blocks))
            if i > 0:   # Sequential indices match the reference: 0 convT, 1 BN, 2 ReLU, 3 conv, 4 BN, 5 ReLU
                setattr(self, f'up_block_{i}', nn.Sequential(ConvTransposeParams(c, c // 2, 2), BNParams(c // 2), nn.Identity(),
                                                             ConvParams(c // 2, c // 2, 3), BNParams(c // 2), nn.Identity()))
            setattr(self, f'out_block_{i}', nn.Sequential(ConvParams(c, out_channels, 3), BNParams(out_channels), nn.Identity()))
        self._device = None
        invalidate_packed_on_load(self)

    def init_weights(self):
        pass

    def prepare(self, device):
        self.fup, self.fout = {}, {}
        for i in range(self.n_scales):
            for blk in getattr(self, f'down_layer_{i}'):
                blk.prepare(device)
            if i > 0:
                u = getattr(self, f'up_block_{i}')
                self.fup[i] = (FusedConvTranspose2x(u[0].weight, bn=u[1].tensors(), relu=True).to(device),
                               FusedConv(u[3].weight, bn=u[4].tensors(), padding=1, relu=True).to(device))
            o = getattr(self, f'out_block_{i}')
            self.fout[i] = FusedConv(o[0].weight, bn=o[1].tensors(), padding=1, relu=True).to(device)
        self._device = device
        return self

    def forward_cl(self, x):
        if self._device is None:
            self.prepare(x.device)
        down = []
        for i in range(self.n_scales):
            for blk in getattr(self, f'down_layer_{i}'):
                x = blk.forward_cl(x)
            down.append(x)
        outs = []
        for i in range(self.n_scales - 1, -1, -1):
            if i < self.n_scales - 1:
                t, c3 = self.fup[i + 1]
                x = c3(t(x), res=down[i], res_after_act=True)        # relu(bn(conv(.))) + skip   (:30-31)
            outs.append(self.fout[i](x))
        return outs[::-1]

    def forward(self, x):
        return [ops.from_channels_last(o, 3) for o in self.forward_cl(ops.to_channels_last(x.contiguous()))]


class _ConditionalProjection(nn.Module):
    """necks/imvoxelnet.py:263-294 with condition=False (all reference configs): conv1x1x1-BN-ReLU of the skip."""

    def __init__(self, n):
        super().__init__()
        self.conv = ConvParams(n, n, 1)
        self.norm = BNParams(n)


class _EncoderDecoder(nn.Module):
    """Atlas 3-D U-Net, necks/imvoxelnet.py:297-372 (norm='BN', drop=0, cond_proj=False)."""

    def __init__(self, channels, layers_down, layers_up, cond_proj=False):
        super().__init__()
        if cond_proj:
            raise NotImplementedError('conditional projection (mask-dependent skip) is unused by the reference configs')
        self.channels = list(channels)
        self.layers_down = nn.ModuleList()
        proj = []
        self.layers_down.append(nn.Sequential(*[BasicBlock3d(channels[0], channels[0]) for _ in range(layers_down[0])]))
        proj.append(_ConditionalProjection(channels[0]))
        for i in range(1, len(channels)):
            # indices as in the reference Sequential: 0 conv(s2, no bias), 1 BN, 2 dropout, 3 ReLU, 4.. blocks
            layer = [ConvParams(channels[i - 1], channels[i], 3), BNParams(channels[i]), nn.Identity(), nn.Identity()]
            layer += [BasicBlock3d(channels[i], channels[i]) for _ in range(layers_down[i])]
            self.layers_down.append(nn.Sequential(*layer))
            if i < len(channels) - 1:
                proj.append(_ConditionalProjection(channels[i]))
        self.proj = nn.ModuleList(proj[::-1])
        rc = self.channels[::-1]
        self.layers_up_conv = nn.ModuleList([ConvParams(rc[i - 1], rc[i], 1) for i in range(1, len(rc))])
        self.layers_up_res = nn.ModuleList([nn.Sequential(*[BasicBlock3d(rc[i], rc[i]) for _ in range(layers_up[i - 1])])
                                            for i in range(1, len(rc))])

    def prepare(self, device):
        self.fdown = []
        for i, layer in enumerate(self.layers_down):
            f = None
            if i > 0:
                f = FusedConv(layer[0].weight, bn=layer[1].tensors(), stride=2, padding=1, relu=True).to(device)
            self.fdown.append(f)
            for m in layer:
                if isinstance(m, BasicBlock3d):
                    m.prepare(device)
        self.fup = [FusedConv(m.weight).to(device) for m in self.layers_up_conv]
        self.fproj = [FusedConv(m.conv.weight, bn=m.norm.tensors(), relu=True).to(device) for m in self.proj]
        for seq in self.layers_up_res:
            for m in seq:
                m.prepare(device)

    def forward_cl(self, x):
        xs = []
        for i, layer in enumerate(self.layers_down):
            if self.fdown[i] is not None:
                x = self.fdown[i](x)
            for m in layer:
                if isinstance(m, BasicBlock3d):
                    x = m.forward_cl(x)
            xs.append(x)
        xs = xs[::-1]
        out = []
        for i in range(len(self.fup)):
            x = self.fup[i](ops.upsample_trilinear2x(x))                          # :359-360
            x = self.fproj[i](xs[i + 1], res=x, res_after_act=True, post_scale=0.5)   # (x + relu(bn(conv(skip)))) / 2  :366-367
            for m in self.layers_up_res[i]:
                x = m.forward_cl(x)
            out.append(x)
        return out


@NECKS.register_module()
class ImVoxelNeck(nn.Module):
    """necks/imvoxelnet.py:70-91: Atlas EncoderDecoder + per-level conv3(bias)-BN-ReLU heads."""

    def __init__(self, channels, out_channels, down_layers, up_layers, conditional):
        super().__init__()
        self.model = _EncoderDecoder(channels, down_layers, up_layers, conditional)
        self.conv_blocks = nn.ModuleList([nn.Sequential(ConvParams(c, out_channels, 3, bias=True), BNParams(out_channels), nn.Identity())
                                          for c in channels])
        self._device = None
        invalidate_packed_on_load(self)

    def init_weights(self):
        pass

    def prepare(self, device):
        self.model.prepare(device)
        self.fblocks = [FusedConv(b[0].weight, b[0].bias, bn=b[1].tensors(), padding=1, relu=True).to(device) for b in self.conv_blocks]
        self._device = device
        return self

    def forward_cl(self, x):
        if self._device is None:
            self.prepare(x.device)
        xs = self.model.forward_cl(x)[::-1]
        return [self.fblocks[i](xs[i]) for i in range(len(xs))]

    def forward(self, x):
        return [ops.from_channels_last(o, 3) for o in self.forward_cl(ops.to_channels_last(x.contiguous()))]
