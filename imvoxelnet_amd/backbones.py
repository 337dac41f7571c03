"""2-D trunk on the MI355X conv kernel: ResNet (depth 50, style='pytorch', eval BN) and FPN.

Replaces the mmdet 2.10.0 modules the reference builds at mmdet3d/models/detectors/imvoxelnet.py:22-23
and calls at :48,:50 (configs/imvoxelnet/imvoxelnet_kitti.py:4-17).  Parameter names follow
torchvision / mmdet so `torchvision://resnet50` and released ImVoxelNet checkpoints load.
Parity status: UNPINNED (mmdet/torchvision sources are not in the reference tree); checked against the
torch-CPU restatement in oracle/imvoxel_oracle.py.
"""
import os

import torch
from torch import nn

from . import ops
from .conv import FusedConv
from .params import ConvParams, BNParams, invalidate_packed_on_load
from .registry import BACKBONES, NECKS


class _Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride, downsample, dcn=False):
        super().__init__()
        self.stride = stride
        self.dcn = dcn
        self.conv1 = ConvParams(cin, planes, 1, dims=2)
        self.bn1 = BNParams(planes)
        self.conv2 = ConvParams(planes, planes, 3, dims=2)
        if dcn:   # ModulatedDeformConv2dPack: offsets (2*9) + masks (9) from a companion 3x3 conv, zero-initialised
            self.conv2.conv_offset = ConvParams(planes, 27, 3, bias=True, dims=2)
            nn.init.zeros_(self.conv2.conv_offset.weight)
        self.bn2 = BNParams(planes)
        self.conv3 = ConvParams(planes, planes * 4, 1, dims=2)
        self.bn3 = BNParams(planes * 4)
        if downsample:
            self.downsample = nn.Sequential(ConvParams(cin, planes * 4, 1, dims=2), BNParams(planes * 4))
        else:
            self.downsample = None

    def prepare(self, device, in_dtype=None, stream_dtype=None, chain=False, conv2_bf16=False):
        """in_dtype: storage type of the block's INPUT when it differs from the block's own (the first bf16 block after the e4m3
        stages of the fp8 trunk): conv1 and the shortcut conv read it.
        stream_dtype (fp8 trunk, residual='bf16'): storage type of the RESIDUAL STREAM -- the block's input, its shortcut and its
        output -- while the branch (conv1 -> conv2 -> conv3 inputs) uses the current storage type (e4m3): the stream is never
        re-quantised, so the e4m3 rounding noise of one block does not ride on into the next ones."""
        from .conv import current_storage_dtype
        sd = current_storage_dtype()
        xd = stream_dtype or in_dtype or sd            # what conv1 / the shortcut read
        od = stream_dtype or sd                        # what conv3 / the shortcut write
        # style='pytorch': the stride sits on the 3x3 conv
        # chain (fp32 storage with FusedConv.trunk_operands == 4): the block's activations as fp16-pair tensors (ops.PairTensor)
        chain = bool(chain) and sd == torch.float32 and xd == torch.float32 and od == torch.float32
        # conv2_bf16 (fp8 trunk with a bf16 residual stream, detector.calibrate_fp8(variant='conv3')): conv1 and conv2 stay bf16 convolutions, conv2
        # WRITES e4m3 and only conv3 reads e4m3 activations and filters (csrc/model.cpp ivx_model_calibrate_fp8_ex, conv2_bf16 = 1)
        c3_only = bool(conv2_bf16) and stream_dtype is not None and sd != stream_dtype and not self.dcn
        mid = stream_dtype if c3_only else sd          # storage type of conv1's output and of conv2's input / filters
        self.f1 = FusedConv(self.conv1.weight, bn=self.bn1.tensors(), relu=True, dims=2, dtype=xd, out_dtype=mid, chain=chain).to(device)
        if self.dcn:
            if sd != torch.float32:
                raise NotImplementedError('the DCNv2 stages are built for float32 storage only')
            co = self.conv2.conv_offset
            # 27 raw channels + one zero channel (csrc/model.cpp cout_zero): Cout % 4 == 0, so the layer can read pair tensors; the column
            # kernel takes the map's channel count as its row stride
            w_off = torch.cat([co.weight.detach().float(), torch.zeros_like(co.weight.detach()[:1]).float()], 0)
            b_off = torch.cat([co.bias.detach().float(), torch.zeros(1, dtype=torch.float32, device=co.bias.device)], 0)
            self.f_off = FusedConv(w_off, b_off, stride=self.stride, padding=1, dims=2, chain=chain).to(device)
            w = self.conv2.weight.detach()                               # [Cout, C, 3, 3] -> 1x1 over K = (tap, c)
            w_col = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1, 1, 1)
            self.f2 = FusedConv(w_col, bn=self.bn2.tensors(), relu=True, dims=2, chain=chain).to(device)
        else:
            self.f2 = FusedConv(self.conv2.weight, bn=self.bn2.tensors(), stride=self.stride, padding=1, relu=True, dims=2, chain=chain,
                                dtype=mid if c3_only else None, out_dtype=sd if c3_only else None).to(device)
        self.f3 = FusedConv(self.conv3.weight, bn=self.bn3.tensors(), relu=True, dims=2, dtype=sd, out_dtype=od, chain=chain).to(device)  # relu after the add
        self.fd = None
        if self.downsample is not None:
            self.fd = FusedConv(self.downsample[0].weight, bn=self.downsample[1].tensors(), stride=self.stride, dims=2,
                                dtype=xd, out_dtype=od, chain=chain).to(device)
        elif in_dtype is not None and in_dtype != sd and stream_dtype is None:
            raise ValueError('a block without a shortcut conv keeps the storage type of its input')
        # the first block of stage 1 in one launch (ops.bottleneck_proj_fwd_pio): conv3 + shortcut conv as one filter bank (csrc/model.cpp
        # ivx_weights_finalize packs the same bank with the same function)
        self.bank = None
        if (chain and self.fd is not None and self.stride == 1 and not self.dcn and self.f3.pair_ok and self.fd.pair_ok and self.f3._w_tap_host is not None
                and self.fd._w_tap_host is not None and self.f3.cin == 64 and self.fd.cin == 64):
            self.bank = ops.ProjBank(self.f3, self.fd).to(device)

    def takes_pairs(self):
        """every layer that reads the block's INPUT has pair filters (csrc/model.cpp make_plan: wants_pair)"""
        return self.f1.pair_ok and (self.fd is None or self.fd.pair_ok)

    def forward_cl(self, x, out_pair=False):
        """x: channels-last fp32 tensor or, in the pair chain, an ops.PairTensor; out_pair: the block's output as a PairTensor (the caller
        knows its consumers).  Inside the block a tensor is a pair tensor when its producer read pairs and its only consumer has pair
        filters -- the rule of the native handle, so both hosts run the same kernels on the same bits."""
        if not isinstance(x, ops.PairTensor):
            idt = x if self.fd is None else self.fd(x)
            y = self.f1(x)
            if self.dcn:
                y = self.f2(ops.dcn_im2col(y, self.f_off(y), 3, self.stride, 1, 1))
            else:
                y = self.f2(y)
            return self.f3(y, res=idt)
        if self.fuses_proj(x, out_pair):
            return self._fused_proj(x)
        idt = x if self.fd is None else self.fd(x)                 # shortcut conv: fp32 out (only read as a residual), max |out| recorded
        if self.dcn:
            # conv1 feeds conv_offset and the column kernel: pairs when both can read them and the columns can be pairs too (wants_pair of
            # csrc/model.cpp: the contraction conv has pair filters, 9x the map stays below 2 GiB); else fp32 as before round 4
            n1 = x.shape[0] * x.shape[2] * x.shape[3] * self.f1.cout
            if self.f_off.pair_ok and self.f2.pair_ok and self.f1.cout % 16 == 0 and n1 * 9 * 4 < 2 ** 31:
                y = self.f1(x, out_pair=True)
                col = ops.dcn_im2col_pair(y, self.f_off(y), 3, self.stride, 1, 1)
                return self.f3(self.f2(col, out_pair=self.f3.pair_ok), res=idt, out_pair=out_pair)
            y = self.f1(x)
            y = self.f2(ops.dcn_im2col(y, self.f_off(y), 3, self.stride, 1, 1))
            return self.f3(y, res=idt)
        if self.fuses(x, out_pair):
            return self._fused(x)
        y = self.f1(x, out_pair=self.f2.pair_ok)
        y = self.f2(y, out_pair=self.f3.pair_ok)
        return self.f3(y, res=idt, out_pair=out_pair)

    # IVX_FUSE_BOTTLENECK=0: the three-launch form everywhere (A/B; csrc/model.cpp make_plan reads the same variable)
    fuse = os.environ.get('IVX_FUSE_BOTTLENECK', '1') != '0'

    def fuses(self, x, out_pair):
        """the block runs as ONE launch (ops.bottleneck_fwd_pio, csrc/bottleneck.hip): an identity block of the pair chain with 64 or 128
        planes whose output is a pair tensor -- the rule of csrc/model.cpp make_plan, so both hosts launch the same kernels"""
        return (_Bottleneck.fuse and self.fd is None and not self.dcn and self.stride == 1 and bool(out_pair) and isinstance(x, ops.PairTensor)
                and self.f1.pair_ok and self.f2.pair_ok and self.f3.pair_ok and self.f1.cin == 4 * self.f1.cout and self.f3.cout == 4 * self.f1.cout
                and x.shape[1] == 1 and ops.bottleneck_supported(x.shape[0], x.shape[2], x.shape[3], self.f1.cout))

    def fuses_proj(self, x, out_pair):
        """the first block of stage 1 (shortcut conv, stride 1, 64 -> 64 -> 64 -> 256) runs as ONE launch (ops.bottleneck_proj_fwd_pio) -- the rule of
        csrc/model.cpp make_plan (fuse 5)"""
        return (_Bottleneck.fuse and getattr(self, 'bank', None) is not None and bool(out_pair) and isinstance(x, ops.PairTensor)
                and self.f1.pair_ok and self.f2.pair_ok and x.shape[1] == 1 and x.shape[4] == self.bank.cin
                and ops.bottleneck_proj_supported(x.shape[0], x.shape[2], x.shape[3], self.f1.cout, x.shape[4]))

    def _fused_proj(self, x):
        tr = FusedConv.trace is not None
        if tr:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        y = ops.bottleneck_proj_fwd_pio(x, self.f1, self.f2, self.bank)
        P, ci = self.f1.cout, self.bank.cin
        fl = 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * (ci * P + 9.0 * P * P + 4.0 * P * P + 4.0 * P * ci)
        if FusedConv.count_flops:
            FusedConv.flops += fl
            FusedConv.exec_flops += 3.0 * fl
        if tr:
            e1.record()
            FusedConv.trace.append(('direct', e0, e1, 3.0 * fl, float(4 * x.numel() + 4 * y.numel()), False,
                                    f'bottleneck {ci}->{P}->{P}->{4 * P} + shortcut conv in {tuple(x.shape[:4])} pair, one launch'))
        return y

    def _fused(self, x):
        tr = FusedConv.trace is not None
        if tr:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        y = ops.bottleneck_fwd_pio(x, self.f1, self.f2, self.f3)
        P = self.f1.cout
        fl = 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * 17.0 * P * P
        if FusedConv.count_flops:
            FusedConv.flops += fl
            FusedConv.exec_flops += 3.0 * fl
        if tr:
            e1.record()
            FusedConv.trace.append(('direct', e0, e1, 3.0 * fl, float(4 * x.numel() + 4 * y.numel()), False,
                                    f'bottleneck {4 * P}->{P}->{P}->{4 * P} in {tuple(x.shape[:4])} pair, one launch'))
        return y


def stem_s2d_weights(w):
    """[Cout, 3, 7, 7] stem filters -> [Cout, 16, 4, 4]: the same convolution as a 4x4 stride-1 pad-1 one over the 2x2
    space-to-depth blocks of ops.image_s2d_bf16 (block p holds the pixels (2p-1, 2p); input channel (a*2+e)*3 + c is pixel
    (2*ph-1+a, 2*pw-1+e), colour c; tap kh = 2*th + a, kw = 2*tw + e, the eighth row / column and channels 12..15 are zero)."""
    w8 = torch.zeros(w.shape[0], 3, 8, 8, dtype=torch.float32)
    w8[:, :, :7, :7] = w.detach().float().cpu()
    # [co, c, th, a, tw, e] -> [co, (a, e, c), th, tw]
    w2 = w8.reshape(-1, 3, 4, 2, 4, 2).permute(0, 3, 5, 1, 2, 4).reshape(-1, 12, 4, 4)
    return torch.cat([w2, torch.zeros(w2.shape[0], 4, 4, 4)], 1)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1, norm_cfg=None, norm_eval=True,
                 style='pytorch', in_channels=3, dcn=None, stage_with_dcn=(False, False, False, False), **kwargs):
        super().__init__()
        if depth not in self.arch:
            raise KeyError(f'ResNet depth {depth} is not built (bottleneck depths: {sorted(self.arch)})')
        if style != 'pytorch':
            raise NotImplementedError("only style='pytorch' (stride on the 3x3 conv) is built")
        if dcn is not None:
            if dcn.get('type') != 'DCNv2' or dcn.get('deform_groups', 1) != 1:
                raise NotImplementedError('only dcn=dict(type="DCNv2", deform_groups=1) (the nuScenes ImVoxelNet config) is built')
            if dcn.get('fallback_on_stride', False):
                raise NotImplementedError('fallback_on_stride=True is not built')
        self.out_indices = tuple(out_indices)
        self.conv1 = ConvParams(in_channels, 64, 7, dims=2)
        self.bn1 = BNParams(64)
        cin = 64
        blocks = self.arch[depth][:num_stages]
        for i, nb in enumerate(blocks):
            planes = 64 * 2 ** i
            layer = []
            for j in range(nb):
                layer.append(_Bottleneck(cin, planes, (2 if i > 0 else 1) if j == 0 else 1, downsample=(j == 0),
                                         dcn=dcn is not None and bool(stage_with_dcn[i])))
                cin = planes * 4
            setattr(self, f'layer{i + 1}', nn.Sequential(*layer))
        self.num_stages = len(blocks)
        self._device = None
        invalidate_packed_on_load(self)

    def init_weights(self, pretrained=None):
        """mmdet ResNet.init_weights(pretrained): a checkpoint path is loaded; the model-zoo scheme of the reference configs
        ('torchvision://resnet50', imvoxelnet_kitti.py:3) needs torchvision's download cache, which this package does
        not have -- it is NOT silently ignored: a warning says the backbone stays randomly initialised until a
        checkpoint is loaded (released ImVoxelNet checkpoints contain the backbone: data.load_checkpoint)."""
        if pretrained is None:
            return
        if str(pretrained).startswith(('torchvision://', 'open-mmlab://', 'http://', 'https://')):
            import warnings
            warnings.warn(f"pretrained='{pretrained}' is a model-zoo URL and cannot be fetched here: the backbone keeps its random "
                          'initialisation until a checkpoint is loaded (imvoxelnet_amd.load_checkpoint)', RuntimeWarning, stacklevel=2)
            return
        from .data import torch_load_trusted
        sd = torch_load_trusted(pretrained, map_location='cpu')
        self.load_state_dict(sd.get('state_dict', sd) if isinstance(sd, dict) else sd, strict=False)

    def prepare(self, device):
        from .conv import current_storage_dtype, FP8
        fp8 = current_storage_dtype() == FP8       # optional: e4m3 storage of the trunk's activations (detector.calibrate_fp8)
        self.stem = None
        # fp32 storage: the trunk's activations chained as fp16-pair tensors (FusedConv.trunk_operands, conv.py)
        self.chain = current_storage_dtype() == torch.float32 and FusedConv.trunk_operands == ops.IVX_F16_PAIR
        self.stem_frag = None
        if not fp8:
            self.stem = FusedConv(self.conv1.weight, bn=self.bn1.tensors(), stride=2, padding=3, relu=True, dims=2, chain=self.chain).to(device)
            # the head of the pair chain in one launch (ops.stem_pool_pair; csrc/model.cpp pack_layer builds the same filters)
            if self.chain and tuple(self.conv1.weight.shape) == (64, 3, 7, 7):
                fr, sp = ops.stem_pool_pack_filters(self.conv1.weight, self.stem._scale_host)
                self.stem_frag = (fr.to(device), sp.to(device))
        # bf16 mode: the 7x7 stride-2 stem as a 4x4 stride-1 convolution over 2x2 space-to-depth blocks of the image
        # (ops.image_s2d_bf16): bf16 MFMA with K = 256 instead of the fp32 kernel on 3 (padded to 4) channels
        self.stem_s2d = None
        w = self.conv1.weight.detach()
        if fp8 and tuple(w.shape[1:]) != (3, 7, 7):
            raise NotImplementedError('fp8 trunk storage is built for the 3-channel 7x7 stem')
        # fp8 trunk, residual='bf16' (detector.calibrate_fp8): only the inside of every bottleneck is e4m3 -- the stem, the
        # max-pool and every block output (the residual stream) stay bf16
        res_bf16 = fp8 and getattr(self, 'fp8_residual', 'fp8') == 'bf16'
        if current_storage_dtype() in (torch.bfloat16, FP8) and tuple(w.shape[1:]) == (3, 7, 7):
            w2 = stem_s2d_weights(w)
            self.stem_s2d = FusedConv(w2, bn=self.bn1.tensors(), stride=1, padding=1, relu=True, dims=2, dtype=torch.bfloat16,
                                      out_dtype=FP8 if (fp8 and not res_bf16) else torch.bfloat16, key=id(self.conv1.weight)).to(device)
        # fp8 trunk: the first `fp8_stages` stages store e4m3, the rest bf16 (their residual streams carry 8 mantissa bits
        # again; the first block of the first bf16 stage reads e4m3).  stage_dtypes: storage type of every stage's OUTPUT.
        from .conv import storage_dtype
        n_req = getattr(self, 'fp8_stages', None)      # set by detector.calibrate_fp8(stages=...); None: all stages
        n8 = 0 if not fp8 else (self.num_stages if n_req is None else max(0, min(int(n_req), self.num_stages)))
        self.stage_dtypes = []
        prev = torch.bfloat16 if res_bf16 else current_storage_dtype()
        first8 = int(getattr(self, 'fp8_first_stage', 0) or 0) if res_bf16 else 0      # stages below it keep plain bf16 bottlenecks
        c3_only = res_bf16 and getattr(self, 'fp8_variant', 'full') == 'conv3'
        for i in range(self.num_stages):
            sd = current_storage_dtype() if not fp8 else (FP8 if (i < n8 and i >= first8) else torch.bfloat16)
            with storage_dtype(sd):
                for j, blk in enumerate(getattr(self, f'layer{i + 1}')):
                    if res_bf16:
                        blk.prepare(device, stream_dtype=torch.bfloat16 if sd == FP8 else None, conv2_bf16=c3_only)
                    else:
                        blk.prepare(device, in_dtype=prev if (j == 0 and prev != sd) else None, chain=self.chain)
            self.stage_dtypes.append(torch.bfloat16 if res_bf16 else sd)
            prev = self.stage_dtypes[-1]
        self._device = device
        return self

    def forward_image(self, img):
        """img [N,3,H,W] (reference layout, fp32) -> tuple of channels-last stage outputs."""
        if self._device is None:
            self.prepare(img.device)
        if self.stem_s2d is not None and img.shape[-1] % 2 == 0 and img.shape[-2] % 2 == 0 and img.dtype == torch.float32:
            return self._stages(self.stem_s2d(ops.image_s2d_bf16(img.contiguous())))
        if self.stem is None:
            raise ValueError('the fp8 trunk takes float32 images with even height and width')
        if getattr(self, 'chain', False):
            if self.fuses_stem(img):       # layout change + stem + max-pool in one launch (csrc/model.cpp make_plan: the same rule)
                tr = FusedConv.trace is not None
                if tr:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                x = ops.stem_pool_pair(img.contiguous(), self.stem_frag[0], self.stem_frag[1], self.stem.shift, self.stem.wbound, self.stem.sbound)
                hc, wc = (img.shape[2] - 1) // 2 + 1, (img.shape[3] - 1) // 2 + 1
                fl = 2.0 * img.shape[0] * hc * wc * 64 * 147
                if FusedConv.count_flops:
                    FusedConv.flops += fl
                    FusedConv.exec_flops += 3.0 * fl
                if tr:
                    e1.record()
                    FusedConv.trace.append(('direct', e0, e1, 3.0 * fl, float(4 * img.numel() + 4 * x.numel()), False,
                                            f'stem 3->64 k7 s2 + max-pool in {tuple(img.shape)} pair, one launch'))
                return self._stages(x, None, pooled=True)
            return self.forward_cl(ops.to_channels_last_amax(img.contiguous(), pad_to=4))
        return self.forward_cl(ops.to_channels_last(img.contiguous(), pad_to=4))

    def forward_cl(self, x):
        """x [N,1,H,W,4] channels-last image (3 channels zero-padded to 4) -> tuple of stage outputs."""
        if self._device is None:
            self.prepare(x.device)
        return self._stages(self.stem(x), ops.slots_of(x))

    # pair chain: which stage outputs may be PairTensors -- their consumers outside this module must be convolutions with pair filters
    # (the FPN laterals); the detector clears the last entry when a LayoutHead pools C5 (features_2d_cl)
    stage_out_pair = (True, True, True, True)

    # IVX_FUSE_STEM=0: the three-launch head (layout change, fp32-MFMA stem, max-pool) -- A/B; csrc/model.cpp make_plan reads the same variable
    fuse_stem = os.environ.get('IVX_FUSE_STEM', '1') != '0'

    def fuses_stem(self, img):
        """the chain's head runs as one launch: the conditions under which _stages would write the pooled map as a pair tensor"""
        if not (ResNet.fuse_stem and getattr(self, 'stem_frag', None) is not None and img.dim() == 4 and img.shape[1] == 3 and img.dtype == torch.float32
                and img.shape[2] >= 7 and img.shape[3] >= 7):
            return False
        hc, wc = (img.shape[2] - 1) // 2 + 1, (img.shape[3] - 1) // 2 + 1
        return (self.layer1[0].takes_pairs() and img.shape[0] * ((hc - 1) // 2 + 1) * ((wc - 1) // 2 + 1) * 64 * 4 < 2 ** 31)

    def _stages(self, x, img_slots=None, pooled=False):
        from .conv import QTensor
        blocks = [list(getattr(self, f'layer{i + 1}')) for i in range(self.num_stages)]
        chain = (getattr(self, 'chain', False) and img_slots is not None and self.stem is not None and isinstance(x, torch.Tensor)
                 and x.shape[-1] % 16 == 0 and blocks[0][0].takes_pairs()
                 and x.shape[0] * ((x.shape[2] - 1) // 2 + 1) * ((x.shape[3] - 1) // 2 + 1) * x.shape[4] * 4 < 2 ** 31)
        if pooled:     # the one-launch head already wrote the pooled pair map
            pass
        elif chain:    # fp32 stem output -> pair tensor, scaled by the bound of the stem's output from max |image|
            x = ops.maxpool2d_pair(x, img_slots, self.stem.wbound, self.stem.sbound, 3, 2, 1)
        else:
            x = QTensor(ops.maxpool2d(x.data, 3, 2, 1), x.scale) if isinstance(x, QTensor) else ops.maxpool2d(x, 3, 2, 1)
        outs = []
        for i in range(self.num_stages):
            for j, blk in enumerate(blocks[i]):
                out_pair = False
                if isinstance(x, ops.PairTensor):      # consumers of the block's output: the next block, the FPN lateral of a stage output
                    if j + 1 < len(blocks[i]):
                        out_pair = blocks[i][j + 1].takes_pairs()
                    else:
                        nxt = blocks[i + 1][0].takes_pairs() if i + 1 < self.num_stages else True
                        lat = self.stage_out_pair[i] if i in self.out_indices else True
                        out_pair = nxt and lat and (i + 1 < self.num_stages or i in self.out_indices)
                    x = blk.forward_cl(x, out_pair=out_pair)
                else:
                    x = blk.forward_cl(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def forward(self, img):
        """img [N,3,H,W] -> tuple of [N,C,h,w] (reference layout)."""
        from .conv import QTensor
        return tuple(ops.from_channels_last(o.float() if isinstance(o, (QTensor, ops.PairTensor)) else o, 2) for o in self.forward_image(img))


@NECKS.register_module()
class FPN(nn.Module):
    """mmdet FPN restated: lateral 1x1 (bias) -> top-down nearest x2 add -> 3x3 (bias); no norm, no
    activation.  Only level 0 is consumed by ImVoxelNet (detectors/imvoxelnet.py:50), so levels 1..3 are
    computed only on request; their parameters are still held so checkpoints load strictly."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False, **kwargs):
        super().__init__()
        if start_level != 0 or end_level != -1 or add_extra_convs:
            raise NotImplementedError('only the plain FPN used by the ImVoxelNet configs is built')
        self.in_channels = list(in_channels)
        self.out_channels = out_channels
        self.num_outs = num_outs
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for c in self.in_channels:
            lat, out = nn.Module(), nn.Module()
            lat.conv = ConvParams(c, out_channels, 1, bias=True, dims=2)
            out.conv = ConvParams(out_channels, out_channels, 3, bias=True, dims=2)
            self.lateral_convs.append(lat)
            self.fpn_convs.append(out)
        self._device = None
        invalidate_packed_on_load(self)

    def init_weights(self):
        pass

    def prepare(self, device, in_dtype=None):
        """in_dtype: storage type(s) of the backbone's stage outputs when they differ from the FPN's own (the fp8 trunk: e4m3 in,
        the FPN's storage type -- bf16 -- out); one type or a list per level."""
        from .conv import current_storage_dtype
        od = current_storage_dtype()
        ind = list(in_dtype) if isinstance(in_dtype, (list, tuple)) else [in_dtype] * len(self.lateral_convs)
        chain = od == torch.float32 and all(t in (None, torch.float32) for t in ind) and FusedConv.trunk_operands == ops.IVX_F16_PAIR
        self.flat = [FusedConv(m.conv.weight, m.conv.bias, dims=2, dtype=ind[i] or od, out_dtype=od, chain=chain).to(device)
                     for i, m in enumerate(self.lateral_convs)]
        self.fout = [FusedConv(m.conv.weight, m.conv.bias, padding=1, dims=2, chain=chain).to(device) for m in self.fpn_convs]
        self._device = device
        return self

    def forward_cl(self, feats, all_levels=False):
        if self._device is None:
            self.prepare(feats[0].device)
        n = len(feats)
        lat = [None] * n
        # pair chain: a lateral map is written as a PairTensor when its lateral conv read one and an output conv with pair filters reads
        # it (level 0 always; the others only with all_levels) -- maps that are only added top-down stay fp32
        # ... unless that output conv is one of the wide layers on a large map whose Winograd form wins (FusedConv.prefers_winograd)
        npos = lambda i: feats[i].shape[0] * feats[i].shape[2] * feats[i].shape[3]
        wp = lambda i: (isinstance(feats[i], ops.PairTensor) and self.fout[i].pair_ok and (i == 0 or all_levels)
                        and not self.fout[i].prefers_winograd(npos(i)))
        lat[n - 1] = self.flat[n - 1](feats[n - 1], out_pair=wp(n - 1))
        for i in range(n - 2, -1, -1):   # lateral conv + nearest-upsampled coarser level, fused in the epilogue
            res = lat[i + 1]
            if isinstance(res, ops.PairTensor) and not isinstance(feats[i], ops.PairTensor):
                res = res.float()
            lat[i] = self.flat[i](feats[i], res=res, res_mode=2, out_pair=wp(i) and ops.slots_of(res) is not None)
        outs = [self.fout[0](lat[0])]
        if all_levels:
            outs += [self.fout[i](lat[i]) for i in range(1, n)]
        return [o.float() if isinstance(o, ops.PairTensor) else o for o in outs]

    def forward(self, inputs, all_levels=True):
        outs = self.forward_cl([ops.to_channels_last(t.contiguous()) for t in inputs], all_levels)
        return tuple(ops.from_channels_last(o, 2) for o in outs)
