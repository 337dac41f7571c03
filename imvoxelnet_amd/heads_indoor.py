"""Anchor-free indoor heads (inference half) under the reference's registry names and state-dict keys:
ScanNetImVoxelHeadV2 / SunRgbdImVoxelHeadV2 (mmdet3d/models/dense_heads/imvoxel_head_v2.py) and the V1
ScanNetImVoxelHead / SunRgbdImVoxelHead (imvoxel_head.py; the reference configs use n_convs=0).

The three 3x3x3 output convs run as ONE fused conv per level (channels [centerness | reg | cls]); per level the
device tail (ivx_fcos_head_level_candidates) does valid-mask resize, sigmoid scoring, top-k and decoding; the
cross-level NMS runs on the device through nms.aligned_3d_nms / nms.box3d_multiclass_nms.
"""
import ctypes as C

import torch
from torch import nn

from . import ops, _lib
from .boxes import DepthInstance3DBoxes
from .conv import FusedConv
from .heads import bias_init_with_prob
from .params import ConvParams, BNParams, ScaleParams, invalidate_packed_on_load
from .registry import HEADS, ConfigDict


class _ImVoxelHeadBase(nn.Module):
    n_reg_default = 6

    def __init__(self, n_classes, n_channels, n_reg_outs, n_scales=3, n_convs=0, limit=None, centerness_topk=-1,
                 regress_ranges=None, loss_centerness=None, loss_bbox=None, loss_cls=None, train_cfg=None, test_cfg=None):
        super().__init__()
        self.n_classes, self.n_channels, self.n_reg_outs, self.n_convs = n_classes, n_channels, n_reg_outs, n_convs
        self.n_scales = len(regress_ranges) if regress_ranges is not None else n_scales
        self.limit, self.centerness_topk = limit, centerness_topk
        self.train_cfg = train_cfg
        self.test_cfg = ConfigDict(test_cfg) if test_cfg is not None else None
        if n_convs > 0 or self._v1:   # towers exist (possibly empty) only in the V1 heads
            mk = lambda: nn.Sequential(*[nn.Sequential(ConvParams(n_channels, n_channels, 3), BNParams(n_channels), nn.Identity())
                                         for _ in range(n_convs)])
            self.reg_convs, self.cls_convs = mk(), mk()
        self.centerness_conv = ConvParams(n_channels, 1, 3)
        self.reg_conv = ConvParams(n_channels, n_reg_outs, 3)
        self.cls_conv = ConvParams(n_channels, n_classes, 3, bias=True)
        self.scales = nn.ModuleList([ScaleParams(1.) for _ in range(self.n_scales)])
        self.voxel_size = None
        self._device = None
        invalidate_packed_on_load(self)
        self.init_weights()

    _v1 = False

    def init_weights(self):
        for m in (self.centerness_conv, self.reg_conv, self.cls_conv):
            nn.init.normal_(m.weight, 0, 0.01)
        nn.init.constant_(self.cls_conv.bias, bias_init_with_prob(0.01))

    def prepare(self, device):
        zeros = torch.zeros(1 + self.n_reg_outs)
        if self.n_convs == 0:
            w = torch.cat([self.centerness_conv.weight, self.reg_conv.weight, self.cls_conv.weight], 0)
            self.fhead = FusedConv(w, torch.cat([zeros, self.cls_conv.bias.detach().cpu()]), padding=1, out_dtype=torch.float32).to(device)
        else:
            self.ftow_reg = [FusedConv(t[0].weight, bn=t[1].tensors(), padding=1, relu=True).to(device) for t in self.reg_convs]
            self.ftow_cls = [FusedConv(t[0].weight, bn=t[1].tensors(), padding=1, relu=True).to(device) for t in self.cls_convs]
            self.freg = FusedConv(torch.cat([self.centerness_conv.weight, self.reg_conv.weight], 0), padding=1, out_dtype=torch.float32).to(device)
            self.fcls = FusedConv(self.cls_conv.weight, self.cls_conv.bias, padding=1, out_dtype=torch.float32).to(device)   # the tail reads fp32
        self._scale_vals = [float(s.scale) for s in self.scales]
        self._device = device
        return self

    def forward_cl(self, xs):
        """list of channels-last level maps -> list of fused head outputs [B,nx,ny,nz,1+R+ncls] (raw regression)."""
        if self._device is None:
            self.prepare(xs[0].device)
        if self.n_convs == 0:
            return [self.fhead(x) for x in xs]
        outs = []
        for x in xs:
            r, c = x, x
            for f in self.ftow_reg:
                r = f(r)
            for f in self.ftow_cls:
                c = f(c)
            outs.append(torch.cat([self.freg(r), self.fcls(c)], dim=-1).contiguous())
        return outs

    def forward(self, x):
        """Reference surface: list of [B,C,nx,ny,nz] -> (centernesses, bbox_preds, cls_scores), bbox_preds already
        exp(scale * .) (angle channel raw), as forward_single returns them."""
        fused = self.forward_cl([ops.to_channels_last(t.contiguous()) for t in x])
        R = self.n_reg_outs
        cs, bs, ss = [], [], []
        for lvl, f in enumerate(fused):
            y = ops.from_channels_last(f, 3)
            d = torch.exp(y[:, 1:7] * self.scales[lvl].scale.to(y.device))
            cs.append(y[:, :1])
            bs.append(d if R == 6 else torch.cat([d, y[:, 7:8]], dim=1))
            ss.append(y[:, 1 + R:])
        return cs, bs, ss

    # ------------------------------------------------------------------ tail
    def _level_geometry(self, shape, lvl, img_metas, device):
        """voxel_size * 2^lvl and origin - n/2 * that, per sample, with the reference's torch CPU ops
        (imvoxel_head_v2.py:206-214 -> detectors/imvoxelnet.py:132-141)."""
        vs = torch.tensor(self.voxel_size) * (2 ** lvl)
        nv = torch.tensor(list(shape))
        no = torch.stack([torch.tensor(m['lidar2img']['origin']).float() - nv / 2. * vs for m in img_metas])
        return vs.unsqueeze(0).expand(len(img_metas), 3).contiguous().to(device), no.contiguous().to(device)

    def get_candidates_cl(self, fused, valid, img_metas, scales=None, want_index=False):
        """-> per-sample (boxes [n,R], scores [n,ncls]) concatenated over levels (device tensors).
        want_index (parity tests): also the source location of every candidate, int64 [n] = level * 2^32 + flat voxel
        index inside the level grid (the top-k indices the tail leaves in its workspace)."""
        cfg = self.test_cfg
        B = fused[0].shape[0]
        v0 = valid.reshape(B, *valid.shape[-3:]).to(torch.uint8).contiguous()
        X, Y, Z = v0.shape[1:]
        L = _lib.lib()
        boxes_l, scores_l, index_l = [], [], []
        for lvl, f in enumerate(fused):
            nx, ny, nz, CH = f.shape[1:]
            n = nx * ny * nz
            k = cfg.nms_pre if 0 < cfg.nms_pre < n else n
            vs, no = self._level_geometry((nx, ny, nz), lvl, img_metas, f.device)
            wsb = L.ivx_fcos_head_workspace_bytes(B, n, int(cfg.nms_pre))
            if wsb < 0:
                raise ValueError('ivx_fcos_head_workspace_bytes: unsupported size (more than 65536 candidates per level?)')
            ws = torch.empty((wsb,), device=f.device, dtype=torch.uint8)
            cb = torch.empty((B, k, self.n_reg_outs), device=f.device, dtype=torch.float32)
            cs = torch.empty((B, k, self.n_classes), device=f.device, dtype=torch.float32)
            cnt = torch.empty((B,), device=f.device, dtype=torch.int32)
            sc = self._scale_vals[lvl] if scales is None else scales[lvl]
            _lib.check(L.ivx_fcos_head_level_candidates(
                ops._ptr(f), ops._ptr(v0), ops._ptr(vs), ops._ptr(no), C.c_float(sc), B, nx, ny, nz, CH, self.n_classes,
                self.n_reg_outs, lvl, X, Y, Z, int(cfg.nms_pre), ops._ptr(ws), wsb, ops._ptr(cb), ops._ptr(cs), ops._ptr(cnt),
                ops._stream()), 'ivx_fcos_head_level_candidates')
            boxes_l.append(cb)
            scores_l.append(cs)
            if want_index:   # workspace layout of ivx_fcos_head_level_candidates: keys [B,n] f32 | top-k indices [B,kpad] i32 | ...
                kpad = 1 << (max(k, 64) - 1).bit_length()
                off = (B * n * 4 + 255) // 256 * 256
                topk = ws[off:off + B * kpad * 4].view(torch.int32).view(B, kpad)[:, :k]
                index_l.append(topk.to(torch.int64) + (lvl << 32))
        boxes, scores = torch.cat(boxes_l, 1), torch.cat(scores_l, 1)
        if want_index:
            index = torch.cat(index_l, 1)
            return [(boxes[b], scores[b], index[b]) for b in range(B)]
        return [(boxes[b], scores[b]) for b in range(B)]

    def get_bboxes_cl(self, fused, valid, img_metas, scales=None):
        return [self._nms(b, s, m) for (b, s), m in zip(self.get_candidates_cl(fused, valid, img_metas, scales), img_metas)]

    def _nms(self, bboxes, scores, img_meta):
        """The cross-level tail on the device (ivx_indoor_tail_get_bboxes) for ONE sample's concatenated candidates [K,R] / [K,ncls]:
        ScanNet (imvoxel_head_v2.py:528-545): class maximum, score threshold, class-aware aligned 3-D NMS, corner -> centre / size;
        SUN RGB-D (:397-417): BEV boxes, per-class rotated NMS, max_num = nms_pre.  One host round trip: the count."""
        tc = self.test_cfg
        sun = self.n_reg_outs == 7
        b7, s, l, cnt = ops.indoor_tail([bboxes.unsqueeze(0).contiguous()], [scores.unsqueeze(0).contiguous()], self.n_classes, tc.score_thr,
                                        tc.nms_thr if sun else tc.iou_thr, tc.use_rotate_nms if sun else False, tc.nms_pre if sun else None)
        n = int(cnt[0])
        box_type = img_meta.get('box_type_3d', DepthInstance3DBoxes)
        return box_type(b7[0, :n], box_dim=7, with_yaw=sun), s[0, :n], l[0, :n]

    def get_bboxes(self, centernesses, bbox_preds, cls_scores, valid, img_metas):
        """Reference signature (inputs as returned by forward).  The decoded distances are fed back through the
        fused-layout tail with scale 1 (exp(log d) == d to 1 ulp)."""
        fused = []
        for c, d, s in zip(centernesses, bbox_preds, cls_scores):
            raw = torch.log(d[:, :6]) if d.shape[1] == 6 else torch.cat([torch.log(d[:, :6]), d[:, 6:]], 1)
            fused.append(ops.to_channels_last(torch.cat([c, raw, s], 1).contiguous()))
        return self.get_bboxes_cl(fused, valid, img_metas, scales=[1.0] * len(fused))


class _ScanNetMixin:
    """n_reg_outs 6: axis-aligned corner boxes, aligned 3-D NMS (the tail lives in _ImVoxelHeadBase._nms)."""


class _SunRgbdMixin:
    """n_reg_outs 7: rotated boxes, multi-class BEV NMS (the tail lives in _ImVoxelHeadBase._nms)."""


@HEADS.register_module()
class ScanNetImVoxelHeadV2(_ScanNetMixin, _ImVoxelHeadBase):
    def __init__(self, n_classes, n_channels, n_reg_outs, n_scales, limit, centerness_topk=-1, **kw):
        super().__init__(n_classes, n_channels, n_reg_outs, n_scales=n_scales, limit=limit, centerness_topk=centerness_topk, **kw)


@HEADS.register_module()
class SunRgbdImVoxelHeadV2(_SunRgbdMixin, _ImVoxelHeadBase):
    def __init__(self, n_classes, n_channels, n_reg_outs, n_scales, limit, centerness_topk=-1, **kw):
        super().__init__(n_classes, n_channels, n_reg_outs, n_scales=n_scales, limit=limit, centerness_topk=centerness_topk, **kw)


_INF = 1e8
_RANGES = ((-1., .75), (.75, 1.5), (1.5, _INF))


@HEADS.register_module()
class ScanNetImVoxelHead(_ScanNetMixin, _ImVoxelHeadBase):
    _v1 = True

    def __init__(self, n_classes, n_channels, n_convs, n_reg_outs, centerness_topk=-1, regress_ranges=_RANGES, **kw):
        super().__init__(n_classes, n_channels, n_reg_outs, n_convs=n_convs, centerness_topk=centerness_topk,
                         regress_ranges=regress_ranges, **kw)


@HEADS.register_module()
class SunRgbdImVoxelHead(_SunRgbdMixin, _ImVoxelHeadBase):
    _v1 = True

    def __init__(self, n_classes, n_channels, n_convs, n_reg_outs, centerness_topk=-1, regress_ranges=_RANGES, **kw):
        super().__init__(n_classes, n_channels, n_reg_outs, n_convs=n_convs, centerness_topk=centerness_topk,
                         regress_ranges=regress_ranges, **kw)
