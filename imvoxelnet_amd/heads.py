"""Anchor3DHead (inference half) under the reference's registry name and state-dict keys
(mmdet3d/models/dense_heads/anchor3d_head.py:16-153, 375-517).  The three 1x1 convs run as ONE fused
conv (channel blocks [cls | reg | dir]) and get_bboxes is one device-side pipeline
(ops.anchor_head_get_bboxes) instead of ~40 small launches + a blocking D2H per sample.
"""
import numpy as np
import torch
from torch import nn

from . import ops
from .boxes import LiDARInstance3DBoxes
from .conv import FusedConv
from .params import ConvParams, invalidate_packed_on_load
from .registry import HEADS, ConfigDict, build_anchor_generator, build_bbox_coder


def bias_init_with_prob(prior_prob):
    return float(-np.log((1 - prior_prob) / prior_prob))


@HEADS.register_module()
class Anchor3DHead(nn.Module):
    def __init__(self, num_classes, in_channels, train_cfg=None, test_cfg=None, feat_channels=256,
                 use_direction_classifier=True, anchor_generator=None, assigner_per_size=False, assign_per_class=False,
                 diff_rad_by_sin=True, dir_offset=0, dir_limit_offset=1, bbox_coder=None, loss_cls=None, loss_bbox=None,
                 loss_dir=None):
        super().__init__()
        anchor_generator = anchor_generator or dict(type='Anchor3DRangeGenerator', ranges=[[0, -39.68, -1.78, 69.12, 39.68, -1.78]],
                                                    sizes=[[1.6, 3.9, 1.56]], rotations=[0, 1.57], reshape_out=False)
        anchor_generator = {('ranges' if k == 'range' else k): ([v] if k == 'range' else v)
                            for k, v in anchor_generator.items() if k != 'strides'}
        bbox_coder = bbox_coder or dict(type='DeltaXYZWLHRBBoxCoder')
        loss_cls = loss_cls or dict(type='CrossEntropyLoss', use_sigmoid=True)
        self.in_channels, self.num_classes, self.feat_channels = in_channels, num_classes, feat_channels
        self.use_direction_classifier = use_direction_classifier
        self.train_cfg = train_cfg
        self.test_cfg = ConfigDict(test_cfg) if test_cfg is not None else None
        self.dir_offset, self.dir_limit_offset = dir_offset, dir_limit_offset
        self.anchor_generator = build_anchor_generator(anchor_generator)
        self.num_anchors = self.anchor_generator.num_base_anchors
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.box_code_size = self.bbox_coder.code_size
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)
        if not self.use_sigmoid_cls:
            raise NotImplementedError('softmax classification is not used by any ImVoxelNet config and is not built')
        if not use_direction_classifier:
            raise NotImplementedError('the ImVoxelNet anchor heads always use the direction classifier')
        if self.box_code_size != 7:
            raise NotImplementedError('box_code_size must be 7')
        self.cls_out_channels = self.num_anchors * self.num_classes
        self.conv_cls = ConvParams(feat_channels, self.cls_out_channels, 1, bias=True, dims=2)
        self.conv_reg = ConvParams(feat_channels, self.num_anchors * self.box_code_size, 1, bias=True, dims=2)
        self.conv_dir_cls = ConvParams(feat_channels, self.num_anchors * 2, 1, bias=True, dims=2)
        self.voxel_size = None
        self._device = None
        invalidate_packed_on_load(self)
        self.init_weights()

    def init_weights(self):
        """anchor3d_head.py:132-136: normal(std=.01) weights, cls bias = -log((1-p)/p), p = .01."""
        nn.init.normal_(self.conv_cls.weight, 0, 0.01)
        nn.init.constant_(self.conv_cls.bias, bias_init_with_prob(0.01))
        nn.init.normal_(self.conv_reg.weight, 0, 0.01)
        nn.init.constant_(self.conv_reg.bias, 0)

    # channel blocks of the fused conv output
    @property
    def _offs(self):
        a = self.cls_out_channels
        return (0, a, a + self.num_anchors * self.box_code_size)

    def prepare(self, device):
        w = torch.cat([self.conv_cls.weight, self.conv_reg.weight, self.conv_dir_cls.weight], 0)
        b = torch.cat([self.conv_cls.bias, self.conv_reg.bias, self.conv_dir_cls.bias], 0)
        self.fhead = FusedConv(w, b, dims=2, out_dtype=torch.float32).to(device)   # the tail decodes fp32 scores / deltas
        self._device = device
        return self

    def forward_cl(self, x):
        """x [B,*,*,*,C] channels-last map -> fused head output [B,*,*,*,A*(ncls+7+2)]."""
        if self._device is None:
            self.prepare(x.device)
        return self.fhead(x if x.dtype == self.fhead.dtype else x.to(self.fhead.dtype))

    def forward_single(self, x):
        """x [B,C,H,W] -> (cls_score, bbox_pred, dir_cls_preds) in the reference layout."""
        y = ops.from_channels_last(self.forward_cl(ops.to_channels_last(x.contiguous())), 2)
        o = self._offs
        return y[:, o[0]:o[1]], y[:, o[1]:o[2]], y[:, o[2]:]

    def forward(self, feats):
        outs = [self.forward_single(f) for f in feats]
        return tuple(map(list, zip(*outs)))

    def _anchors(self, H, W, device):
        return self.anchor_generator.grid_anchors([(H, W)], device=device)[0].reshape(-1, self.box_code_size)

    def get_bboxes_cl(self, head_out, H, W, input_metas, cfg=None, hw_transposed=False, want_candidates=False):
        """Device tail on the fused channels-last head output.  Returns the raw device tensors
        (boxes [B,max_num,7], scores, labels, count[, candidates])."""
        cfg = self.test_cfg if cfg is None else cfg
        anchors = self._anchors(H, W, head_out.device)
        return ops.anchor_head_get_bboxes(head_out, anchors, H, W, self.num_anchors, self.num_classes, self._offs, cfg,
                                          self.dir_offset, self.dir_limit_offset, hw_transposed, want_candidates)

    @staticmethod
    def _wrap(boxes, scores, labels, count, input_metas):
        n = count.tolist()          # the one D2H sync of the tail
        res = []
        for b, meta in enumerate(input_metas):
            box_type = meta.get('box_type_3d', LiDARInstance3DBoxes)
            res.append((box_type(boxes[b, :n[b]], box_dim=7), scores[b, :n[b]], labels[b, :n[b]]))
        return res

    def get_bboxes(self, cls_scores, bbox_preds, dir_cls_preds, valid, input_metas, cfg=None, rescale=False):
        """Reference signature (this fork passes `valid`, which is ignored: anchor3d_head.py:375-397).
        Single feature level, as in every ImVoxelNet config."""
        assert len(cls_scores) == len(bbox_preds) == len(dir_cls_preds)
        if len(cls_scores) != 1:
            raise NotImplementedError('multi-level anchor heads are not used by ImVoxelNet and are not built')
        y = torch.cat([cls_scores[0], bbox_preds[0], dir_cls_preds[0]], dim=1).contiguous()
        H, W = y.shape[-2:]
        out = self.get_bboxes_cl(ops.to_channels_last(y), H, W, input_metas, cfg)
        return self._wrap(*out, input_metas)
