"""Import the REAL reference (SamsungLabs/imvoxelnet @ /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Works only in the build container, where
/root/reference exists; it never travels to the GPU box (nothing under
`-m gpu`, `smoke()` or `bench.py` imports this file).  It is used by
`oracle/gen_golden.py` to produce the committed fixtures under
`tests/golden/` and by `tests/test_oracle_vs_reference.py` (skipped when the
reference is absent) to pin the restatement in `oracle/`.

mmcv / mmdet / numba are not installed, so tiny stand-in *Python modules* are
planted in `sys.modules` for the registries and decorators the reference
files touch at import time.  None of the arithmetic on the hot path comes from
those stand-ins: every function we call (get_points, backproject,
_compute_projection, the necks, Anchor3DHead.get_bboxes, the anchor
generator, the box coder, box utils, aligned_3d_nms, box3d_multiclass_nms) is
the reference's own source, loaded from where it lies.
The one thing that cannot run here is the CUDA extension `iou3d_cuda`
(rotated NMS): `nms_gpu` is injected from `oracle/` by the caller.
"""
import importlib.util
import os
import sys
import types

import torch
from torch import nn

REF = os.environ.get('IMVOXELNET_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF, 'mmdet3d'))


class _Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, **default):
        cfg = dict(cfg)
        cfg.update({k: v for k, v in default.items() if k not in cfg})
        t = cfg.pop('type')
        cls = self.module_dict[t] if isinstance(t, str) else t
        return cls(**cfg)


def _mod(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        if '.' in name:
            parent, child = name.rsplit('.', 1)
            setattr(_mod(parent), child, m)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _load(dotted, relpath):
    path = os.path.join(REF, relpath)
    spec = importlib.util.spec_from_file_location(dotted, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[dotted] = m
    if '.' in dotted:
        parent, child = dotted.rsplit('.', 1)
        setattr(_mod(parent), child, m)
    spec.loader.exec_module(m)
    return m


_loaded = {}


def load():
    """Returns a namespace of reference modules (loaded once)."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    if not available():
        raise RuntimeError('reference not present at ' + REF)

    DET, NECKS, HEADS, BACKBONES = (_Registry(n) for n in ('det', 'neck', 'head', 'bb'))
    ANCH, CODERS = _Registry('anchor'), _Registry('coder')

    def ident_deco(*a, **k):
        def d(f):
            return f
        return d

    def multi_apply(func, *args, **kwargs):
        from functools import partial
        pfunc = partial(func, **kwargs) if kwargs else func
        res = map(pfunc, *args)
        return tuple(map(list, zip(*res)))

    class Scale(nn.Module):
        def __init__(self, scale=1.0):
            super().__init__()
            self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

        def forward(self, x):
            return x * self.scale

    def bias_init_with_prob(p):
        import numpy as np
        return float(-np.log((1 - p) / p))

    def normal_init(module, mean=0, std=1, bias=0):
        nn.init.normal_(module.weight, mean, std)
        if getattr(module, 'bias', None) is not None:
            nn.init.constant_(module.bias, bias)

    def is_list_of(seq, t):
        return isinstance(seq, list) and all(isinstance(i, t) for i in seq)

    class BaseBBoxCoder:
        def __init__(self, **kw):
            pass

    _mod('mmcv', is_list_of=is_list_of)
    _mod('mmcv.runner', auto_fp16=ident_deco, force_fp32=ident_deco)
    _mod('mmcv.cnn', Scale=Scale, bias_init_with_prob=bias_init_with_prob, normal_init=normal_init)
    _mod('mmdet')
    _mod('mmdet.models', DETECTORS=DET, NECKS=NECKS, HEADS=HEADS, BACKBONES=BACKBONES,
         build_backbone=BACKBONES.build, build_neck=NECKS.build, build_head=HEADS.build)
    _mod('mmdet.models.builder', HEADS=HEADS, build_loss=lambda cfg: None)
    _mod('mmdet.models.detectors', BaseDetector=nn.Module)
    _mod('mmdet.core', multi_apply=multi_apply, reduce_mean=lambda x: x,
         build_assigner=lambda c: None, build_sampler=lambda c: None,
         build_bbox_coder=CODERS.build, build_anchor_generator=ANCH.build)
    _mod('mmdet.core.anchor', ANCHOR_GENERATORS=ANCH)
    _mod('mmdet.core.bbox', BaseBBoxCoder=BaseBBoxCoder)
    _mod('mmdet.core.bbox.builder', BBOX_CODERS=CODERS)

    class _Numba(types.ModuleType):
        @staticmethod
        def jit(*a, **k):
            if a and callable(a[0]):
                return a[0]
            return lambda f: f
    sys.modules['numba'] = _Numba('numba')

    # placeholders for native ops the reference imports at module level
    _mod('mmdet3d')
    _mod('mmdet3d.ops', points_in_boxes_batch=None, points_in_boxes_gpu=None)
    _mod('mmdet3d.ops.iou3d', iou3d_cuda=None)
    _mod('mmdet3d.ops.iou3d.iou3d_utils', nms_gpu=None, nms_normal_gpu=None)
    _mod('mmdet3d.ops.rotated_iou')
    _mod('mmdet3d.ops.rotated_iou.oriented_iou_loss', cal_giou_3d=None, cal_iou_3d=None)
    _mod('mmdet3d.ops.roiaware_pool3d', points_in_boxes_gpu=None)
    _mod('mmdet3d.core.points', BasePoints=type('BasePoints', (), {}))
    _mod('mmdet3d.core.bbox')
    _mod('mmdet3d.core.bbox.structures')

    utils = _load('mmdet3d.core.bbox.structures.utils', 'mmdet3d/core/bbox/structures/utils.py')
    base = _load('mmdet3d.core.bbox.structures.base_box3d', 'mmdet3d/core/bbox/structures/base_box3d.py')
    lidar = _load('mmdet3d.core.bbox.structures.lidar_box3d', 'mmdet3d/core/bbox/structures/lidar_box3d.py')
    depth = _load('mmdet3d.core.bbox.structures.depth_box3d', 'mmdet3d/core/bbox/structures/depth_box3d.py')
    nms = _load('mmdet3d.core.post_processing.box3d_nms', 'mmdet3d/core/post_processing/box3d_nms.py')
    anchor = _load('mmdet3d.core.anchor.anchor_3d_generator', 'mmdet3d/core/anchor/anchor_3d_generator.py')
    coder = _load('mmdet3d.core.bbox.coders.delta_xyzwhlr_bbox_coder',
                  'mmdet3d/core/bbox/coders/delta_xyzwhlr_bbox_coder.py')

    def bbox3d2result(bboxes, scores, labels):
        return dict(boxes_3d=bboxes.to('cpu'), scores_3d=scores.cpu(), labels_3d=labels.cpu())

    class PseudoSampler:
        pass

    _mod('mmdet3d.core', bbox3d2result=bbox3d2result, PseudoSampler=PseudoSampler,
         box3d_multiclass_nms=nms.box3d_multiclass_nms, limit_period=utils.limit_period,
         xywhr2xyxyr=utils.xywhr2xyxyr, aligned_3d_nms=nms.aligned_3d_nms)
    _mod('mmdet3d.core.bbox.structures', rotation_3d_in_axis=utils.rotation_3d_in_axis,
         limit_period=utils.limit_period, xywhr2xyxyr=utils.xywhr2xyxyr)
    _mod('mmdet3d.core.post_processing', aligned_3d_nms=nms.aligned_3d_nms,
         box3d_multiclass_nms=nms.box3d_multiclass_nms)

    detector = _load('mmdet3d.models.detectors.imvoxelnet', 'mmdet3d/models/detectors/imvoxelnet.py')
    necks = _load('mmdet3d.models.necks.imvoxelnet', 'mmdet3d/models/necks/imvoxelnet.py')

    _mod('mmdet3d.models')
    _mod('mmdet3d.models.builder', build_loss=lambda cfg: None)
    _mod('mmdet3d.models.dense_heads')

    class AnchorTrainMixin:
        pass
    _mod('mmdet3d.models.dense_heads.train_mixins', AnchorTrainMixin=AnchorTrainMixin)
    anchor_head = _load('mmdet3d.models.dense_heads.anchor3d_head', 'mmdet3d/models/dense_heads/anchor3d_head.py')
    _mod('mmdet3d.models.detectors')
    head_v1 = _load('mmdet3d.models.dense_heads.imvoxel_head', 'mmdet3d/models/dense_heads/imvoxel_head.py')
    head_v2 = _load('mmdet3d.models.dense_heads.imvoxel_head_v2', 'mmdet3d/models/dense_heads/imvoxel_head_v2.py')

    _loaded.update(utils=utils, base=base, lidar=lidar, depth=depth, nms=nms, anchor=anchor,
                   coder=coder, detector=detector, necks=necks, anchor_head=anchor_head,
                   head_v1=head_v1, head_v2=head_v2,
                   registries=dict(DET=DET, NECKS=NECKS, HEADS=HEADS, ANCH=ANCH, CODERS=CODERS))
    return types.SimpleNamespace(**_loaded)


if __name__ == '__main__':
    ns = load()
    print('loaded:', [k for k in vars(ns)])
