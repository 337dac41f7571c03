"""Python caller of the native model handle (include/imvoxel.h "model handle", csrc/model.cpp).

For the anchor-head families (KITTI / nuScenes with a plain ResNet-50) the whole device side of
ImVoxelNet.simple_test -- layer sequence, weight packing, Winograd / tile selection, workspace planning, execution --
lives in libimvoxel_hip.so; this module only feeds it the reference state dict, the per-batch camera set-up and
caller-owned buffers.  For the indoor families (FastIndoorImVoxelNeck / ImVoxelNeck without a LayoutHead) the handle
covers extract_feat (trunk + unprojection + neck_3d -> three levels); the anchor-free head and its tail stay on the
op-level C-ABI.  The layer-by-layer Python composition (backbones.py / necks3d.py / heads.py over the op-level
C-ABI) remains for the other configurations and as the cross-check: both run the same kernels with the same plans, so
their results are bit-identical (tests/test_gpu_engine.py).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ModelCfg, TraceRec, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def family(model):
    """Which native handle covers this module, or None:
      'anchor'  ResNet-50 (plain or DCNv2 stages) + FPN, Kitti / NuScenes stack neck, Anchor3DHead with one anchor range: the whole
                simple_test device side (ivx_model_forward / ivx_model_detect);
      'indoor'  ResNet-50 + FPN + FastIndoorImVoxelNeck / ImVoxelNeck + an anchor-free head without conv towers (n_convs = 0, every
                reference config), optionally a LayoutHead: the whole simple_test (ivx_model_detect), or extract_feat alone
                (ivx_model_forward_levels);
      'levels'  the same necks with a head the handle does not hold (V1 heads with towers): trunk + unprojection + neck_3d only."""
    from .backbones import ResNet, FPN
    from .heads import Anchor3DHead
    from .heads_indoor import _ImVoxelHeadBase
    from .heads_layout import LayoutHead
    from .necks3d import KittiImVoxelNeck, NuScenesImVoxelNeck, FastIndoorImVoxelNeck, ImVoxelNeck
    bb = model.backbone
    if not (isinstance(bb, ResNet) and isinstance(model.neck, FPN)):
        return None
    if model.head_2d is not None and not (isinstance(model.head_2d, LayoutHead) and model.head_2d.angle_mlp[0].weight.shape[1] == 2048):
        return None
    if bb.num_stages != 4 or [len(getattr(bb, f'layer{i + 1}')) for i in range(4)] != [3, 4, 6, 3] or tuple(bb.out_indices) != (0, 1, 2, 3):
        return None
    n3 = model.neck_3d
    if isinstance(n3, (FastIndoorImVoxelNeck, ImVoxelNeck)):
        if isinstance(n3, FastIndoorImVoxelNeck) and n3.n_scales != 3:
            return None
        if isinstance(n3, ImVoxelNeck) and len(n3.model.channels) not in (3, 4):
            return None
        h = model.bbox_head
        tc = getattr(h, 'test_cfg', None)
        n_levels = 3 if isinstance(n3, FastIndoorImVoxelNeck) else len(n3.model.channels) - 1
        ok = (isinstance(h, _ImVoxelHeadBase) and h.n_convs == 0 and tc is not None and 0 < int(tc.get('nms_pre', 0)) <= 65536
              and h.n_scales >= n_levels and (h.n_reg_outs == 6 or h.n_classes <= 64))
        if ok:
            return 'indoor'
        return 'levels' if model.head_2d is None else None
    if model.head_2d is not None:
        return None
    if not isinstance(n3, (KittiImVoxelNeck, NuScenesImVoxelNeck)) or not isinstance(model.bbox_head, Anchor3DHead):
        return None
    g = model.bbox_head.anchor_generator
    ok = len(g.ranges) == 1 and len(g.sizes) <= 4 and len(g.rotations) <= 4 and not g.custom_values and g.scales == [1]
    return 'anchor' if ok else None


def eligible(model):
    return family(model) is not None


def model_cfg(model, with_trunk=True, winograd=None, winograd_tile=None):
    """ivx_model_cfg of an ImVoxelNet module (host-only: no device or library call)."""
    from .conv import FusedConv
    from .necks3d import KittiImVoxelNeck, FastIndoorImVoxelNeck, BasicBlock3d
    fam = family(model)
    if fam is None:
        raise NotImplementedError('the native model handle covers ResNet-50 (+ DCNv2 stages) + FPN with a Kitti / NuScenes neck + Anchor3DHead, or '
                                  'with FastIndoorImVoxelNeck / ImVoxelNeck + an anchor-free head without towers (+ LayoutHead), fp32')
    head, n3, bb = model.bbox_head, model.neck_3d, model.backbone
    cfg = ModelCfg()
    cfg.with_trunk = int(bool(with_trunk))
    cfg.fpn_channels = model.neck.out_channels
    cfg.n_voxels[:] = list(model.n_voxels)
    cfg.voxel_size[:] = list(model.voxel_size)
    if with_trunk:
        cfg.dcn_stages[:] = [int(any(getattr(blk, 'dcn', False) for blk in getattr(bb, f'layer{i + 1}'))) for i in range(4)]
        if model.head_2d is not None:
            cfg.layout_head, cfg.layout_linear_size = 1, int(model.head_2d.angle_mlp[0].weight.shape[0])
    if fam == 'anchor':
        tc, g = head.test_cfg, head.anchor_generator
        cfg.neck_type = 0 if isinstance(n3, KittiImVoxelNeck) else 1
        cfg.neck_out_channels = head.in_channels
        cfg.num_classes = head.num_classes
        cfg.n_sizes, cfg.n_rotations = len(g.sizes), len(g.rotations)
        cfg.anchor_range[:] = [float(v) for v in g.ranges[0]]
        for i, s in enumerate(g.sizes):
            cfg.anchor_sizes[3 * i:3 * i + 3] = [float(v) for v in s]
        cfg.anchor_rotations[:len(g.rotations)] = [float(v) for v in g.rotations]
        cfg.nms_pre, cfg.max_num = int(tc['nms_pre']), int(tc['max_num'])
        cfg.use_rotate_nms = int(bool(tc['use_rotate_nms']))
        cfg.score_thr, cfg.nms_thr = float(tc.get('score_thr', 0)), float(tc['nms_thr'])
        cfg.dir_offset, cfg.dir_limit_offset = float(head.dir_offset), float(head.dir_limit_offset)
    else:
        if isinstance(n3, FastIndoorImVoxelNeck):
            cfg.neck_type = 2
            cfg.neck_out_channels = n3.out_block_0[0].weight.shape[0]
            cfg.fast_n_blocks[:] = [len(getattr(n3, f'down_layer_{i}')) for i in range(3)]
        else:
            cfg.neck_type = 3
            cfg.neck_out_channels = n3.conv_blocks[0][0].weight.shape[0]
            pad = lambda v, n: list(v) + [0] * (n - len(v))        # a 3-scale U-Net leaves the last entries 0
            cfg.unet_channels[:] = pad(n3.model.channels, 4)
            cfg.unet_down_layers[:] = pad([sum(isinstance(b, BasicBlock3d) for b in layer) for layer in n3.model.layers_down], 4)
            cfg.unet_up_layers[:] = pad([len(seq) for seq in n3.model.layers_up_res], 3)
        if fam == 'indoor':
            tc = head.test_cfg
            cfg.head_type = 1 if head.n_reg_outs == 6 else 2
            cfg.head_classes, cfg.head_nms_pre = int(head.n_classes), int(tc['nms_pre'])
            cfg.head_score_thr = float(tc.get('score_thr', 0))
            cfg.head_nms_thr = float(tc['iou_thr'] if head.n_reg_outs == 6 else tc['nms_thr'])
            cfg.head_use_rotate_nms = int(bool(tc.get('use_rotate_nms', False)))
    cfg.winograd = int(FusedConv.winograd if winograd is None else winograd)
    cfg.winograd_tile = int(FusedConv.winograd_tile if winograd_tile is None else winograd_tile)
    cfg.wino_operands = int(FusedConv.wino_operands)
    cfg.trunk_operands = int(FusedConv.trunk_operands) if with_trunk else 0
    cfg.storage = 1 if getattr(model, 'storage_dtype', None) == torch.bfloat16 else 0       # IVX_BF16: the optional reduced-precision mode
    return cfg


class NativeModel:
    """ivx_model handle built from an ImVoxelNet module (its config and its state dict)."""

    def __init__(self, model, device, with_trunk=True, winograd=None, winograd_tile=None):
        self.family = family(model)
        cfg = model_cfg(model, with_trunk, winograd, winograd_tile)
        self.device = torch.device(device)
        L = self.L = _lib.lib()
        head = model.bbox_head
        g = getattr(head, 'anchor_generator', None)
        self.cfg = cfg
        self.act_dtype = torch.bfloat16 if cfg.storage == 1 else torch.float32      # type of the sub-path tensors (fpn0, volume, levels)
        self.max_num, self.n_voxels = cfg.max_num, tuple(model.n_voxels)
        self.has_head = self.family == 'anchor' or cfg.head_type != 0          # ivx_model_detect runs the whole simple_test
        self.layout = bool(cfg.layout_head)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(L.ivx_create(C.byref(cfg), C.byref(h)), 'ivx_create')
            self.h = h
            for key, t in model.state_dict().items():
                if not t.dtype.is_floating_point:
                    continue                                   # num_batches_tracked
                if not with_trunk and (key.startswith('backbone.') or key.startswith('neck.')):
                    continue
                a = t.detach().to('cpu', torch.float32).contiguous()
                shape = (C.c_int64 * max(a.dim(), 1))(*a.shape)
                check(L.ivx_weights_load(h, key.encode(), C.c_void_p(a.data_ptr()), shape, a.dim()), f'ivx_weights_load({key})')
            check(L.ivx_weights_finalize(h, _stream()), 'ivx_weights_finalize')
            if self.family == 'anchor':
                # the anchor grid: the generator's own output (same torch ops as the reference's CPU path), not the built-in one
                X, Y, Cn = C.c_int32(), C.c_int32(), C.c_int32()
                check(L.ivx_neck3d_out_dims(h, 1, C.byref(X), C.byref(Y), C.byref(Cn)), 'ivx_neck3d_out_dims')
                self.grid_hw = (Y.value, X.value)                  # the reference's (H, W) = (Y', X') (necks/imvoxelnet.py:120)
                anc = g.grid_anchors([self.grid_hw], device='cpu')[0].reshape(-1, 7).contiguous().float()
                shape = (C.c_int64 * 2)(*anc.shape)
                check(L.ivx_weights_load(h, b'anchors', C.c_void_p(anc.data_ptr()), shape, 2), 'ivx_weights_load(anchors)')
            else:
                dims = ((C.c_int32 * 4) * 3)()
                check(L.ivx_neck3d_levels(h, 1, dims), 'ivx_neck3d_levels')
                self.level_dims = [tuple(d) for d in dims if d[3] > 0]    # (X, Y, Z, C) per level, finest first
        self._ws = {}

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.L.ivx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ whole path
    def _workspace(self, key, nbytes):
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws[key] = torch.empty((max(int(nbytes), 256),), device=self.device, dtype=torch.uint8)
        return ws

    def forward(self, x, B, V, H, W, proj, new_origin, crop_hw, want_valid=False):
        """x: image batch [B*V,3,H,W] (with_trunk) or FPN level-0 maps [B*V,1,H/4,W/4,Cf]; -> (boxes [B,max_num,7], scores,
        labels int64, count int32[, valid bool [B,X,Y,Z]]) device tensors."""
        L = self.L
        for t, nm in ((x, 'input'), (proj, 'proj'), (new_origin, 'new_origin')):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                raise ValueError(f'{nm} must be a contiguous float32 device tensor')
        n = L.ivx_model_workspace_bytes(self.h, B, V, H, W)
        if n < 0:
            check(-1, 'ivx_model_workspace_bytes')
        ws = self._workspace('fwd', n)
        dev, M = x.device, self.max_num

        def outputs():
            return (torch.empty((B, M, 7), device=dev, dtype=torch.float32), torch.empty((B, M), device=dev, dtype=torch.float32),
                    torch.empty((B, M), device=dev, dtype=torch.int64), torch.empty((B,), device=dev, dtype=torch.int32),
                    torch.empty((B,) + self.n_voxels, device=dev, dtype=torch.uint8) if want_valid else None)

        def call(xi, pj, no, cr, out, stream):
            check(L.ivx_model_forward(self.h, C.c_void_p(xi.data_ptr()), B, V, H, W, C.c_void_p(pj.data_ptr()), C.c_void_p(no.data_ptr()),
                                      C.c_void_p(cr.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(out[0].data_ptr()),
                                      C.c_void_p(out[1].data_ptr()), C.c_void_p(out[2].data_ptr()), C.c_void_p(out[3].data_ptr()),
                                      C.c_void_p(out[4].data_ptr()) if out[4] is not None else None, stream), 'ivx_model_forward')

        out = outputs()
        call(x, proj, new_origin, crop_hw, out, _stream())
        return (out[0], out[1], out[2], out[3], out[4].view(torch.bool)) if want_valid else out[:4]

    def detect(self, img, img_metas, want_valid=False):
        """simple_test in ONE native call (ivx_model_detect), every family with a head: img [B,V,3,H,W] device tensor + the reference's
        img_metas -> (boxes [B,M,7] rows of the box object's tensor, scores [B,M], labels int64 [B,M], count int32 [B]) device
        tensors [+ valid bool [B,X,Y,Z]] [+ (angles [B,2], layouts [B,7]) host tensors with a LayoutHead].  The camera set-up is
        computed inside the library from the metas' intrinsic / extrinsic / origin / img_shape / ori_shape."""
        import numpy as np
        from ._lib import SampleMeta
        L = self.L
        B, V, _, H, W = img.shape
        x = img.reshape(B * V, 3, H, W)
        if not (x.is_cuda and x.is_contiguous() and x.dtype == torch.float32):
            raise ValueError('img must be a contiguous float32 device tensor')
        metas = (SampleMeta * B)()
        keep = []
        for b, meta in enumerate(img_metas):
            K = np.zeros((4, 4), np.float32)
            Ki = np.asarray(meta['lidar2img']['intrinsic'])
            if Ki.dtype != np.float32:
                raise TypeError('lidar2img intrinsic/extrinsic must be float32 (as the reference datasets produce)')
            K[:Ki.shape[0], :Ki.shape[1]] = Ki
            metas[b].intrinsic[:] = K.reshape(-1).tolist()
            if not self.layout:
                ex = list(meta['lidar2img']['extrinsic'])
                if len(ex) != V or any(np.asarray(e).dtype != np.float32 for e in ex):
                    raise ValueError('every sample needs V float32 extrinsics')
                E = np.zeros((V, 4, 4), np.float32)
                for v, e in enumerate(ex):
                    e = np.asarray(e)
                    E[v, :e.shape[0], :e.shape[1]] = e
                keep.append(E)
                metas[b].extrinsics = E.ctypes.data
            metas[b].origin[:] = [float(v) for v in np.asarray(meta['lidar2img']['origin'], np.float32)]
            metas[b].img_h, metas[b].img_w, metas[b].ori_h = int(meta['img_shape'][0]), int(meta['img_shape'][1]), int(meta['ori_shape'][0])
        n = L.ivx_model_detect_workspace_bytes(self.h, B, V, H, W)
        M = L.ivx_model_max_detections(self.h, B, V, H, W)
        if n < 0 or M < 0:
            check(-1, 'ivx_model_detect_workspace_bytes')
        ws = self._workspace('fwd', n)
        dev = x.device
        # the four caller-owned detection buffers as views of ONE allocation: the host side of simple_test copies that block back in one
        # D2H transfer without packing kernels (detector._results_one_copy; round 6: the torch.cat / cast launches were ~40 us of a 15 ms step)
        n_b, n_s, n_l = B * M * 7 * 4, B * M * 4, B * M * 8
        o_s, o_l, o_c = n_b, (n_b + n_s + 7) // 8 * 8, 0
        o_c = o_l + n_l
        block = torch.empty((o_c + B * 4,), device=dev, dtype=torch.uint8)
        out = (block[:n_b].view(torch.float32).view(B, M, 7), block[o_s:o_s + n_s].view(torch.float32).view(B, M),
               block[o_l:o_l + n_l].view(torch.int64).view(B, M), block[o_c:o_c + B * 4].view(torch.int32).view(B))
        out[0].ivx_block = (block, M, o_s, o_l, o_c)
        valid = torch.empty((B,) + self.n_voxels, device=dev, dtype=torch.uint8) if want_valid else None
        ang = torch.empty((B, 2), dtype=torch.float32) if self.layout else None
        lay = torch.empty((B, 7), dtype=torch.float32) if self.layout else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(L.ivx_model_detect(self.h, p(x), B, V, H, W, C.cast(metas, C.c_void_p), p(ws), ws.numel(), p(out[0]), p(out[1]), p(out[2]), p(out[3]),
                                 p(valid), p(ang), p(lay), _stream()), 'ivx_model_detect')
        res = out + ((valid.view(torch.bool),) if want_valid else ())
        return res + ((ang, lay),) if self.layout else res

    def _level_buffers(self, B, dev):
        outs = [torch.empty((B, X, Y, Z, Cn), device=dev, dtype=self.act_dtype) for X, Y, Z, Cn in self.level_dims]
        return outs, (C.c_void_p * 3)(*([o.data_ptr() for o in outs] + [None] * (3 - len(outs))))

    def forward_levels(self, x, B, V, H, W, proj, new_origin, crop_hw):
        """Indoor families: x as for forward -> ([level0, level1, level2] channels-last [B,X_l,Y_l,Z_l,Cout] finest first,
        valid bool [B,X,Y,Z]) -- extract_feat of detectors/imvoxelnet.py:43-88 in one native call."""
        L = self.L
        for t, nm in ((x, 'input'), (proj, 'proj'), (new_origin, 'new_origin')):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                raise ValueError(f'{nm} must be a contiguous float32 device tensor')
        n = L.ivx_model_workspace_bytes(self.h, B, V, H, W)
        if n < 0:
            check(-1, 'ivx_model_workspace_bytes')
        ws = self._workspace('fwd', n)
        outs, ptrs = self._level_buffers(B, x.device)
        valid = torch.empty((B,) + self.n_voxels, device=x.device, dtype=torch.uint8)
        check(L.ivx_model_forward_levels(self.h, C.c_void_p(x.data_ptr()), B, V, H, W, C.c_void_p(proj.data_ptr()), C.c_void_p(new_origin.data_ptr()),
                                         C.c_void_p(crop_hw.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), ptrs, C.c_void_p(valid.data_ptr()),
                                         _stream()), 'ivx_model_forward_levels')
        return outs, valid.view(torch.bool)

    def neck3d_levels(self, volume):
        """volume [B,X,Y,Z,Cf] -> the three neck levels (channels-last, finest first)."""
        B = volume.shape[0]
        n = self.L.ivx_neck3d_workspace_bytes(self.h, B)
        if n < 0:
            check(-1, 'ivx_neck3d_workspace_bytes')
        ws = self._workspace('neck', n)
        outs, ptrs = self._level_buffers(B, volume.device)
        fn = self.L.ivx_neck3d_fast_fwd if self.cfg.neck_type == 2 else self.L.ivx_neck3d_unet_fwd
        check(fn(self.h, C.c_void_p(volume.data_ptr()), B, ptrs, C.c_void_p(ws.data_ptr()), ws.numel(), _stream()), 'ivx_neck3d_levels_fwd')
        return outs

    def calibrate_fp8(self, img, margin=1.0, first_stage=0, conv2_bf16=False):
        """bf16 handle: one bf16 pass of the trunk over img [BV,3,H,W] fp32 records the maxima of the bottleneck interiors, which are
        stored as e4m3 from then on (ivx_model_calibrate_fp8: BASELINE config 5's "bf16 with fp8 2-D conv MFMA" inside the C-ABI)."""
        BV, _, H, W = img.shape
        n = self.L.ivx_backbone_fpn_workspace_bytes(self.h, BV, H, W)
        if n < 0:
            check(-1, 'ivx_backbone_fpn_workspace_bytes')
        ws = self._workspace('trunk', n)
        check(self.L.ivx_model_calibrate_fp8_ex(self.h, C.c_void_p(img.data_ptr()), BV, H, W, C.c_float(float(margin)), int(first_stage), int(bool(conv2_bf16)),
                                                C.c_void_p(ws.data_ptr()), C.c_int64(ws.numel()), _stream()), 'ivx_model_calibrate_fp8_ex')
        self._ws = {}            # plans (and workspace sizes) are rebuilt on the next call

    # ------------------------------------------------------------------ sub-paths
    def backbone_fpn(self, img):
        """img [BV,3,H,W] -> FPN level 0 [BV,1,H/4,W/4,Cf] channels-last."""
        BV, _, H, W = img.shape
        n = self.L.ivx_backbone_fpn_workspace_bytes(self.h, BV, H, W)
        if n < 0:
            check(-1, 'ivx_backbone_fpn_workspace_bytes')
        ws = self._workspace('trunk', n)
        out = torch.empty((BV, 1, H // 4, W // 4, self.cfg.fpn_channels), device=img.device, dtype=self.act_dtype)
        check(self.L.ivx_backbone_fpn_fwd(self.h, C.c_void_p(img.data_ptr()), BV, H, W, C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                                          ws.numel(), _stream()), 'ivx_backbone_fpn_fwd')
        return out

    def neck3d(self, volume):
        """volume [B,X,Y,Z,Cf] -> [B,X',Y',1,Cout]."""
        B = volume.shape[0]
        n = self.L.ivx_neck3d_workspace_bytes(self.h, B)
        if n < 0:
            check(-1, 'ivx_neck3d_workspace_bytes')
        ws = self._workspace('neck', n)
        out = torch.empty((B, self.grid_hw[1], self.grid_hw[0], 1, self.cfg.neck_out_channels), device=volume.device, dtype=self.act_dtype)
        fn = self.L.ivx_neck3d_kitti_fwd if self.cfg.neck_type == 0 else self.L.ivx_neck3d_nuscenes_fwd
        check(fn(self.h, C.c_void_p(volume.data_ptr()), B, C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), _stream()),
              'ivx_neck3d_fwd')
        return out

    # ------------------------------------------------------------------ stage timing
    def trace(self, level=2):
        """0 / False: off; 2 / True: an event pair around every launch group; 1: coarse -- the neck stages, the unprojection and
        the tail individually, the 2-D trunk as one span (stage 6): a third of the events; 3: only the grouped Winograd-domain GEMM launches
        (stage 2) -- what bench.py keeps inside its timed region."""
        level = 2 if level is True else int(level)
        check(self.L.ivx_model_trace(self.h, level), 'ivx_model_trace')

    def trace_records(self):
        """After torch.cuda.synchronize(): list of dicts (step, stage, is3d, ms, flops, bytes, name) in launch order;
        stage 0 direct conv, 1 Winograd input transform, 2 grouped GEMM, 3 output transform, 4 unprojection, 5 tail, 6 the
        2-D trunk as one span (coarse level)."""
        out, rec = [], TraceRec()
        for i in range(self.L.ivx_model_trace_count(self.h)):
            check(self.L.ivx_model_trace_read(self.h, i, C.byref(rec)), 'ivx_model_trace_read')
            out.append(dict(step=rec.step, stage=rec.stage, is3d=bool(rec.is3d), ms=rec.ms, start_ms=rec.start_ms, flops=rec.flops, bytes=rec.bytes,
                            name=rec.name.decode()))
        return out


def compute_projection(intrinsic, extrinsics, ratio):
    """(K[:3,:3] with rows 0,1 / ratio) @ E_v[:3] for every view, in the library's fixed fp32 operation order
    (ivx_compute_projection; detectors/imvoxelnet.py:114-129).  numpy / torch in, float32 torch tensor [V,3,4] out."""
    import numpy as np
    K = np.ascontiguousarray(np.asarray(intrinsic, dtype=np.float32))
    if K.shape != (4, 4):
        K4 = np.eye(4, dtype=np.float32)
        K4[:K.shape[0], :K.shape[1]] = K
        K = K4
    E = np.ascontiguousarray(np.stack([np.asarray(e, dtype=np.float32) for e in extrinsics]))
    if E.shape[1:] != (4, 4):
        E4 = np.zeros((E.shape[0], 4, 4), np.float32)
        E4[:, :E.shape[1], :E.shape[2]] = E
        E = E4
    P = np.empty((E.shape[0], 3, 4), np.float32)
    check(_lib.lib().ivx_compute_projection(K.ctypes.data_as(C.c_void_p), E.ctypes.data_as(C.c_void_p), E.shape[0], float(ratio),
                                            P.ctypes.data_as(C.c_void_p)), 'ivx_compute_projection')
    return torch.from_numpy(P)
