"""ImVoxelNet detector (inference) under the reference's registry name, constructor kwargs and method
surface (mmdet3d/models/detectors/imvoxelnet.py:8-187): extract_feat / simple_test / forward_test.

Differences in HOW, not WHAT:
  * the whole path runs channels-last on the device through libimvoxel_hip.so; tensors are converted to
    the reference layout only where the public methods hand them to the caller;
  * the per-sample Python loop (:58-76) is one fused launch for the batch; only the tiny per-sample
    camera set-up (:114-129, :139) stays on the host, using the same torch CPU ops as the reference, and is
    uploaded once per batch;
  * FPN levels 1..3 are not computed (only level 0 is consumed, :50).
"""
import torch
from torch import nn

from . import ops
from .boxes import bbox3d2result
from .heads import Anchor3DHead
from .registry import DETECTORS, build_backbone, build_head, build_neck


@torch.no_grad()
def get_points(n_voxels, voxel_size, origin):
    """detectors/imvoxelnet.py:132-141 (host helper kept for API parity; the kernel computes the same
    idx * voxel_size + new_origin in registers)."""
    points = torch.stack(torch.meshgrid([torch.arange(n_voxels[0]), torch.arange(n_voxels[1]),
                                         torch.arange(n_voxels[2])], indexing='ij'))
    new_origin = origin - n_voxels / 2. * voxel_size
    return points * voxel_size.view(3, 1, 1, 1) + new_origin.view(3, 1, 1, 1)


@DETECTORS.register_module()
class ImVoxelNet(nn.Module):
    def __init__(self, backbone, neck, neck_3d, bbox_head, n_voxels, voxel_size, head_2d=None, train_cfg=None,
                 test_cfg=None, pretrained=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck)
        self.neck_3d = build_neck(neck_3d)
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg, test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.bbox_head.voxel_size = voxel_size
        self.head_2d = build_head(head_2d) if head_2d is not None else None      # LayoutHead (SUN RGB-D Total configs)
        self.n_voxels = tuple(int(v) for v in n_voxels)
        self.voxel_size = tuple(float(v) for v in voxel_size)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.storage_dtype, self._prepared_device, self._native = None, None, None
        self.trunk_fp8 = False
        # weights loaded AFTER prepare() (load_state_dict / data.load_checkpoint) must reach the packed device copies: the
        # sub-modules drop theirs (params.invalidate_packed_on_load), and the detector re-packs in the dtype it was prepared in
        self.register_load_state_dict_post_hook(
            lambda mod, keys: mod.prepare(mod._prepared_device, dtype=mod.storage_dtype) if mod._prepared_device is not None else None)
        self.init_weights(pretrained=pretrained)

    def init_weights(self, pretrained=None):
        self.backbone.init_weights(pretrained=pretrained)
        self.neck.init_weights()
        self.neck_3d.init_weights()
        self.bbox_head.init_weights()
        if self.head_2d is not None:
            self.head_2d.init_weights()

    def prepare(self, device, dtype=torch.float32, native=None):
        """Pack every layer's parameters for the device kernels (call again after changing weights).
        native (default: on unless IVX_NATIVE_MODEL=0): build the native model handle (engine.NativeModel over csrc/model.cpp) for every
        family it covers (engine.family) and let simple_test run the whole device side through ONE C-ABI call;
        the layer-by-layer composition below stays available (extract_feat, forward_cl, the other configurations).
        dtype: storage type of activations and weights between layers.  float32 (default) is the reference's precision
        and the one every parity claim is made for; bfloat16 is an optional reduced-precision mode (fp32 accumulate,
        fp32 epilogues, fp32 head output and detection tail) built for the single-view anchor-head configs."""
        from .conv import storage_dtype
        with storage_dtype(dtype):
            for m in (self.backbone, self.neck, self.neck_3d, self.bbox_head):
                m.prepare(device)
            if self.head_2d is not None:
                self.head_2d.prepare(device)
                if hasattr(self.backbone, 'stage_out_pair'):      # the LayoutHead pools C5: that stage output stays fp32 in the pair chain
                    self.backbone.stage_out_pair = (True, True, True, False)
        self.storage_dtype, self._prepared_device, self.trunk_fp8 = dtype, device, False
        import os
        from . import engine
        if native is None:
            native = os.environ.get('IVX_NATIVE_MODEL', '1') != '0'
        if self._native is not None:
            self._native.close()
        self._native = None
        # fp32, or the bf16 storage mode (ivx_model_cfg.storage; not with DCNv2 stages / a LayoutHead, which are fp32-only everywhere)
        bf16_ok = dtype == torch.bfloat16 and self.head_2d is None and not any(
            getattr(blk, 'dcn', False) for i in range(4) for blk in getattr(self.backbone, f'layer{i + 1}', []))
        if native and (dtype == torch.float32 or bf16_ok) and engine.eligible(self):
            self._native = engine.NativeModel(self, device)
        return self

    def calibrate_fp8(self, img, margin=1.0, stages=None, residual='bf16', variant='conv3'):
        """Optional, on top of prepare(device, dtype=torch.bfloat16) (BASELINE config 5: "bf16 with fp8 2D-conv MFMA"): store
        the 2-D trunk's activations and weights as OCP e4m3 bytes.  One bf16 pass over `img` ([B,V,3,H,W] or [N,3,H,W], a
        representative batch) records max |output| of every trunk layer; the layers are then rebuilt with per-tensor activation
        scales amax * margin / 448 and per-output-channel weight scales folded into their epilogues (v_mfma_f32_32x32x16_fp8_fp8,
        fp32 accumulate).  The FPN laterals read the e4m3 stage outputs and produce bf16; everything after the trunk is unchanged.
        residual: 'bf16' (default) keeps the residual stream -- stem, max-pool, every bottleneck's input / shortcut / output -- in
        bf16 and stores only the inside of each bottleneck (conv1 / conv2 outputs, all conv weights of conv2 / conv3) as e4m3, so
        the rounding noise of a block does not ride on through the later ones (FPN level 0 within a few % rms of fp32);
        'fp8' stores every trunk activation as e4m3 (half the bf16 traffic; FPN level 0 ~10 % rms away: a bandwidth stress mode).
        variant (residual='bf16' only): 'conv3' (default since round 6) = e4m3 only where the noise budget allows it -- in ResNet stages 3 and 4
        conv2 writes e4m3 and conv3 runs on the fp8 matrix cores (e4m3 input and filters), everything else stays bf16: FPN level 0 within
        ~2.2 % rms of the fp32 oracle (tools/fp8_noise_budget.py, profiles/r06_config5.md); 'full' = the round-3 mode, every bottleneck's
        conv1 / conv2 outputs and conv2 / conv3 filters e4m3 in all four stages (3.6 %).
        Weights loaded afterwards need a new calibration.  Returns {layer key: amax}."""
        from .conv import FusedConv, storage_dtype, FP8
        if self._prepared_device is None or self.storage_dtype != torch.bfloat16:
            raise RuntimeError('calibrate_fp8 needs prepare(device, dtype=torch.bfloat16) first')
        if self.head_2d is not None or not hasattr(self.backbone, 'forward_image'):
            raise NotImplementedError('the fp8 trunk is built for the plain ResNet + FPN configurations without a LayoutHead')
        dev = self._prepared_device
        x = img.reshape([-1] + list(img.shape)[-3:]).contiguous().to(dev)
        with storage_dtype(torch.bfloat16):          # a fresh bf16 trunk (a second calibration starts from bf16 again)
            self.backbone.prepare(dev)
        FusedConv.calib = {}
        try:
            self.backbone.forward_image(x)
            torch.cuda.synchronize()
            FusedConv.calib_margin = float(margin)
            self.backbone.fp8_stages = stages             # None: all four stages; n: the first n (the rest keep a bf16 residual stream)
            if residual not in ('bf16', 'fp8'):
                raise ValueError("residual must be 'bf16' or 'fp8'")
            self.backbone.fp8_residual = residual
            if variant not in ('conv3', 'full'):
                raise ValueError("variant must be 'conv3' or 'full'")
            c3 = variant == 'conv3' and residual == 'bf16' and stages is None
            self.backbone.fp8_variant = 'conv3' if c3 else 'full'
            self.backbone.fp8_first_stage = 2 if c3 else 0
            with storage_dtype(FP8):
                self.backbone.prepare(dev)
            with storage_dtype(torch.bfloat16):
                self.neck.prepare(dev, in_dtype=[self.backbone.stage_dtypes[i] for i in self.backbone.out_indices])
            calib = dict(FusedConv.calib)
        finally:
            FusedConv.calib, FusedConv.calib_margin = None, 1.0
        self.trunk_fp8 = 'fp8' if residual == 'fp8' else 'fp8-branches'
        if self._native is not None:
            if residual == 'bf16' and stages is None and self._native.cfg.with_trunk:
                try:
                    self._native.calibrate_fp8(x, margin, first_stage=2 if c3 else 0, conv2_bf16=c3)     # the same mode inside the native handle (its own calibration pass: same maxima)
                except ValueError as e:      # the library's invalid-argument status only; a HIP error / OOM (IvxError) propagates (round-5 advisor)
                    # (e.g. calibration images whose H / W are not multiples of 32: ivx_model_calibrate_fp8 refuses them.)  The Python
                    # modules are e4m3 already; a handle left in bf16 would make simple_test run ANOTHER mode than extract_feat without a
                    # word, so the handle goes and the layer-by-layer composition (the calibrated one) serves every call.
                    import warnings
                    self._native.close()
                    self._native = None
                    warnings.warn(f'calibrate_fp8: the native handle could not be calibrated ({e}); simple_test runs the layer-by-layer e4m3 composition',
                                  RuntimeWarning, stacklevel=2)
            else:                                         # forms the handle does not hold: the e4m3 trunk runs layer by layer
                self._native.close()
                self._native = None
        return calib

    # ------------------------------------------------------------------ host-side camera set-up
    @staticmethod
    def _compute_projection(img_meta, stride, angles=None):
        """detectors/imvoxelnet.py:114-129 with the same torch CPU ops.  angles (SUN RGB-D Total test mode): the list
        of predicted (pitch, roll) the reference iterates as if they were views (:121-124; one entry at batch size 1)."""
        intrinsic = torch.tensor(img_meta['lidar2img']['intrinsic'][:3, :3])
        ratio = img_meta['ori_shape'][0] / (img_meta['img_shape'][0] / stride)
        if angles is not None:
            from .heads_layout import layout_extrinsics
            extrinsics = [layout_extrinsics(a).to(intrinsic.device) for a in angles]
        else:
            extrinsics = [torch.tensor(e) for e in img_meta['lidar2img']['extrinsic']]
        if intrinsic.dtype != torch.float32 or any(e.dtype != torch.float32 for e in extrinsics):
            intrinsic[:2] /= ratio                     # non-fp32 metas: the reference's own ops; _camera_setup rejects the result
            return torch.stack([intrinsic @ e[:3] for e in extrinsics])
        # fp32 (what the reference datasets produce): `intrinsic[:2] /= ratio; intrinsic @ extrinsic[:3]` in the library's
        # fixed operation order -- a host matmul picks different fp32 summation orders on different CPUs (1 ulp, which
        # flips the rounded pixel of a few voxels in a million); the fixed order is the one the golden vectors pin
        from .engine import compute_projection
        return compute_projection(intrinsic.numpy(), [e.numpy() for e in extrinsics], ratio)

    def _camera_setup(self, img_metas, stride, device, angles=None):
        proj, orig, crop = [], [], []
        nv = torch.tensor(self.n_voxels)
        vs = torch.tensor(self.voxel_size)
        for b, meta in enumerate(img_metas):
            # the reference hands the whole batch's angle list to every sample (:60), which only works at batch size 1
            # (the SUN RGB-D Total test setting); here sample b gets its own prediction
            p = self._compute_projection(meta, stride, None if angles is None else [angles[b]])
            if p.dtype != torch.float32:
                raise TypeError('lidar2img intrinsic/extrinsic must be float32 (as the reference datasets produce)')
            proj.append(p)
            origin = torch.tensor(meta['lidar2img']['origin'])
            if origin.dtype != torch.float32:
                origin = origin.float()
            orig.append(origin - nv / 2. * vs)                      # :139
            crop.append([meta['img_shape'][0] // stride, meta['img_shape'][1] // stride])   # :67-68
        V = proj[0].shape[0]
        if any(p.shape[0] != V for p in proj):
            raise ValueError('all samples of a batch must have the same number of views')
        return (torch.stack(proj).contiguous().to(device), torch.stack(orig).contiguous().to(device),
                torch.tensor(crop, dtype=torch.int32).to(device))

    # ------------------------------------------------------------------ channels-last fast path
    def features_2d_cl(self, img, img_metas=None, want_2d=False):
        """img [B,V,3,H,W] -> FPN level 0, channels-last [B*V,1,H/4,W/4,Cf]
        (with want_2d: also the LayoutHead output (angles, layouts) computed from C5, or None without a head_2d)."""
        x = img.reshape([-1] + list(img.shape)[2:]).contiguous()
        feats = self.backbone.forward_image(x) if hasattr(self.backbone, 'forward_image') else self.backbone.forward_cl(ops.to_channels_last(x, pad_to=4))
        features_2d = self.head_2d.forward_cl(feats[-1], img_metas) if (want_2d and self.head_2d is not None) else None
        p0 = self.neck.forward_cl(list(feats))[0]
        stride = x.shape[-1] / p0.shape[3]
        assert stride == 4, 'stride of FPN level 0 must be 4 (detectors/imvoxelnet.py:53-54)'
        return (p0, features_2d) if want_2d else p0

    def lift_cl(self, p0, img_metas, angles=None):
        """FPN level 0 [B*V,1,h,w,C] + metas -> (volume [B,X,Y,Z,C], valid [B,X,Y,Z] bool).
        angles: predicted (pitch, roll) list of the LayoutHead (test mode of the Total configs) or None."""
        proj, new_origin, crop = self._camera_setup(img_metas, 4, p0.device, angles)
        return ops.backproject_mean(p0, proj, new_origin, crop, self.voxel_size, self.n_voxels)

    def detect_cl(self, volume, img_metas, want_candidates=False):
        """Anchor-head configs (KITTI / nuScenes): raw device tensors (boxes, scores, labels, count)."""
        y = self.neck_3d.forward_cl(volume)                      # [B,X',Y',1,C]
        h = self.bbox_head.forward_cl(y)                         # [B,X',Y',1,CH]
        # the reference transposes to [B,C,Y',X'] (necks/imvoxelnet.py:120): H = Y', W = X'
        return self.bbox_head.get_bboxes_cl(h, y.shape[2], y.shape[1], img_metas, hw_transposed=True,
                                            want_candidates=want_candidates)

    # ------------------------------------------------------------------ reference surface
    def extract_feat(self, img, img_metas, mode='test'):
        """-> (list of neck outputs in the reference layout, valids [B,1,X,Y,Z] bool, features_2d), features_2d =
        (angles, layouts) of the LayoutHead or None (detectors/imvoxelnet.py:45-80)."""
        p0, features_2d = self.features_2d_cl(img, img_metas, want_2d=True)
        angles = features_2d[0] if features_2d is not None and mode == 'test' else None        # :60
        volume, valid = self.lift_cl(p0, img_metas, angles)
        y = self.neck_3d.forward_cl(volume)
        if isinstance(y, (list, tuple)):                      # indoor necks: multi-level [B,C,X,Y,Z]
            return [ops.from_channels_last(t, 3) for t in y], valid.unsqueeze(1), features_2d
        out = ops.from_channels_last(y, 3)
        return [out[..., 0].transpose(-1, -2)], valid.unsqueeze(1), features_2d

    def detect_indoor_cl(self, volume, valid, img_metas):
        """Anchor-free indoor configs (SUN RGB-D / ScanNet): list of (boxes object, scores, labels)."""
        levels = self.neck_3d.forward_cl(volume)
        return self.bbox_head.get_bboxes_cl(self.bbox_head.forward_cl(levels), valid, img_metas)

    def simple_test(self, img, img_metas, gather=False, global_batch=None):
        """detectors/imvoxelnet.py:93-106.  gather (native-handle families, torch.distributed initialised): every rank passes
        its slice of the batch; ONE all-gather of the fixed-size padded device tensors (dist.all_gather_detections) replaces
        mmdet's pickle-based collect_results after the loop (tools/test.py:131-136), and -- as there -- the collected result list
        (whole batch, rank order) is built and returned on rank 0 only; the other ranks return None, so the host work of a step
        does not grow with the number of ranks.  global_batch (with gather): the size of the batch dist.shard_batch() partitioned, needed when
        it does not divide by the world size (shards that differ by one sample are padded for the fixed-size all-gather)."""
        if self._prepared_device is None and self._native is None and img.is_cuda:
            self.prepare(img.device)                             # first call: pack the weights (and build the native handle)
        H, W = img.shape[-2:]
        native_ok = self._native is not None and H % 32 == 0 and W % 32 == 0 and img.dtype == torch.float32
        if self._native is not None and not native_ok and not getattr(self, '_warned_off_native', False):
            import warnings                              # said once: same results, another (slower) path
            self._warned_off_native = True
            warnings.warn(f'simple_test: input {tuple(img.shape)} {img.dtype} does not go through the native model handle (it takes float32 images whose '
                          'padded height and width are multiples of 32, Pad(size_divisor=32)); running the layer-by-layer composition instead',
                          RuntimeWarning, stacklevel=2)
        if native_ok and self._native.family == 'indoor':
            # the whole of simple_test in ONE native call (ivx_model_detect): trunk [+ LayoutHead -> predicted angles -> projection]
            # -> unprojection -> neck_3d -> anchor-free head -> per-level candidates -> cross-level NMS; camera set-up inside the library
            out = self._native.detect(img.contiguous(), img_metas)
            boxes, scores, labels, count = out[:4]
            if gather and self.head_2d is not None and not getattr(self, '_warned_gather', False):
                import warnings                      # said once: the per-rank result list is returned, nothing is collected
                self._warned_gather = True
                warnings.warn('simple_test(gather=True) is not honoured for models with a head_2d (LayoutHead): its angles / layouts are host tensors '
                              'outside the padded detection block; every rank returns its own results', RuntimeWarning, stacklevel=2)
            if gather and self.head_2d is None:      # sample-sharded ranks: one all-gather of the padded detections, result list on rank 0
                from .dist import all_gather_detections, is_collecting_rank
                boxes, scores, labels, count = all_gather_detections(boxes, scores, labels, count, global_batch=global_batch)
                if not is_collecting_rank():
                    return None
                img_metas = [img_metas[i % len(img_metas)] for i in range(boxes.shape[0])]
            results = self._results_one_copy(boxes, scores, labels, count, img_metas, with_yaw=self.bbox_head.n_reg_outs == 7, indoor=True)
            if self.head_2d is not None:                           # detectors/imvoxelnet.py:101-105
                ang, lay = out[-1]
                angles, layouts = self.head_2d.get_bboxes(list(ang), list(lay), img_metas)
                for i in range(len(results)):
                    results[i]['angles'] = angles[i]
                    results[i]['layout'] = layouts[i]
            return results
        if native_ok and self._native.family == 'levels':
            # indoor necks with a head the handle does not hold: extract_feat in one native call, the head on the op-level ABI
            B, V = img.shape[0], img.shape[1]
            proj, new_origin, crop = self._camera_setup(img_metas, 4, img.device)
            levels, valid = self._native.forward_levels(img.reshape(B * V, 3, H, W).contiguous(), B, V, H, W, proj, new_origin, crop)
            dets = self.bbox_head.get_bboxes_cl(self.bbox_head.forward_cl(levels), valid, img_metas)
            return [bbox3d2result(b, s, l) for b, s, l in dets]
        if native_ok:
            # the whole device side in one native call (csrc/model.cpp); host work: the camera set-up, as the reference
            B, V = img.shape[0], img.shape[1]
            # camera set-up inside the library too (ivx_model_detect): one H2D of the camera block
            boxes, scores, labels, count = self._native.detect(img.contiguous(), img_metas)
            if gather:
                from .dist import all_gather_detections, is_collecting_rank
                boxes, scores, labels, count = all_gather_detections(boxes, scores, labels, count, global_batch=global_batch)
                if not is_collecting_rank():
                    return None                  # as collect_results (tools/test.py:131-136): the collected list exists on rank 0 only
                img_metas = [img_metas[i % len(img_metas)] for i in range(boxes.shape[0])]
            return self._results_one_copy(boxes, scores, labels, count, img_metas)
        p0, features_2d = self.features_2d_cl(img, img_metas, want_2d=True)
        volume, valid = self.lift_cl(p0, img_metas, features_2d[0] if features_2d is not None else None)
        if isinstance(self.bbox_head, Anchor3DHead):
            boxes, scores, labels, count = self.detect_cl(volume, img_metas)
            if gather:
                from .dist import all_gather_detections, is_collecting_rank
                boxes, scores, labels, count = all_gather_detections(boxes, scores, labels, count, global_batch=global_batch)
                if not is_collecting_rank():
                    return None
                img_metas = [img_metas[i % len(img_metas)] for i in range(boxes.shape[0])]     # box type of the other ranks' samples
            if features_2d is None:
                return self._results_one_copy(boxes, scores, labels, count, img_metas)
            dets = self.bbox_head._wrap(boxes, scores, labels, count, img_metas)
        else:
            dets = self.detect_indoor_cl(volume, valid, img_metas)
        results = [bbox3d2result(b, s, l) for b, s, l in dets]
        if features_2d is not None:                            # detectors/imvoxelnet.py:101-105
            angles, layouts = self.head_2d.get_bboxes(*features_2d, img_metas)
            for i in range(len(results)):
                results[i]['angles'] = angles[i]
                results[i]['layout'] = layouts[i]
        return results

    @staticmethod
    def _results_one_copy(boxes, scores, labels, count, img_metas, with_yaw=True, indoor=False):
        """bbox3d2result (core/bbox/transforms.py:49-67) for the fixed-size padded device tensors of the anchor tail: ONE
        packed D2H copy for the whole batch instead of three copies (and syncs) per sample; the box objects are built
        on the host from it."""
        from .boxes import LiDARInstance3DBoxes, DepthInstance3DBoxes
        from .dist import pack_detections, unpack_detections
        blk = getattr(boxes, 'ivx_block', None)
        if blk is not None:          # the native handle wrote the four tensors into one allocation (engine.NativeModel.detect): one copy, no packing kernels
            block, M, o_s, o_l, o_c = blk
            B = boxes.shape[0]
            h = block.cpu()                                                  # the one sync of the step
            b = h[:o_s].view(torch.float32).view(B, M, 7)
            s = h[o_s:o_s + B * M * 4].view(torch.float32).view(B, M)
            l = h[o_l:o_l + B * M * 8].view(torch.int64).view(B, M)
            c = h[o_c:o_c + B * 4].view(torch.int32)
        else:
            packed = pack_detections(boxes, scores, labels, count).cpu()     # the one sync of the step
            b, s, l, c = unpack_detections(packed, scores.shape[1])
        res = []
        for i, meta in enumerate(img_metas):
            n = int(c[i])
            box_type = meta.get('box_type_3d', DepthInstance3DBoxes if indoor else LiDARInstance3DBoxes)
            res.append(dict(boxes_3d=box_type(b[i, :n], box_dim=7, with_yaw=bool(with_yaw)), scores_3d=s[i, :n].clone(), labels_3d=l[i, :n].clone()))
        return res

    def simple_test_view_sharded(self, img, img_metas, group=None, exchange='auto'):
        """simple_test with the VIEWS of the scene(s) sharded over the ranks of `group` (second mode): every
        rank computes the 2-D features and the partial unprojection of its views; then
          exchange='all_reduce': one all-reduce of the partial volume sums and view counts (RCCL), the 3-D neck / head / NMS run
            replicated;
          exchange='reduce_scatter' (stack necks + Anchor3DHead: the nuScenes family): every rank receives the totals of ITS x-slab of
            the volume (+ the neck's receptive field as a halo), normalises and convolves that slab only, and the cropped neck outputs
            are all-gathered (dist.view_sharded_neck_slabs) -- half the bytes on the wire, the neck divided by the ranks too;
          'auto': reduce_scatter where it applies and there is more than one rank.
        Every rank returns the full result.  Same output as simple_test up to fp32 rounding (the order of the view sum; with the slab
        form also the alignment of the neck's Winograd tiles)."""
        from .dist import view_sharded_lift, view_sharded_neck_slabs, _rank_world
        from .necks3d import _StackNeck
        slab_ok = isinstance(self.neck_3d, _StackNeck) and isinstance(self.bbox_head, Anchor3DHead) and self.head_2d is None
        if exchange == 'auto':
            world = _rank_world(None, None, group)[1]
            exchange = 'reduce_scatter' if (slab_ok and world > 1) else 'all_reduce'
            if exchange == 'reduce_scatter':
                # the slab plan has its own preconditions (x extent a multiple of the neck's x stride, no more ranks than output rows): 'auto'
                # falls back to the all-reduce form where they fail -- pure index arithmetic, the same decision on every rank
                from .dist import StackNeckSlabs
                try:
                    StackNeckSlabs(self.neck_3d, int(self.n_voxels[0]), world, world - 1)
                except ValueError:
                    exchange = 'all_reduce'
        if exchange == 'reduce_scatter':
            if not slab_ok:
                raise NotImplementedError('the reduce-scatter exchange is built for the stack necks (Kitti / NuScenes) with an Anchor3DHead')
            y = view_sharded_neck_slabs(self, img, img_metas, group=group)
            h = self.bbox_head.forward_cl(y)
            boxes, scores, labels, count = self.bbox_head.get_bboxes_cl(h, y.shape[2], y.shape[1], img_metas, hw_transposed=True)
            dets = self.bbox_head._wrap(boxes, scores, labels, count, img_metas)
            return [bbox3d2result(b, s, l) for b, s, l in dets]
        if exchange != 'all_reduce':
            raise ValueError("exchange must be 'auto', 'all_reduce' or 'reduce_scatter'")
        volume, valid = view_sharded_lift(self, img, img_metas, group=group)
        if isinstance(self.bbox_head, Anchor3DHead):
            boxes, scores, labels, count = self.detect_cl(volume, img_metas)
            dets = self.bbox_head._wrap(boxes, scores, labels, count, img_metas)
        else:
            dets = self.detect_indoor_cl(volume, valid, img_metas)
        return [bbox3d2result(b, s, l) for b, s, l in dets]

    def forward_test(self, img, img_metas, **kwargs):
        return self.simple_test(img, img_metas, gather=bool(kwargs.get('gather', False)))

    def forward(self, img, img_metas, return_loss=False, **kwargs):
        if return_loss:
            raise NotImplementedError('training (forward_train / losses) is outside the built path')
        return self.forward_test(img, img_metas, **kwargs)

    def aug_test(self, imgs, img_metas):
        pass

    def show_results(self, *args, **kwargs):
        pass
