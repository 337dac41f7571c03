"""-m gpu: the CHAINED path (image -> detections, every stage fed by this library's own previous stage, exactly what
simple_test runs) against the CPU oracle at the BASELINE sizes -- kept (level, voxel, class) / anchor indices identical --
and BASELINE config 5 in the precision mode it NAMES ("bf16 with fp8 2D-conv MFMA", 50 views 3x480x640 -> 80x80x32) against
the FP32 ORACLE: per-stage relative errors asserted at their measured values plus a detection-level criterion.

  config 3  ScanNet fast, 20 views (BASELINE's count; the reference's test pipeline uses 50: tests/test_gpu_model.py)
  config 1  SUN RGB-D fast, 1 view
  config 5  ScanNet v1 (Atlas neck, V1 head), 50 views, 80x80x32: fp32 chained, bf16, bf16 + fp8 2-D conv, bf16 + fp8 storage
  config 2  KITTI batch 4: chained kept anchor indices
The stage-isolated full-size tests (neck / head / NMS fed with the ORACLE's volume) stay in test_gpu_model.py /
test_gpu_configs.py; here nothing is substituted.
"""
import numpy as np
import pytest
import torch

from gpu_util import match_rows, assert_same_kept
import kitti_cfg as kc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ia():
    import imvoxelnet_amd
    from imvoxelnet_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return imvoxelnet_amd


def _cpu_sd(model):
    return {k: v.detach().cpu() for k, v in model.state_dict().items()}


def _indoor_case(ia, cfg_name, V, head_seed=5, cls_gain=1.0, cls_bias=-2.0):
    """Model + inputs + the oracle's stage outputs and detections (torch-CPU fp32 / C restatement) of one indoor config."""
    from oracle import imvoxel_oracle as orc
    if cfg_name == 'scannet_fast':
        mcfg, tcfg, n_reg, w = kc.scannet_fast_model_cfg(), kc.SCANNET_FAST_TEST_CFG, 6, (0.001, 0.0005, 0.0002)
        meta = kc.indoor_meta(V, box_type=ia.DepthInstance3DBoxes)
    elif cfg_name == 'sunrgbd_fast':
        mcfg, tcfg, n_reg, w = kc.sunrgbd_fast_model_cfg(), kc.SUNRGBD_FAST_TEST_CFG, 7, (0.001, 0.0005, 0.0002)
        meta = kc.indoor_meta(1, origin=(0, 3, -1), box_type=ia.DepthInstance3DBoxes)
    else:
        mcfg, tcfg, n_reg, w = kc.scannet_v1_model_cfg(), kc.SCANNET_V1_TEST_CFG, 6, (0.01, 0.005, 0.002)
        meta = kc.indoor_meta(V, box_type=ia.DepthInstance3DBoxes)
    model = ia.build_detector(mcfg, test_cfg=tcfg)
    ia.randomize_(model, 77 if cfg_name != 'scannet_v1' else 78)
    with torch.no_grad():
        g = torch.Generator().manual_seed(head_seed)
        model.bbox_head.cls_conv.weight.normal_(0, w[0] * cls_gain, generator=g)
        model.bbox_head.cls_conv.bias.fill_(cls_bias)
        model.bbox_head.centerness_conv.weight.normal_(0, w[1], generator=g)
        model.bbox_head.reg_conv.weight.normal_(0, w[2], generator=g)
    img = torch.randn(1, V, 3, 480, 640, generator=torch.Generator().manual_seed(13))
    sd = _cpu_sd(model)
    nv, vs = mcfg['n_voxels'], mcfg['voxel_size']
    with torch.no_grad():
        f0 = orc.fpn_level0(orc.resnet50(img[0], sd), sd)
        vol_ref, ok_ref = orc.extract_volume(f0.numpy(), meta, nv, vs)
        sdn = {k[len('neck_3d.'):]: v for k, v in sd.items() if k.startswith('neck_3d.')}
        sdh = {k[len('bbox_head.'):]: v for k, v in sd.items() if k.startswith('bbox_head.')}
        if cfg_name == 'scannet_v1':
            nk = mcfg['neck_3d']
            lv = orc.atlas_neck(torch.from_numpy(vol_ref)[None], sdn, nk['channels'], nk['down_layers'], nk['up_layers'])
            cs, bs, ss = orc.fcos_head_forward(lv, sdh, n_reg, n_convs=0)
        else:
            lv = orc.fast_indoor_neck(torch.from_numpy(vol_ref)[None], sdn)
            cs, bs, ss = orc.fcos_head_forward(lv, sdh, n_reg)
        rb, rs, rl, (ocb, ocs, oci) = orc.fcos_get_bboxes_single([c[0] for c in cs], [b[0] for b in bs], [s[0] for s in ss],
                                                                 torch.from_numpy(ok_ref).float(), meta['lidar2img']['origin'], vs, n_reg,
                                                                 tcfg, return_candidates=True)
    ref = dict(f0=f0, vol=vol_ref, valid=ok_ref, levels=lv, boxes=rb, scores=rs, labels=rl, cand_boxes=ocb, cand_scores=ocs, cand_index=oci)
    return model, img, meta, ref


def _chained_indoor(model, img, meta):
    """The stages of simple_test one by one on this library's own intermediates, keeping what the parity checks need."""
    p0 = model.features_2d_cl(img)
    vol, valid = model.lift_cl(p0, [meta])
    levels = model.neck_3d.forward_cl(vol)
    fused = model.bbox_head.forward_cl(levels)
    (cb, csc, cidx), = model.bbox_head.get_candidates_cl(fused, valid, [meta], want_index=True)
    boxes, scores, labels = model.bbox_head._nms(cb, csc, meta)
    return dict(p0=p0, vol=vol, valid=valid, levels=levels, cand_boxes=cb, cand_scores=csc, cand_index=cidx, boxes=boxes, scores=scores,
                labels=labels)


def _kept_ids(ia, boxes_tensor, labels, cand_boxes, cand_index):
    """(level, voxel, class) of every kept detection, by exact row matching against the candidate list."""
    from test_gpu_configs import _scannet_box_tensor
    if cand_boxes.shape[1] == 6:
        allb = _scannet_box_tensor(ia, cand_boxes)
    else:
        allb = ia.DepthInstance3DBoxes(cand_boxes, origin=(.5, .5, .5)).tensor
    src = cand_index.cpu()[match_rows(boxes_tensor, allb)]
    return torch.stack([src >> 32, src & 0xffffffff, labels.cpu()], 1).numpy()


def _iou3d_aligned(a, b):
    """Axis-aligned 3-D IoU matrix of two [n,7]-tensor box sets in the box object's layout (x, y, z_bottom, dx, dy, dz, yaw = 0)."""
    def corners(t):
        lo = torch.stack([t[:, 0] - t[:, 3] / 2, t[:, 1] - t[:, 4] / 2, t[:, 2]], 1)
        return lo, lo + t[:, 3:6]
    alo, ahi = corners(a.double())
    blo, bhi = corners(b.double())
    inter = (torch.minimum(ahi[:, None], bhi[None]) - torch.maximum(alo[:, None], blo[None])).clamp_min(0).prod(-1)
    va, vb = (ahi - alo).prod(-1), (bhi - blo).prod(-1)
    return inter / (va[:, None] + vb[None] - inter).clamp_min(1e-12)


def _rel_rms(a, b):
    a, b = torch.as_tensor(a).double().flatten(), torch.as_tensor(b).double().flatten()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


@pytest.mark.parametrize('cfg_name,V', [('scannet_fast', 20), ('sunrgbd_fast', 1), ('scannet_v1', 50)])
def test_indoor_chained_kept_indices_identical(ia, cfg_name, V):
    """fp32, nothing substituted: image -> trunk -> lift -> neck -> head -> top-k -> NMS on the device vs the oracle's chain.
    The kept (level, voxel, class) triples must be identical and in identical order (score ties to 1e-5 relative may swap,
    counted and printed); scores within 1e-4 relative; and simple_test returns exactly these detections."""
    model, img, meta, ref = _indoor_case(ia, cfg_name, V)
    model.prepare(torch.device('cuda'))
    dimg = img.cuda()
    ch = _chained_indoor(model, dimg, meta)
    assert np.array_equal(ch['valid'][0].cpu().numpy(), ref['valid'][0])
    print(cfg_name, V, 'views: volume rel rms', _rel_rms(ch['vol'][0].permute(3, 0, 1, 2).cpu(), ref['vol']),
          'levels', [_rel_rms(l.permute(0, 4, 1, 2, 3).cpu(), r) for l, r in zip(ch['levels'], ref['levels'])])
    n = len(ch['scores'])
    print(cfg_name, 'chained detections', n, 'oracle', len(ref['scores']))
    assert n > 10
    got = _kept_ids(ia, ch['boxes'].tensor, ch['labels'], ch['cand_boxes'], ch['cand_index'])
    want = _kept_ids(ia, ref['boxes'], ref['labels'], ref['cand_boxes'], ref['cand_index'])
    assert_same_kept(f'{cfg_name} chained', got, ch['scores'].cpu().numpy(), want, ref['scores'].numpy())
    assert torch.allclose(ch['scores'].cpu().sort(descending=True)[0], ref['scores'].sort(descending=True)[0], rtol=1e-4, atol=1e-6)
    out = model.simple_test(dimg, [meta])[0]          # the drop-in call runs the same chain (native handle + op-level tail)
    assert torch.equal(out['scores_3d'], ch['scores'].cpu()) and torch.equal(out['labels_3d'], ch['labels'].cpu())
    assert torch.equal(out['boxes_3d'].tensor, ch['boxes'].tensor.cpu())


def test_kitti_batch4_chained_kept_anchor_indices(ia):
    """BASELINE config 2, chained: the anchor indices of the kept boxes of every sample from this library's own volume == the
    oracle's, in order, and simple_test returns those boxes."""
    from oracle import imvoxel_oracle as orc
    model = ia.build_detector(kc.kitti_model_cfg(), test_cfg=kc.KITTI_TEST_CFG)
    ia.randomize_(model, 123)
    with torch.no_grad():
        model.bbox_head.conv_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(5))
        model.bbox_head.conv_cls.bias.fill_(-2.0)
        model.bbox_head.conv_reg.weight.normal_(0, 0.002, generator=torch.Generator().manual_seed(6))
        model.bbox_head.conv_dir_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(7))
    B = 4
    metas = [kc.kitti_meta(t=(0.02 * b, 0.01 * b, 0.0), box_type=ia.LiDARInstance3DBoxes) for b in range(B)]
    img = torch.randn(B, 1, 3, 384, 1280, generator=torch.Generator().manual_seed(21))
    cfg = dict(n_voxels=(216, 248, 12), voxel_size=(.32, .32, .32), neck='kitti', num_classes=1, test_cfg=kc.KITTI_TEST_CFG,
               anchor=dict(ranges=[[0, -39.68, -1.78, 69.12 - .32, 39.68 - .32, -1.78]], sizes=[[1.6, 3.9, 1.56]], rotations=[0, 1.57]))
    ref, mid = orc.simple_test_anchor(img, metas, _cpu_sd(model), cfg)
    model.prepare(torch.device('cuda'))
    dimg = img.cuda()
    vol, valid = model.lift_cl(model.features_2d_cl(dimg), metas)
    boxes, scores, labels, count, (ci, cb, cs) = model.detect_cl(vol, metas, want_candidates=True)
    anchors = orc.grid_anchors(mid['cls'].shape[-2:], cfg['anchor']['ranges'], cfg['anchor']['sizes'], cfg['anchor']['rotations'])
    out = model.simple_test(dimg, metas)
    for b in range(B):
        ob, osc, _, topk = orc.anchor_head_candidates(mid['cls'][b], mid['reg'][b], mid['dir'][b], anchors, 1, 100)
        rb, rs, rl = ref[b]
        n = int(count[b])
        got = ci[b].cpu()[match_rows(torch.cat([boxes[b, :n, :6], scores[b, :n, None]], 1), torch.cat([cb[b, :, :6], cs[b, :, None]], 1))]
        want = topk[match_rows(torch.cat([rb[:, :6], rs[:, None]], 1), torch.cat([ob[:, :6], osc[:, :1]], 1))]
        assert_same_kept(f'kitti chained sample {b}', got.numpy(), scores[b, :n].cpu().numpy(), want.numpy(), rs.numpy())
        assert torch.equal(out[b]['scores_3d'], scores[b, :n].cpu()) and torch.equal(out[b]['boxes_3d'].tensor, boxes[b, :n].cpu())


def _detection_metrics(ia, ch, ref):
    """Detection-level agreement of a chained run `ch` with the oracle `ref`:
      cand_recall          of the oracle's per-level top-k candidates (level, voxel), the fraction that is a candidate here too
      recall / recall_top_quartile   of the (level, voxel, class) ids the oracle KEEPS after NMS, the fraction kept here (all / best quartile)
      box_recall / box_precision     an oracle detection is FOUND when a detection of the same class overlaps it with 3-D IoU >= 0.5
      max_dscore_found     score difference to that match."""
    m = {}
    got = _kept_ids(ia, ch['boxes'].tensor, ch['labels'], ch['cand_boxes'], ch['cand_index'])
    want = _kept_ids(ia, ref['boxes'], ref['labels'], ref['cand_boxes'], ref['cand_index'])
    gs, rs = ch['scores'].cpu().numpy(), ref['scores'].numpy()
    gpos = {tuple(r): i for i, r in enumerate(got.tolist())}
    shared = [(i, gpos[tuple(r)]) for i, r in enumerate(want.tolist()) if tuple(r) in gpos]
    m['n_here'], m['n_oracle'], m['n_shared'] = len(got), len(want), len(shared)
    m['recall'] = len(shared) / max(len(want), 1)
    q = max(1, len(want) // 4)                       # the oracle's kept list is in descending score order
    m['recall_top_quartile'] = sum(1 for i, _ in shared if i < q) / q
    m['max_dscore_shared'] = float(max((abs(gs[j] - rs[i]) for i, j in shared), default=0.0))
    m['score_range_oracle'] = (float(rs.min()), float(rs.max()))
    cand_here, cand_ref = set(ch['cand_index'].cpu().tolist()), set(ref['cand_index'].tolist())
    m['cand_recall'] = len(cand_here & cand_ref) / max(len(cand_ref), 1)
    iou = _iou3d_aligned(ref['boxes'], ch['boxes'].tensor.cpu())
    same = ref['labels'][:, None] == ch['labels'].cpu()[None]
    ok = (iou >= 0.5) & same
    m['box_recall'] = float(ok.any(1).double().mean())
    m['box_precision'] = float(ok.any(0).double().mean())
    best = torch.where(ok, iou, torch.full_like(iou, -1.0)).argmax(1)
    found = ok.any(1)
    m['mean_iou_found'] = float(iou[torch.arange(len(best)), best][found].mean()) if bool(found.any()) else 0.0
    m['max_dscore_found'] = float((ch['scores'].cpu()[best] - ref['scores']).abs()[found].max()) if bool(found.any()) else 0.0
    return m


def _control_run(ia, case, level_noise, seed=0):
    """The fp32 chain with the three neck levels (the head's inputs) perturbed by white noise of `level_noise[l]` x the level's rms --
    the error a precision mode was measured to leave there: what a perturbation of that size does to THIS network's detections,
    whatever its source.  The synthetic head scores a dense field of near-equal, heavily overlapping candidates (all boxes ~2 m in a
    6.4 m room) and greedy NMS amplifies any reordering, so the detection-level criterion of a mode is stated relative to this
    control, not as an absolute."""
    model, img, meta, ref = case
    model.prepare(torch.device('cuda'))
    p0 = model.features_2d_cl(img.cuda())
    vol, valid = model.lift_cl(p0, [meta])
    levels = model.neck_3d.forward_cl(vol)
    g = torch.Generator(device='cuda').manual_seed(seed)
    levels = [lv + torch.randn(lv.shape, device='cuda', generator=g) * (eps * float(lv.pow(2).mean().sqrt())) for lv, eps in zip(levels, level_noise)]
    fused = model.bbox_head.forward_cl(levels)
    (cb, csc, cidx), = model.bbox_head.get_candidates_cl(fused, valid, [meta], want_index=True)
    boxes, scores, labels = model.bbox_head._nms(cb, csc, meta)
    ch = dict(cand_boxes=cb, cand_scores=csc, cand_index=cidx, boxes=boxes, scores=scores, labels=labels)
    return _detection_metrics(ia, ch, ref)


# ------------------------------------------------------------------------------------------------ config 5 as named
# Measured on MI355X (the test prints the values; bounds = measured + ~30 % head-room; profiles/r03_config5_named_mode.md):
#   stage errors = rms(x - x_oracle) / rms(x_oracle)            measured bf16 / +fp8conv / +fp8storage
#     fpn0 0.0075 / 0.036 / 0.094, volume 0.0041 / 0.021 / 0.040, neck levels 0.0086, 0.0072, 0.0068 / 0.021, 0.020, 0.015 / 0.039, 0.028, 0.026
#   detections vs the oracle's 113: kept-id recall 0.79 / 0.73 / 0.43, box recall 0.89 / 0.90 / 0.62 (control with white noise of the same
#   rms on the neck levels: 0.60 / 0.34 / 0.21 and 0.78 / 0.65 / 0.54 -- rounding error is far more benign than white noise)
CONFIG5_BOUNDS = {
    #                    fpn0    volume  level0  level1  level2  cand_recall  id recall  box recall
    'bf16':             (0.010,  0.006,  0.012,  0.010,  0.010,  0.90,        0.70,      0.80),
    # round 6: the named mode = calibrate_fp8(variant='conv3') (e4m3 on conv3 of stages 3 - 4): ABSOLUTE feature bars of the round-5 verdict (item 6):
    # FPN and volume <= 2.5 %, neck levels <= 3 % rms of the fp32 oracle (measured: profiles/r06_config5.md)
    'bf16+fp8conv':     (0.025,  0.025,  0.030,  0.030,  0.030,  0.88,        0.62,      0.80),
    'bf16+fp8conv_full': (0.048, 0.028,  0.028,  0.026,  0.020,  0.88,        0.62,      0.80),       # the round-3 mode (every interior tensor e4m3)
    'bf16+fp8storage':  (0.125,  0.053,  0.052,  0.038,  0.035,  0.80,        0.35,      0.50),
}


@pytest.fixture(scope='module')
def config5_case(ia):
    return _indoor_case(ia, 'scannet_v1', 50)


def measure_config5(ia, case, mode):
    model, img, meta, ref = case
    dev = torch.device('cuda')
    model.prepare(dev, dtype=torch.bfloat16)
    dimg = img.cuda()
    if mode != 'bf16':
        model.calibrate_fp8(dimg, residual='fp8' if mode == 'bf16+fp8storage' else 'bf16', variant='full' if mode == 'bf16+fp8conv_full' else 'conv3')
    ch = _chained_indoor(model, dimg, meta)
    m = dict(mode=mode, valid_equal=bool(np.array_equal(ch['valid'][0].cpu().numpy(), ref['valid'][0])))
    m['fpn0'] = _rel_rms(ch['p0'].float().permute(0, 4, 1, 2, 3)[:, :, 0].cpu(), ref['f0'])
    m['volume'] = _rel_rms(ch['vol'][0].float().permute(3, 0, 1, 2).cpu(), ref['vol'])
    for l in range(3):
        m[f'level{l}'] = _rel_rms(ch['levels'][l].float().permute(0, 4, 1, 2, 3).cpu(), ref['levels'][l])
    m.update(_detection_metrics(ia, ch, ref))
    out = model.simple_test(dimg, [meta])[0]
    m['simple_test_equal'] = bool(torch.equal(out['scores_3d'], ch['scores'].cpu()))
    model.prepare(dev)                                # back to fp32 for whoever shares the fixture
    return m


@pytest.mark.parametrize('mode', ['bf16', 'bf16+fp8conv', 'bf16+fp8conv_full', 'bf16+fp8storage'])
def test_config5_named_precision_mode_vs_fp32_oracle(ia, config5_case, mode):
    """BASELINE config 5 AS NAMED -- 50 views 3x480x640, 80x80x32 voxels, Atlas neck, V1 head, bf16 storage with the 2-D
    convolutions on fp8 MFMA -- against the FP32 ORACLE (not against this library's own fp32 or bf16 path).
      'bf16'             bf16 activations / weights everywhere, fp32 accumulate (the mode's base)
      'bf16+fp8conv'     + ImVoxelNet.calibrate_fp8(residual='bf16') (variant 'conv3'): conv3 of ResNet stages 3 - 4 on v_mfma_f32_32x32x16_fp8_fp8
                         (e4m3 input written by conv2, e4m3 filters), everything else bf16 -- the mode the config names at a usable accuracy:
                         FPN / volume <= 2.5 %, neck levels <= 3 % of the fp32 oracle, asserted ABSOLUTELY
      'bf16+fp8conv_full' the round-3 form of it (variant 'full': conv1 / conv2 outputs and conv2 / conv3 filters e4m3 in every stage; 3.6 %)
      'bf16+fp8storage'  + calibrate_fp8(residual='fp8'): every trunk activation e4m3 -- bandwidth stress mode
    Asserted: valid mask identical (the projection stays fp32); rms error of the FPN map, the volume and the three neck levels
    relative to the oracle tensor's rms, at the measured values; the fraction of the oracle's top-k candidates that are candidates
    here; and the detection criterion -- kept-id recall, and box-level recall / precision (same class, 3-D IoU >= 0.5) against the
    oracle's detections must be within 0.10-0.12 of a CONTROL: the fp32 chain with white noise of the mode's measured rms added to
    the three neck levels (the head's inputs).  (An absolute bar is meaningless on a random-weight head: its candidates are a dense field of ~2 m boxes in a
    6.4 m room with near-equal scores, and greedy NMS turns ANY 1 % perturbation into ~10 % different picks -- measured: bf16, at
    0.75 % feature error, keeps 79 % of the oracle's ids / finds 89 % of its boxes; the control shows the same.)"""
    m = measure_config5(ia, config5_case, mode)
    ctl = _control_run(ia, config5_case, [m['level0'], m['level1'], m['level2']])
    print('config5', m)
    print('config5 control (fp32 + white noise of the measured rms on the three neck levels)', ctl)
    b = CONFIG5_BOUNDS[mode]
    assert m['valid_equal'] and m['simple_test_equal']
    for k, bound in zip(('fpn0', 'volume', 'level0', 'level1', 'level2'), b[:5]):
        assert m[k] <= bound, (mode, k, m[k], bound)
    assert m['cand_recall'] >= b[5], (mode, m['cand_recall'])
    # detections: no worse than what ANY perturbation of this size does to this network (control), within 0.1; and sane in absolute terms
    assert m['box_recall'] >= ctl['box_recall'] - 0.10 and m['box_precision'] >= ctl['box_precision'] - 0.10, (mode, m['box_recall'], ctl['box_recall'])
    assert m['recall'] >= ctl['recall'] - 0.12, (mode, m['recall'], ctl['recall'])
    assert m['recall'] >= b[6] and m['box_recall'] >= b[7], (mode, m['recall'], m['box_recall'])      # and at the measured level
    assert abs(m['n_here'] - m['n_oracle']) <= max(5, m['n_oracle'] // 4)
    assert m['max_dscore_found'] <= 0.1


if __name__ == '__main__':           # tools-style use on the GPU box: print the measurements the bounds above were set from
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import imvoxelnet_amd as _ia
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    for gain, bias in [(float(a.split(',')[0]), float(a.split(',')[1])) for a in (sys.argv[2:] or ['1,-2'])]:
        case = _indoor_case(_ia, 'scannet_v1', V, cls_gain=gain, cls_bias=bias)
        for mode in ('bf16', 'bf16+fp8conv', 'bf16+fp8conv_full', 'bf16+fp8storage'):
            m = measure_config5(_ia, case, mode)
            m['head'] = (gain, bias)
            print(json.dumps(m))
            print(json.dumps(dict(_control_run(_ia, case, [m['level0'], m['level1'], m['level2']]), mode='control for ' + mode)))
