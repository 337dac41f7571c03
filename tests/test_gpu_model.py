"""-m gpu: whole-path parity of the MI355X ImVoxelNet against the CPU oracle (torch fp32 + C)."""
import os

import numpy as np
import pytest
import torch

from gpu_util import assert_close, uncl, match_rows, assert_same_kept
from kitti_cfg import kitti_model_cfg, KITTI_TEST_CFG, kitti_meta

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


def test_resnet50_fpn_vs_oracle(ia):
    """ResNet-50 + FPN level 0 on a 128x192 image, batch 2, against the torch-CPU restatement
    (parity unpinned upstream: mmdet/torchvision sources are absent)."""
    from oracle import imvoxel_oracle as orc
    torch.manual_seed(0)
    bb = ia.ResNet(depth=50)
    fpn = ia.FPN([256, 512, 1024, 2048], 64, 4)
    ia.randomize_(bb, 1)
    ia.randomize_(fpn, 2)
    img = torch.randn(2, 3, 128, 192, generator=torch.Generator().manual_seed(3))
    sd = {'backbone.' + k: v for k, v in _cpu_sd(bb).items()}
    sd.update({'neck.' + k: v for k, v in _cpu_sd(fpn).items()})
    with torch.no_grad():
        feats = orc.resnet50(img, sd)
        ref0 = orc.fpn_level0(feats, sd)
    outs = bb(img.cuda())
    for i, (o, r) in enumerate(zip(outs, feats)):
        assert_close(f'C{i + 2}', o, r, 2e-3, 2e-3 * float(r.abs().max()))
    p = fpn(outs, all_levels=False)[0]
    assert_close('fpn0', p, ref0, 2e-3, 2e-3 * float(ref0.abs().max()))


def test_resnet50_dcnv2_vs_oracle(ia):
    """nuScenes reference backbone: ResNet-50 with DCNv2 in stages 3-4 (deformable im2col kernel + MFMA GEMM) against
    the torch restatement of mmcv's ModulatedDeformConv2dPack (parity unpinned upstream)."""
    from oracle import imvoxel_oracle as orc
    bb = ia.ResNet(depth=50, dcn=dict(type='DCNv2', deform_groups=1, fallback_on_stride=False), stage_with_dcn=(False, False, True, True))
    ia.randomize_(bb, 4)
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        for name, m in bb.named_modules():
            if name.endswith('conv_offset'):          # sub-pixel offsets of a few pixels, mixed mask values
                m.weight.normal_(0, 0.02, generator=g)
                m.bias.normal_(0, 0.5, generator=g)
    img = torch.randn(2, 3, 96, 128, generator=torch.Generator().manual_seed(6))
    sd = {'backbone.' + k: v for k, v in _cpu_sd(bb).items()}
    with torch.no_grad():
        feats = orc.resnet50(img, sd)
    outs = bb(img.cuda())
    for i, (o, r) in enumerate(zip(outs, feats)):
        assert_close(f'C{i + 2} (dcn)', o, r, 2e-3, 2e-3 * float(r.abs().max()))


def test_kitti_full_path_vs_oracle(ia):
    """BASELINE config 2 at full size (1 x 3x384x1280, 216x248x12 voxels), batch 1: feature volume within 1e-3,
    valid mask exact, identical kept anchors after NMS (north_star parity clause)."""
    from oracle import imvoxel_oracle as orc
    model = ia.build_detector(kitti_model_cfg(), test_cfg=KITTI_TEST_CFG)
    ia.randomize_(model, 123)
    with torch.no_grad():      # make some anchors fire: cls bias as trained nets have, wider weights
        model.bbox_head.conv_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(5))
        model.bbox_head.conv_cls.bias.fill_(-2.0)
        model.bbox_head.conv_reg.weight.normal_(0, 0.002, generator=torch.Generator().manual_seed(6))
        model.bbox_head.conv_dir_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(7))
    meta = kitti_meta(box_type=ia.LiDARInstance3DBoxes)
    img = torch.randn(1, 1, 3, 384, 1280, generator=torch.Generator().manual_seed(11))
    sd = _cpu_sd(model)
    cfg = dict(n_voxels=(216, 248, 12), voxel_size=(.32, .32, .32), neck='kitti', num_classes=1, test_cfg=KITTI_TEST_CFG,
               anchor=dict(ranges=[[0, -39.68, -1.78, 69.12 - .32, 39.68 - .32, -1.78]], sizes=[[1.6, 3.9, 1.56]], rotations=[0, 1.57]))
    ref, mid = orc.simple_test_anchor(img, [meta], sd, cfg)

    dimg = img.cuda()
    p0 = model.features_2d_cl(dimg)
    assert_close('fpn0', uncl(p0)[:, :, 0], mid['fpn0'][0], 0, 2e-4 * float(mid['fpn0'].abs().max()))
    vol, valid = model.lift_cl(p0, [meta])
    # valid mask depends only on geometry -> exact
    assert np.array_equal(valid.cpu().numpy(), mid['valids'][:, 0].numpy())
    assert_close('volume', vol.permute(0, 4, 1, 2, 3), mid['volume'], 0, 2e-4 * float(mid['volume'].abs().max()))
    # neck + head from the ORACLE's volume so the 3-D stack is compared on identical inputs
    vol_ref = mid['volume'].permute(0, 2, 3, 4, 1).contiguous().cuda()
    y = model.neck_3d.forward_cl(vol_ref)
    assert_close('neck', y[:, :, :, 0].permute(0, 3, 2, 1), mid['neck'], 0, 2e-4 * float(mid['neck'].abs().max()))
    boxes, scores, labels, count, cands = model.detect_cl(vol_ref, [meta], want_candidates=True)
    rb, rs, rl = ref[0]
    n = int(count[0])
    print('detections', n, 'reference', len(rs))
    # candidate anchors (top-k indices) identical
    ocls, oreg, odir = mid['cls'][0], mid['reg'][0], mid['dir'][0]
    anchors = orc.grid_anchors(ocls.shape[-2:], cfg['anchor']['ranges'], cfg['anchor']['sizes'], cfg['anchor']['rotations'])
    _, _, _, topk = orc.anchor_head_candidates(ocls, oreg, odir, anchors, 1, 100)
    got_idx = cands[0][0].cpu()
    same = (got_idx == topk).float().mean().item()
    print('top-k index agreement', same)
    assert same == 1.0, f'top-k anchors differ: {got_idx.tolist()[:10]} vs {topk.tolist()[:10]}'
    assert n == len(rs)
    assert_close('scores', scores[0, :n], rs, 1e-3, 1e-4)
    assert_close('boxes', boxes[0, :n], rb, 1e-3, 1e-3)
    # whole pipeline end to end through the public API
    out = model.simple_test(dimg, [meta])
    assert len(out) == 1 and len(out[0]['scores_3d']) == len(rs)
    assert_close('e2e boxes', out[0]['boxes_3d'].tensor, rb, 2e-3, 2e-3)


def test_nuscenes_full_path_vs_oracle(ia):
    """BASELINE config 3 at full size (6 x 3x928x1600 through ResNet-50 + DCNv2 + FPN, 312x312x12 voxels, NuScenes neck,
    dir_offset pi/4, nms_pre 1000 / max_num 500): six-view feature volume, valid mask (exact), neck, kept boxes."""
    from oracle import imvoxel_oracle as orc
    from kitti_cfg import nuscenes_model_cfg, nuscenes_meta, NUSCENES_TEST_CFG
    model = ia.build_detector(nuscenes_model_cfg(), test_cfg=NUSCENES_TEST_CFG)
    ia.randomize_(model, 321)
    with torch.no_grad():
        g = torch.Generator().manual_seed(9)
        for name, m in model.backbone.named_modules():
            if name.endswith('conv_offset'):
                m.weight.normal_(0, 0.02, generator=g)
                m.bias.normal_(0, 0.5, generator=g)
        model.bbox_head.conv_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(5))
        model.bbox_head.conv_cls.bias.fill_(-2.0)
        model.bbox_head.conv_reg.weight.normal_(0, 0.002, generator=torch.Generator().manual_seed(6))
        model.bbox_head.conv_dir_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(7))
    meta = nuscenes_meta(box_type=ia.LiDARInstance3DBoxes)
    img = torch.randn(1, 6, 3, 928, 1600, generator=torch.Generator().manual_seed(12))
    sd = _cpu_sd(model)
    cfg = dict(n_voxels=(312, 312, 12), voxel_size=(.32, .32, .32), neck='nuscenes', num_classes=1, test_cfg=NUSCENES_TEST_CFG,
               dir_offset=0.7854, dir_limit_offset=0,
               anchor=dict(ranges=[[-49.92, -49.92, -1.0, 49.92 - .64, 49.92 - .64, -1.0]], sizes=[[1.98, 4.67, 1.74]], rotations=[0, 1.57]))
    ref, mid = orc.simple_test_anchor(img, [meta], sd, cfg)

    dimg = img.cuda()
    p0 = model.features_2d_cl(dimg)
    assert_close('fpn0', uncl(p0)[:, :, 0], mid['fpn0'][0], 0, 2e-4 * float(mid['fpn0'].abs().max()))
    vol, valid = model.lift_cl(p0, [meta])
    assert np.array_equal(valid.cpu().numpy(), mid['valids'][:, 0].numpy())
    assert_close('volume', vol.permute(0, 4, 1, 2, 3), mid['volume'], 0, 2e-4 * float(mid['volume'].abs().max()))
    vol_ref = mid['volume'].permute(0, 2, 3, 4, 1).contiguous().cuda()
    y = model.neck_3d.forward_cl(vol_ref)
    assert_close('neck', y[:, :, :, 0].permute(0, 3, 2, 1), mid['neck'], 0, 2e-4 * float(mid['neck'].abs().max()))
    boxes, scores, labels, count, (ci, cb, cs) = model.detect_cl(vol_ref, [meta], want_candidates=True)
    rb, rs, rl = ref[0]
    n = int(count[0])
    print('detections', n, 'reference', len(rs))
    assert n > 100
    # north_star: identical box indices after NMS.  1000 candidates of 48 672 anchors -> 500 kept at IoU 0.2; kept
    # detections are traced back to their anchor index on both sides by exact row matching against the candidate lists
    anchors = orc.grid_anchors(mid['cls'].shape[-2:], cfg['anchor']['ranges'], cfg['anchor']['sizes'], cfg['anchor']['rotations'])
    ob, osc, _, topk = orc.anchor_head_candidates(mid['cls'][0], mid['reg'][0], mid['dir'][0], anchors, 1, 1000)
    assert_same_kept('nuscenes top-k anchors', ci[0].cpu().numpy(), cs[0].cpu().numpy(), topk.numpy(), osc[:, 0].numpy(), boundary=True)
    got = ci[0].cpu()[match_rows(torch.cat([boxes[0, :n, :6], scores[0, :n, None]], 1), torch.cat([cb[0, :, :6], cs[0, :, None]], 1))]
    want = topk[match_rows(torch.cat([rb[:, :6], rs[:, None]], 1), torch.cat([ob[:, :6], osc[:, :1]], 1))]
    assert_same_kept('nuscenes kept anchors', got.numpy(), scores[0, :n].cpu().numpy(), want.numpy(), rs.numpy())
    assert_close('scores', scores[0, :n].cpu().sort(descending=True)[0], rs.sort(descending=True)[0], 1e-4, 1e-6)


@pytest.mark.parametrize('cfg_name', ['scannet_fast', 'sunrgbd_fast'])
def test_indoor_full_path_vs_oracle(ia, cfg_name):
    """BASELINE configs 4 / 5 shapes at full size (ScanNet fast: 50 views 3x480x640; SUN RGB-D fast: 1 view; 40x40x16
    voxels, FastIndoorImVoxelNeck, V2 heads with 18 / 10 classes, nms_pre 1000): volume, valid mask, neck levels and
    the detections after aligned / rotated multi-class NMS against the oracle."""
    from oracle import imvoxel_oracle as orc
    import kitti_cfg as kc
    if cfg_name == 'scannet_fast':
        mcfg, tcfg, V, n_reg = kc.scannet_fast_model_cfg(), kc.SCANNET_FAST_TEST_CFG, 50, 6
        meta = kc.indoor_meta(V, box_type=ia.DepthInstance3DBoxes)
    else:
        mcfg, tcfg, V, n_reg = kc.sunrgbd_fast_model_cfg(), kc.SUNRGBD_FAST_TEST_CFG, 1, 7
        meta = kc.indoor_meta(1, origin=(0, 3, -1), box_type=ia.DepthInstance3DBoxes)
    model = ia.build_detector(mcfg, test_cfg=tcfg)
    ia.randomize_(model, 77)
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        model.bbox_head.cls_conv.weight.normal_(0, 0.001, generator=g)      # neck outputs reach ~300: keep the logits O(1)
        model.bbox_head.cls_conv.bias.fill_(-2.0)
        model.bbox_head.centerness_conv.weight.normal_(0, 0.0005, generator=g)
        model.bbox_head.reg_conv.weight.normal_(0, 0.0002, generator=g)
    img = torch.randn(1, V, 3, 480, 640, generator=torch.Generator().manual_seed(13))
    sd = _cpu_sd(model)
    nv, vs = mcfg['n_voxels'], mcfg['voxel_size']
    with torch.no_grad():
        f0 = orc.fpn_level0(orc.resnet50(img[0], sd), sd)
        vol_ref, ok_ref = orc.extract_volume(f0.numpy(), meta, nv, vs)
        sdn = {k[len('neck_3d.'):]: v for k, v in sd.items() if k.startswith('neck_3d.')}
        sdh = {k[len('bbox_head.'):]: v for k, v in sd.items() if k.startswith('bbox_head.')}
        lv = orc.fast_indoor_neck(torch.from_numpy(vol_ref)[None], sdn)
        cs, bs, ss = orc.fcos_head_forward(lv, sdh, n_reg)
        rb, rs, rl, (ocb, ocs, oci) = orc.fcos_get_bboxes_single([c[0] for c in cs], [b[0] for b in bs], [s[0] for s in ss],
                                                                 torch.from_numpy(ok_ref).float(), meta['lidar2img']['origin'], vs, n_reg,
                                                                 tcfg, return_candidates=True)
    model.prepare(torch.device('cuda'))
    p0 = model.features_2d_cl(img.cuda())
    assert_close('fpn0', uncl(p0)[:, :, 0], f0, 0, 2e-4 * float(f0.abs().max()))
    vol, valid = model.lift_cl(p0, [meta])
    assert np.array_equal(valid[0].cpu().numpy(), ok_ref[0])
    vmax = float(np.abs(vol_ref).max())
    assert_close('volume', vol[0].permute(3, 0, 1, 2), vol_ref, 0, 2e-4 * vmax)
    # neck, head and NMS from the ORACLE's volume, so the 3-D stack and the tail are compared on identical inputs
    vol_in = torch.from_numpy(vol_ref).permute(1, 2, 3, 0)[None].contiguous().cuda()
    levels = model.neck_3d.forward_cl(vol_in)
    for l in range(3):
        assert_close(f'neck level {l}', uncl(levels[l]), lv[l], 0, 2e-4 * float(lv[l].abs().max()))
    fused = model.bbox_head.forward_cl(levels)
    (cb, csc, cidx), = model.bbox_head.get_candidates_cl(fused, valid, [meta], want_index=True)
    boxes, scores, labels = model.bbox_head._nms(cb, csc, meta)
    n = len(scores)
    print(cfg_name, 'detections', n, 'oracle', len(rs))
    assert n > 10
    # north_star: identical box indices after NMS -- (level, voxel, class) of every kept detection on both sides
    from test_gpu_configs import _assert_indoor_kept_identical
    _assert_indoor_kept_identical(ia, cfg_name, boxes, scores, labels, cb, cidx, rb, rs, rl, ocb, oci)
    assert_close('scores', scores.cpu().sort(descending=True)[0], rs.sort(descending=True)[0], 1e-4, 1e-6)
    (b2, s2, l2), = model.detect_indoor_cl(vol_in, valid, [meta])      # the public tail returns the same thing
    assert torch.equal(s2, scores) and torch.equal(l2, labels) and torch.equal(b2.tensor, boxes.tensor)


def test_kitti_bf16_storage_mode_tracks_fp32(ia):
    """Optional reduced-precision mode (BASELINE config 5; the reference itself is fp32-only): bf16 activations and
    weights, fp32 accumulate / epilogue / head output / tail.  Checked against THIS library's fp32 path (which is the
    one pinned to the oracle) at the full KITTI size.  Tolerances (relative to the tensor's max magnitude): FPN level 0
    3e-2, neck output 5e-2, head logits 5e-2; the 10 best detections agree within 0.3 m / 0.05 score."""
    model = ia.build_detector(kitti_model_cfg(), test_cfg=KITTI_TEST_CFG)
    ia.randomize_(model, 123)
    with torch.no_grad():
        model.bbox_head.conv_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(5))
        model.bbox_head.conv_cls.bias.fill_(-2.0)
        model.bbox_head.conv_reg.weight.normal_(0, 0.002, generator=torch.Generator().manual_seed(6))
        model.bbox_head.conv_dir_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(7))
    meta = kitti_meta(box_type=ia.LiDARInstance3DBoxes)
    img = torch.randn(2, 1, 3, 384, 1280, generator=torch.Generator().manual_seed(11)).cuda()
    outs = {}
    for name, dt in (('f32', torch.float32), ('bf16', torch.bfloat16)):
        model.prepare(torch.device('cuda'), dtype=dt)
        p0 = model.features_2d_cl(img)
        vol, valid = model.lift_cl(p0, [meta, meta])
        assert p0.dtype == dt and vol.dtype == dt
        y = model.neck_3d.forward_cl(vol)
        h = model.bbox_head.forward_cl(y)
        assert h.dtype == torch.float32
        det = model.simple_test(img, [meta, meta])
        outs[name] = (p0.float(), valid, y.float(), h, det)
    model.prepare(torch.device('cuda'))
    a, b = outs['f32'], outs['bf16']
    assert torch.equal(a[1], b[1]), 'the valid mask is geometry only'
    for nm, i, tol in (('fpn0', 0, 3e-2), ('neck', 2, 5e-2), ('head', 3, 5e-2)):
        err = (a[i] - b[i]).abs().max().item() / a[i].abs().max().item()
        print(f'bf16 vs f32 {nm}: max err / max |x| = {err:.4f}')
        assert err < tol, (nm, err)
    for da, db in zip(a[4], b[4]):
        na, nb = len(da['scores_3d']), len(db['scores_3d'])
        print('detections f32', na, 'bf16', nb)
        assert na > 0 and abs(na - nb) <= max(3, na // 5)
        k = min(10, na, nb)
        ca, cb = da['boxes_3d'].tensor[:k, :3], db['boxes_3d'].tensor[:k, :3]
        d = torch.cdist(ca, cb).min(dim=1).values
        print('top-k centre distance to the nearest bf16 detection', d.tolist())
        assert (d < 0.3).float().mean().item() >= 0.8
        assert abs(float(da['scores_3d'][0]) - float(db['scores_3d'][0])) < 0.05


def test_indoor_eval_on_device_matches_reference(ia):
    """indoor_eval with the 3-D IoU from the device kernel (BaseInstance3DBoxes.overlaps -> ivx_boxes_overlap_bev)."""
    from test_host_cpu import _eval_inputs
    gt, dt, want = _eval_inputs()
    got = ia.indoor_eval(gt, dt, [0.25, 0.5], {i: f'c{i}' for i in range(4)}, box_type_3d=ia.DepthInstance3DBoxes, box_mode_3d=2)
    assert set(got) == set(want)
    for k in want:
        assert abs(got[k] - want[k]) < 1e-5, (k, got[k], want[k])


def test_view_sharded_mode_matches_single_gpu(ia):
    """Second multi-GPU mode on one device: the ranks are simulated one after the other (rank / world passed
    explicitly, the all-reduce replaced by adding the partial tensors).  world = 1 is bit-identical to lift_cl (same
    view order); world = 3 over 7 views differs only by the order of fp32 additions -- of the view sum, and inside the 2-D
    trunk, whose tile / split-K plan depends on the number of views in the launch -- (<= 1e-5 of the max), with an
    identical valid mask; the replicated tail then returns the same detections."""
    from kitti_cfg import scannet_fast_model_cfg, SCANNET_FAST_TEST_CFG, indoor_meta
    from imvoxelnet_amd import dist as ivd, ops
    model = ia.build_detector(scannet_fast_model_cfg(), test_cfg=SCANNET_FAST_TEST_CFG)
    ia.randomize_(model, 9)
    with torch.no_grad():       # trained-net-like head statistics: finite box sizes, a spread of scores
        g = torch.Generator().manual_seed(5)
        model.bbox_head.cls_conv.weight.normal_(0, 0.01, generator=g)
        model.bbox_head.cls_conv.bias.fill_(-2.0)
        model.bbox_head.reg_conv.weight.normal_(0, 0.002, generator=g)
    model.prepare(torch.device('cuda'))
    V = 7
    meta = indoor_meta(V, box_type=ia.DepthInstance3DBoxes)
    img = torch.randn(1, V, 3, 480, 640, generator=torch.Generator().manual_seed(4)).cuda()
    p0 = model.features_2d_cl(img)
    vol_ref, valid_ref = model.lift_cl(p0, [meta])
    vol1, valid1 = ivd.view_sharded_lift(model, img, [meta], rank=0, world=1)
    assert torch.equal(vol1, vol_ref) and torch.equal(valid1, valid_ref)
    tot = cnt = None
    for r in range(3):
        img_l, metas_l, (v0, v1) = ivd.shard_views(img, [meta], r, 3)
        assert (v0, v1) == [(0, 3), (3, 5), (5, 7)][r]
        p0_l = model.features_2d_cl(img_l)
        proj, origin, crop = model._camera_setup(metas_l, 4, p0_l.device)
        s, c = ops.backproject_sum(p0_l, proj, origin, crop, model.voxel_size, model.n_voxels)
        tot, cnt = (s, c) if tot is None else (tot + s, cnt + c)
    vol3, valid3 = ops.volume_normalize_(tot.contiguous(), cnt.contiguous())
    assert torch.equal(valid3, valid_ref)
    err = (vol3 - vol_ref).abs().max().item() / vol_ref.abs().max().item()
    print('view-sharded (3 ranks) vs single: max err / max', err)
    assert err <= 1e-5
    a = model.simple_test(img, [meta])
    b = model.simple_test_view_sharded(img, [meta])          # no process group: world 1
    print('detections', len(a[0]['scores_3d']), len(b[0]['scores_3d']))
    assert len(a[0]['scores_3d']) > 0 and torch.isfinite(a[0]['boxes_3d'].tensor).all()
    assert torch.equal(a[0]['scores_3d'], b[0]['scores_3d'])
    assert torch.equal(a[0]['labels_3d'], b[0]['labels_3d'])
    assert torch.equal(a[0]['boxes_3d'].tensor, b[0]['boxes_3d'].tensor)


def test_detections_to_kitti_ap_chain(ia):
    """simple_test -> bbox2result_kitti -> kitti_eval on the device overlaps: with the model's own (score-thresholded)
    detections as ground truth the precision is 1 at every recall sample that exists, for the 2-D, BEV and 3-D metrics,
    and dropping the best half of the detections from the "ground truth" lowers the AP (a false-positive path) -- the whole chain is wired consistently."""
    model = ia.build_detector(kitti_model_cfg(), test_cfg=KITTI_TEST_CFG)
    ia.randomize_(model, 123)
    with torch.no_grad():
        model.bbox_head.conv_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(5))
        model.bbox_head.conv_cls.bias.fill_(-2.0)
        model.bbox_head.conv_reg.weight.normal_(0, 0.002, generator=torch.Generator().manual_seed(6))
    model.prepare(torch.device('cuda'))
    metas = [kitti_meta(t=(0.01 * b, 0.0, 0.0), box_type=ia.LiDARInstance3DBoxes) for b in range(3)]
    img = torch.randn(3, 1, 3, 384, 1280, generator=torch.Generator().manual_seed(11)).cuda()
    outs = model.simple_test(img, metas)
    assert sum(len(o['scores_3d']) for o in outs) > 10
    rect = np.eye(4, dtype=np.float32)
    trv2c = np.array([[0, -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]], np.float32)
    p2 = np.array([[721.5377, 0, 609.5593, 0], [0, 721.5377, 172.854, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    infos = [dict(image=dict(image_idx=i, image_shape=np.array([384, 1280], np.int32)),
                  calib=dict(R0_rect=rect, Tr_velo_to_cam=trv2c, P2=p2)) for i in range(3)]
    dts = ia.bbox2result_kitti(outs, infos, ['Car'])
    assert sum(len(d['score']) for d in dts) > 5
    gts = []
    for d in dts:
        n = len(d['score'])
        gts.append(dict(name=d['name'].copy(), truncated=np.zeros(n), occluded=np.zeros(n, dtype=np.int64), alpha=d['alpha'].copy(),
                        bbox=d['bbox'].copy(), dimensions=d['dimensions'].copy(), location=d['location'].copy(),
                        rotation_y=d['rotation_y'].copy()))
    keep = [d for d, g in zip(dts, gts) if len(g['name'])]
    gts = [g for g in gts if len(g['name'])]
    _, res = ia.kitti_eval(gts, keep, ['Car'])
    # perfect detections: precision 1 at every sampled recall point that exists (with N valid objects only the first
    # ~N of the 41 recall samples exist, so the 11-point AP is 100 * (#existing sample points) / 11, not 100)
    from imvoxelnet_amd import kitti_ap as ke
    mo = np.stack([np.array([[0.7, 0.5, 0.5, 0.7, 0.5]] * 3)] * 1)[:, :, [0]]
    for metric in (0, 1, 2):
        pr = ke.eval_class(gts, keep, [0], [2], metric, mo)['precision'][0, 0, 0]
        nz = pr[pr > 0]
        assert len(nz) >= 8 and np.allclose(nz, 1.0), (metric, pr)
        assert np.all(pr[len(nz):] == 0)
    assert res['KITTI/Car_3D_hard_strict'] > 60.0
    half = []
    for d, g in zip(keep, gts):
        order = np.argsort(-d['score'])
        sel = np.sort(order[len(order) // 2:]) if len(order) > 1 else order
        half.append({k: v[sel] for k, v in g.items()})
    _, res2 = ia.kitti_eval(half, keep, ['Car'])
    assert res2['KITTI/Car_3D_hard_strict'] < res['KITTI/Car_3D_hard_strict']


@pytest.mark.parametrize('cfg_name', ['scannet_v1', 'scannet_fast'])
def test_indoor_bf16_storage_mode_tracks_fp32(ia, cfg_name):
    """Optional bf16 storage on the multi-view indoor path (BASELINE config 5's dtype; the reference is fp32-only):
    bf16 multi-view lift (fp32 view sum), Atlas / FastIndoor necks incl. the trilinear and transposed-conv up-paths,
    fp32 head outputs and tail.  Against this library's fp32 path: valid mask identical, volume and neck levels within
    3 % / 6 % of the tensor's max, fused head outputs within 8 %, detection count within 25 % and best score within 0.05
    (the synthetic heads produce near-tied scores, so individual low-margin detections may swap)."""
    import kitti_cfg as kc
    cfg, tcfg = ((kc.scannet_v1_model_cfg(), kc.SCANNET_V1_TEST_CFG) if cfg_name == 'scannet_v1'
                 else (kc.scannet_fast_model_cfg(), kc.SCANNET_FAST_TEST_CFG))
    model = ia.build_detector(cfg, test_cfg=tcfg)
    ia.randomize_(model, 31)
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        model.bbox_head.cls_conv.weight.normal_(0, 0.01, generator=g)
        model.bbox_head.cls_conv.bias.fill_(-2.0)
        model.bbox_head.reg_conv.weight.normal_(0, 0.002, generator=g)
    V = 5
    meta = kc.indoor_meta(V, box_type=ia.DepthInstance3DBoxes)
    img = torch.randn(1, V, 3, 480, 640, generator=torch.Generator().manual_seed(4)).cuda()
    outs = {}
    for name, dt in (('f32', torch.float32), ('bf16', torch.bfloat16)):
        model.prepare(torch.device('cuda'), dtype=dt)
        p0 = model.features_2d_cl(img)
        vol, valid = model.lift_cl(p0, [meta])
        assert vol.dtype == dt
        levels = model.neck_3d.forward_cl(vol)
        heads = model.bbox_head.forward_cl(levels)
        det = model.simple_test(img, [meta])
        outs[name] = (vol.float(), valid, [lv.float() for lv in levels], det, [h.float() for h in heads])
    model.prepare(torch.device('cuda'))
    a, b = outs['f32'], outs['bf16']
    assert torch.equal(a[1], b[1])
    err = (a[0] - b[0]).abs().max().item() / a[0].abs().max().item()
    print(cfg_name, 'volume err / max', err)
    assert err < 3e-2
    for i, (x, y) in enumerate(zip(a[2], b[2])):
        e = (x - y).abs().max().item() / x.abs().max().item()
        print(cfg_name, 'neck level', i, 'err / max', e)
        assert e < 6e-2
    for i, (x, y) in enumerate(zip(a[4], b[4])):                 # fused head outputs (centerness, regression, class logits)
        e = (x - y).abs().max().item() / x.abs().max().item()
        print(cfg_name, 'head level', i, 'err / max', e)
        assert e < 8e-2
    na, nb = len(a[3][0]['scores_3d']), len(b[3][0]['scores_3d'])
    print(cfg_name, 'detections', na, nb)
    assert na > 0 and abs(na - nb) <= max(5, na // 4)
    assert abs(float(a[3][0]['scores_3d'].max()) - float(b[3][0]['scores_3d'].max())) < 0.05


@pytest.mark.parametrize('cfg_name,views', [('scannet_v1', 4), ('scannet_fast', 3)])
def test_fp8_trunk_tracks_bf16(ia, cfg_name, views):
    import kitti_cfg as kc
    """BASELINE config 5's "bf16 with fp8 2D-conv MFMA": ImVoxelNet.calibrate_fp8 stores the ResNet-50 activations and weights
    as e4m3 (per-tensor / per-output-channel scales from one calibration pass) on top of the bf16 mode.  Not the reference's
    precision -- the check is that the FPN level-0 map and the detections track the bf16 mode of the same weights: FPN map
    within 6 % of its max magnitude (rms within 1.5 %), the lifted volume likewise, and most detections shared."""
    mcfg = getattr(kc, f'{cfg_name}_model_cfg')()
    model = ia.build_detector(mcfg, test_cfg=dict(getattr(kc, f'{cfg_name.upper()}_TEST_CFG')))
    ia.randomize_(model, 41)
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        model.bbox_head.cls_conv.weight.normal_(0, 0.01, generator=g)
        model.bbox_head.cls_conv.bias.fill_(-2.0)
        model.bbox_head.centerness_conv.weight.normal_(0, 0.005, generator=g)
        model.bbox_head.reg_conv.weight.normal_(0, 0.002, generator=g)
    img = torch.randn(1, views, 3, 480, 640, generator=torch.Generator().manual_seed(2)).cuda()
    metas = [kc.indoor_meta(views, box_type=ia.DepthInstance3DBoxes)]
    model.prepare(torch.device('cuda'), dtype=torch.bfloat16)
    p0_bf = model.features_2d_cl(img).float()
    vol_bf, valid_bf = model.lift_cl(model.features_2d_cl(img), metas)
    calib = model.calibrate_fp8(img)
    assert model.trunk_fp8 and len(calib) >= 53
    p0 = model.features_2d_cl(img)
    assert p0.dtype == torch.bfloat16
    vol, valid = model.lift_cl(p0, metas)
    assert torch.equal(valid, valid_bf)
    for nm, a, b in (('fpn0', p0.float(), p0_bf), ('volume', vol.float(), vol_bf.float())):
        mx = float(b.abs().max())
        err, rms = float((a - b).abs().max()) / mx, float((a - b).pow(2).mean().sqrt()) / mx
        rel = float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
        print(f'{cfg_name} fp8 trunk vs bf16: {nm} max err {err:.4f} of max, rms {rms:.5f} of max, {rel:.4f} of the signal rms')
        assert err < 0.25 and rms < 0.04 and rel < 0.2, (nm, err, rms, rel)
    res = model.simple_test(img, metas)
    assert len(res) == 1 and len(res[0]['scores_3d']) > 0
    # the image generator of a second, different batch runs through the calibrated trunk as well (scales are static)
    img2 = torch.randn(1, views, 3, 480, 640, generator=torch.Generator().manual_seed(3)).cuda()
    p0_2 = model.features_2d_cl(img2).float()
    model.prepare(torch.device('cuda'), dtype=torch.bfloat16)
    assert not model.trunk_fp8
    ref2 = model.features_2d_cl(img2).float()
    mx = float(ref2.abs().max())
    rel2 = float((p0_2 - ref2).pow(2).mean().sqrt() / ref2.pow(2).mean().sqrt())
    print(f'{cfg_name} fp8 trunk, second batch with the first batch\'s scales: {rel2:.4f} of the signal rms')
    assert rel2 < 0.25


@pytest.mark.parametrize('neck_name,nv,world', [('nuscenes', (48, 40, 12), 3), ('kitti', (42, 36, 12), 4)])
def test_view_sharded_slab_neck_matches_replicated_neck(ia, neck_name, nv, world):
    """SURVEY 8e's reduce-scatter form of the view-sharded mode on one device (ranks simulated one after the other): every rank
    normalises and convolves only its x-slab of the volume, widened by the neck's receptive field (dist.StackNeckSlabs), crops its
    rows, and the concatenated rows equal the neck run on the whole volume -- to the rounding of the Winograd tiles, whose alignment
    moves with the slab's origin -- and the detections are the same."""
    from imvoxelnet_amd import dist as ivd, ops
    from imvoxelnet_amd.workloads import kitti_model_cfg, KITTI_TEST_CFG, nuscenes_model_cfg, NUSCENES_TEST_CFG
    cfg = nuscenes_model_cfg(n_voxels=nv, dcn=False) if neck_name == 'nuscenes' else kitti_model_cfg(n_voxels=nv)
    model = ia.build_detector(cfg, test_cfg=dict(NUSCENES_TEST_CFG if neck_name == 'nuscenes' else KITTI_TEST_CFG, score_thr=0.05))
    ia.randomize_(model, 17)
    with torch.no_grad():
        model.bbox_head.conv_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(5))
        model.bbox_head.conv_cls.bias.fill_(-1.5)
        model.bbox_head.conv_reg.weight.normal_(0, 0.002, generator=torch.Generator().manual_seed(6))
    model.prepare(torch.device('cuda'))
    g = torch.Generator().manual_seed(3)
    C = model.neck.out_channels
    B = 2
    # a partial view sum / view count per simulated rank (what ops.backproject_sum leaves), totals as the exchange would deliver them
    parts = [(torch.randn(B, *nv, C, generator=g).cuda(), torch.randint(0, 3, (B,) + tuple(nv), generator=g, dtype=torch.int32).cuda()) for _ in range(world)]
    tot_v, tot_c = sum(p[0] for p in parts), sum(p[1] for p in parts)
    vol, _ = ops.volume_normalize_(tot_v.clone(), tot_c.clone())
    y_ref = model.neck_3d.forward_cl(vol)
    plans = [ivd.StackNeckSlabs(model.neck_3d, nv[0], world, r) for r in range(world)]
    rows = []
    for r, pl in enumerate(plans):
        sv, sc = ivd.exchange_volume_slabs(tot_v, tot_c, plans, rank=r)          # no process group: this rank's rows of the totals
        assert sv.shape[1] == pl.eb - pl.ea
        slab, _ = ops.volume_normalize_(sv.clone(), sc.clone())
        rows.append(pl.crop(model.neck_3d.forward_cl(slab)))
        assert rows[-1].shape[1] == pl.ob - pl.oa
    y = torch.cat(rows, 1)
    assert y.shape == y_ref.shape
    rng = float(y_ref.abs().max())
    from gpu_util import assert_close
    assert_close(f'{neck_name} neck: {world} x-slabs vs the whole volume', y, y_ref, 0, 1e-4 * rng)
    metas = [dict(box_type_3d=ia.LiDARInstance3DBoxes)] * B
    outs = []
    for t in (y_ref, y):
        h = model.bbox_head.forward_cl(t)
        outs.append(model.bbox_head.get_bboxes_cl(h, t.shape[2], t.shape[1], metas, hw_transposed=True))
    (b0, s0, l0, c0), (b1, s1, l1, c1) = outs
    assert torch.equal(c0, c1) and int(c0.sum()) > 10
    assert_close('scores', s1, s0, 0, 1e-4)
    assert_close('boxes', b1, b0, 0, 1e-3)
