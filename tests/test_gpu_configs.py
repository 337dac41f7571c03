"""-m gpu: every BASELINE.json configuration at its FULL size against the CPU oracle, with the north_star parity
clause asserted as written: feature volumes within 1e-3 (asserted at the measured 2e-4 of the tensor's range), valid
masks exact, identical box indices after NMS.

  config 2  KITTI 1 x 3x384x1280, 216x248x12, batch 4                  test_kitti_batch4_full_path_vs_oracle
  config 4  nuScenes 6 views 64x232x400 -> 192x192x32 (lift stress)     test_lift_config4_nuscenes_192x192x32_bit_exact
  config 5  ScanNet 50 views 3x480x640, 80x80x32, Atlas neck, V1 head   test_scannet_v1_50view_full_path_vs_oracle
(configs 1 / 3 and the reference nuScenes grid: tests/test_gpu_model.py)
"""
import numpy as np
import pytest
import torch

from gpu_util import assert_close, uncl, cl, match_rows, assert_same_kept
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


def test_lift_config4_nuscenes_192x192x32_bit_exact(ia):
    """BASELINE config 4's unprojection: 6 cameras, FPN maps 64 x 232 x 400, voxel grid 192 x 192 x 32 (the reference's
    own nuScenes grid is 312 x 312 x 12 -- tests/test_gpu_model.py).  Two scenes with different camera rigs and a
    cropped (padded) image: HIP volume == C oracle bit for bit, valid masks equal, and an all-ones map lifts to exactly
    the valid mask."""
    from imvoxelnet_amd import ops
    from oracle import imvoxel_oracle as orc
    nv, vs = (192, 192, 32), (.32, .32, .32)
    g = torch.Generator().manual_seed(40)
    metas = [kc.nuscenes_meta(), kc.nuscenes_meta()]
    metas[1]['img_shape'] = (900, 1600, 3)                    # the real nuScenes image is 900 rows, padded to 928: crop 225 of 232
    metas[1]['ori_shape'] = (900, 1600, 3)
    for e in metas[1]['lidar2img']['extrinsic']:              # a second rig: cameras shifted by a few centimetres
        e[:3, 3] += np.float32(0.03) * e[:3, :3].sum(1)
    metas[1]['lidar2img']['origin'] = np.array([0.4, -0.2, -0.8], np.float32)
    feats = [torch.randn(6, 64, 232, 400, generator=g) for _ in range(2)]
    P = torch.from_numpy(np.stack([orc.compute_projection(m, 4) for m in metas])).cuda().contiguous()
    no = torch.stack([torch.tensor(m['lidar2img']['origin']) - torch.tensor(nv) / 2. * torch.tensor(vs) for m in metas]).cuda().contiguous()
    crop = torch.tensor([[m['img_shape'][0] // 4, m['img_shape'][1] // 4] for m in metas], dtype=torch.int32).cuda()
    vol, valid = ops.backproject_mean(cl(torch.cat(feats)), P, no, crop, vs, nv)
    torch.cuda.synchronize()
    for b in range(2):
        ref, ok = orc.extract_volume(feats[b].numpy(), metas[b], nv, vs)
        got = vol[b].permute(3, 0, 1, 2).cpu().numpy()
        assert np.array_equal(valid[b].cpu().numpy(), ok[0]), f'scene {b}: valid mask differs'
        assert np.array_equal(got, ref), f'scene {b}: {(got != ref).sum()} values differ (max {np.abs(got - ref).max()})'
        print(f'scene {b}: {int(ok.sum())} of {ok.size} voxels seen, bit-exact')
        assert 0.2 < ok.mean() < 0.99
    ones = torch.ones(12, 1, 232, 400, 64, device='cuda')
    v1, m1 = ops.backproject_mean(ones, P, no, crop, vs, nv)
    assert torch.equal(m1, valid) and torch.equal(v1, m1.unsqueeze(-1).float().expand_as(v1))


def test_kitti_batch4_full_path_vs_oracle(ia):
    """BASELINE config 2 exactly as benchmarked: batch 4 (four different cameras), 1 x 3x384x1280, 216x248x12, through the
    public simple_test and stage by stage: FPN / volume / neck within 2e-4 of their range, valid masks exact, identical
    top-k anchors, identical kept anchor indices after rotated NMS for every sample."""
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
    sd = _cpu_sd(model)
    cfg = dict(n_voxels=(216, 248, 12), voxel_size=(.32, .32, .32), neck='kitti', num_classes=1, test_cfg=kc.KITTI_TEST_CFG,
               anchor=dict(ranges=[[0, -39.68, -1.78, 69.12 - .32, 39.68 - .32, -1.78]], sizes=[[1.6, 3.9, 1.56]], rotations=[0, 1.57]))
    ref, mid = orc.simple_test_anchor(img, metas, sd, cfg)
    dimg = img.cuda()
    p0 = model.features_2d_cl(dimg)
    assert_close('fpn0', uncl(p0)[:, :, 0], mid['fpn0'][:, 0], 0, 2e-4 * float(mid['fpn0'].abs().max()))
    vol, valid = model.lift_cl(p0, metas)
    assert np.array_equal(valid.cpu().numpy(), mid['valids'][:, 0].numpy())
    assert_close('volume', vol.permute(0, 4, 1, 2, 3), mid['volume'], 0, 2e-4 * float(mid['volume'].abs().max()))
    vol_ref = mid['volume'].permute(0, 2, 3, 4, 1).contiguous().cuda()
    y = model.neck_3d.forward_cl(vol_ref)
    assert_close('neck', y[:, :, :, 0].permute(0, 3, 2, 1), mid['neck'], 0, 2e-4 * float(mid['neck'].abs().max()))
    boxes, scores, labels, count, (ci, cb, cs) = model.detect_cl(vol_ref, metas, want_candidates=True)
    anchors = orc.grid_anchors(mid['cls'].shape[-2:], cfg['anchor']['ranges'], cfg['anchor']['sizes'], cfg['anchor']['rotations'])
    total = 0
    for b in range(B):
        ob, osc, _, topk = orc.anchor_head_candidates(mid['cls'][b], mid['reg'][b], mid['dir'][b], anchors, 1, 100)
        assert torch.equal(ci[b].cpu(), topk), f'sample {b}: top-k anchors differ'
        rb, rs, rl = ref[b]
        n = int(count[b])
        got = ci[b].cpu()[match_rows(torch.cat([boxes[b, :n, :6], scores[b, :n, None]], 1), torch.cat([cb[b, :, :6], cs[b, :, None]], 1))]
        want = topk[match_rows(torch.cat([rb[:, :6], rs[:, None]], 1), torch.cat([ob[:, :6], osc[:, :1]], 1))]
        assert_same_kept(f'kitti sample {b}', got.numpy(), scores[b, :n].cpu().numpy(), want.numpy(), rs.numpy())
        assert_close(f'boxes {b}', boxes[b, :n], rb, 1e-4, 1e-4)
        assert_close(f'scores {b}', scores[b, :n], rs, 1e-4, 1e-6)
        total += n
    assert total > 20
    # and the drop-in call on the whole batch
    out = model.simple_test(dimg, metas)
    for b in range(B):
        assert len(out[b]['scores_3d']) == len(ref[b][1])
        assert_close(f'simple_test boxes {b}', out[b]['boxes_3d'].tensor, ref[b][0], 1e-3, 1e-3)


def test_scannet_v1_50view_full_path_vs_oracle(ia):
    """BASELINE config 5 at full size in the reference's precision (fp32): 50 views 3x480x640 -> ResNet-50 + FPN(64) ->
    fifty-view lift into 80x80x32 -> ImVoxelNeck (Atlas 3-D U-Net, trilinear up-path) -> ScanNetImVoxelHead (V1,
    n_convs 0) -> aligned 3-D NMS, against the torch/C restatement (necks/imvoxelnet.py:297-372,
    dense_heads/imvoxel_head.py:237-306): valid mask exact, volume / the three neck levels within 2e-4 of their range,
    identical kept (level, voxel) indices and labels."""
    from oracle import imvoxel_oracle as orc
    mcfg, tcfg, V = kc.scannet_v1_model_cfg(), kc.SCANNET_V1_TEST_CFG, 50
    meta = kc.indoor_meta(V, box_type=ia.DepthInstance3DBoxes)
    model = ia.build_detector(mcfg, test_cfg=tcfg)
    ia.randomize_(model, 78)
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        model.bbox_head.cls_conv.weight.normal_(0, 0.01, generator=g)
        model.bbox_head.cls_conv.bias.fill_(-2.0)
        model.bbox_head.centerness_conv.weight.normal_(0, 0.005, generator=g)
        model.bbox_head.reg_conv.weight.normal_(0, 0.002, generator=g)
    img = torch.randn(1, V, 3, 480, 640, generator=torch.Generator().manual_seed(14))
    sd = _cpu_sd(model)
    nv, vs = mcfg['n_voxels'], mcfg['voxel_size']
    nk = mcfg['neck_3d']
    with torch.no_grad():
        f0 = orc.fpn_level0(orc.resnet50(img[0], sd), sd)
        vol_ref, ok_ref = orc.extract_volume(f0.numpy(), meta, nv, vs)
        sdn = {k[len('neck_3d.'):]: v for k, v in sd.items() if k.startswith('neck_3d.')}
        sdh = {k[len('bbox_head.'):]: v for k, v in sd.items() if k.startswith('bbox_head.')}
        lv = orc.atlas_neck(torch.from_numpy(vol_ref)[None], sdn, nk['channels'], nk['down_layers'], nk['up_layers'])
        cs, bs, ss = orc.fcos_head_forward(lv, sdh, 6, n_convs=0)
        rb, rs, rl, (ocb, ocs, oci) = orc.fcos_get_bboxes_single([c[0] for c in cs], [b[0] for b in bs], [s[0] for s in ss],
                                                                 torch.from_numpy(ok_ref).float(), meta['lidar2img']['origin'], vs, 6, tcfg,
                                                                 return_candidates=True)
    model.prepare(torch.device('cuda'))
    p0 = model.features_2d_cl(img.cuda())
    assert_close('fpn0', uncl(p0)[:, :, 0], f0, 0, 2e-4 * float(f0.abs().max()))
    vol, valid = model.lift_cl(p0, [meta])
    assert np.array_equal(valid[0].cpu().numpy(), ok_ref[0])
    # geometry first: the product's host camera set-up (torch CPU ops, as the reference) against the oracle's (C), then the
    # 50-view lift of the ORACLE's FPN maps, which must reproduce the oracle volume bit for bit (same pixels, same order)
    proj, new_origin, crop = model._camera_setup([meta], 4, torch.device('cpu'))
    P_ref = orc.compute_projection(meta, 4)
    print('projection matrices equal:', np.array_equal(proj[0].numpy(), P_ref), ' max |d|', float(np.abs(proj[0].numpy() - P_ref).max()))
    from imvoxelnet_amd import ops
    vol_o, valid_o = ops.backproject_mean(cl(f0), torch.from_numpy(P_ref)[None].cuda().contiguous(), new_origin.cuda(), crop.cuda(), vs, nv)
    got_o = vol_o[0].permute(3, 0, 1, 2).cpu().numpy()
    bad = np.argwhere((got_o != vol_ref).any(0))
    print('lift of the oracle maps with the oracle projection:', len(bad), 'voxels differ', bad[:4].tolist())
    assert np.array_equal(got_o, vol_ref), f'{len(bad)} voxels differ from the C oracle'
    assert np.array_equal(proj[0].numpy(), P_ref), 'host camera set-up differs from the oracle (see the printed maximum)'
    assert_close('volume', vol[0].permute(3, 0, 1, 2), vol_ref, 0, 2e-4 * float(np.abs(vol_ref).max()))
    vol_in = torch.from_numpy(vol_ref).permute(1, 2, 3, 0)[None].contiguous().cuda()
    levels = model.neck_3d.forward_cl(vol_in)
    assert [tuple(l.shape[1:4]) for l in levels] == [(80, 80, 32), (40, 40, 16), (20, 20, 8)]
    for l in range(3):
        assert_close(f'neck level {l}', uncl(levels[l]), lv[l], 0, 2e-4 * float(lv[l].abs().max()))
    fused = model.bbox_head.forward_cl(levels)
    (cb, csc, cidx), = model.bbox_head.get_candidates_cl(fused, valid, [meta], want_index=True)
    boxes, scores, labels = model.bbox_head._nms(cb, csc, meta)
    n = len(scores)
    print('scannet v1 x50: detections', n, 'oracle', len(rs))
    assert n > 10
    _assert_indoor_kept_identical(ia, 'scannet_v1', boxes, scores, labels, cb, cidx, rb, rs, rl, ocb, oci)
    assert_close('scores', scores, rs, 1e-4, 1e-6)
    assert_close('boxes', boxes.tensor, rb, 1e-3, 1e-3)
    out = model.simple_test(img.cuda(), [meta])          # drop-in call from the image
    assert abs(len(out[0]['scores_3d']) - len(rs)) <= max(2, len(rs) // 50)   # from this library's own volume (within 2e-4 of the oracle's)


def _scannet_box_tensor(ia, corners):
    """corner boxes [m,6] -> the [m,7] tensor the returned box object holds (imvoxel_head_v2.py:538-544)."""
    c = torch.stack(((corners[:, 0] + corners[:, 3]) / 2., (corners[:, 1] + corners[:, 4]) / 2., (corners[:, 2] + corners[:, 5]) / 2.,
                     corners[:, 3] - corners[:, 0], corners[:, 4] - corners[:, 1], corners[:, 5] - corners[:, 2]), dim=1)
    return ia.DepthInstance3DBoxes(c, origin=(.5, .5, .5), box_dim=6, with_yaw=False).tensor


def _assert_indoor_kept_identical(ia, name, boxes, scores, labels, cand_boxes, cand_index, rb, rs, rl, ocand_boxes, ocand_index):
    """Kept detections -> (level, voxel) of the candidate they come from, on both sides, by exact row matching."""
    if cand_boxes.shape[1] == 6:
        dev_all, ref_all = _scannet_box_tensor(ia, cand_boxes), _scannet_box_tensor(ia, ocand_boxes)
    else:
        dev_all = ia.DepthInstance3DBoxes(cand_boxes, origin=(.5, .5, .5)).tensor
        ref_all = ia.DepthInstance3DBoxes(ocand_boxes, origin=(.5, .5, .5)).tensor
    got = cand_index.cpu()[match_rows(boxes.tensor, dev_all)]
    want = ocand_index[match_rows(rb, ref_all)]
    got_ids = torch.stack([got >> 32, got & 0xffffffff, labels.cpu()], 1).numpy()
    want_ids = torch.stack([want >> 32, want & 0xffffffff, rl], 1).numpy()
    return assert_same_kept(name, got_ids, scores.cpu().numpy(), want_ids, rs.numpy())
