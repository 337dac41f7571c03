"""The oracle (oracle/) against the golden vectors generated from the imported
reference (oracle/gen_golden.py) and the reference's own test vectors.  CPU only."""
import hashlib
import json

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import imvoxel_oracle as orc
from helpers import load_npz, load_json, sub, sd_from, meta_from_case


@pytest.mark.parametrize('case', list('ABCDE'))
def test_unprojection_bit_exact(case):
    g = load_npz('backproject_cases.npz')
    c = sub(g, case + '::')
    meta = meta_from_case(c)
    P = orc.compute_projection(meta, 4)
    assert np.array_equal(P, c['projection'])                       # _compute_projection, bit exact
    pts = orc.get_points(c['n_voxels'], c['voxel_size'], c['origin'])
    assert np.array_equal(pts, c['points'])                         # get_points, bit exact
    h, w = meta['img_shape'][0] // 4, meta['img_shape'][1] // 4
    vol, valid, xi, yi = co.backproject(c['feat'], pts, P, h, w, want_idx=True)
    assert np.array_equal(valid, c['valid'])
    m = c['valid'].reshape(xi.shape)
    assert np.array_equal(xi[m], c['xi'][m]) and np.array_equal(yi[m], c['yi'][m])
    # invalid lanes: the reference's .long() of inf/nan is INT64_MIN on x86, restated as such
    assert np.array_equal(xi, c['xi']) and np.array_equal(yi, c['yi'])
    assert np.array_equal(vol, c['volume'])
    mean, ok = co.backproject_mean(c['feat'], pts, P, h, w)
    assert np.array_equal(ok, c['mean_valid'])
    assert np.array_equal(mean, c['mean'])                          # view mean incl. 0-fill, bit exact


def test_unprojection_fullsize_kitti_hashes():
    info = load_json('kitti_fullsize_backproject.json')
    g = torch.Generator().manual_seed(info['seed'])
    feat = torch.randn(tuple(info['feat_shape']), generator=g).numpy()
    assert hashlib.sha256(feat.tobytes()).hexdigest() == info['feat_sha256']
    meta = dict(img_shape=(384, 1280, 3), ori_shape=(384, 1280, 3),
                lidar2img=dict(intrinsic=np.array(info['intrinsic'], np.float32),
                               extrinsic=[np.array(e, np.float32) for e in info['extrinsic']],
                               origin=np.array(info['origin'], np.float32)))
    mean, ok = orc.extract_volume(feat, meta, info['n_voxels'], info['voxel_size'])
    assert hashlib.sha256(np.ascontiguousarray(ok).tobytes()).hexdigest() == info['valid_sha256']
    assert hashlib.sha256(np.ascontiguousarray(mean).tobytes()).hexdigest() == info['mean_sha256']


NECK_TOL = dict(rtol=1e-4, atol=2e-5)   # different conv summation order only (oneDNN vs oneDNN here: ~0)


@pytest.mark.parametrize('name', ['kitti', 'nuscenes', 'fast', 'atlas'])
def test_necks(name):
    g = load_npz('necks.npz')
    sd = sd_from(g, name + '::sd::')
    x = torch.from_numpy(g[name + '::x'])
    with torch.no_grad():
        if name == 'kitti':
            ys = orc.kitti_neck(x, sd)
        elif name == 'nuscenes':
            ys = orc.nuscenes_neck(x, sd)
        elif name == 'fast':
            ys = orc.fast_indoor_neck(x, sd, (1, 1, 1))
        else:
            ys = orc.atlas_neck(x, sd, [4, 8, 16], [1, 2, 2], [2, 1])
    for i, y in enumerate(ys):
        np.testing.assert_allclose(y.numpy(), g[f'{name}::y{i}'], **NECK_TOL)


def test_direct_conv_c_matches_torch():
    """The plain-C direct convolution against torch (used as the small-case conv oracle)."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 6, 7, 4, generator=g)
    w = torch.randn(7, 5, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(7, generator=g)
    for stride, pad in [((1, 1, 1), (1, 1, 1)), ((1, 1, 2), (1, 1, 1)), ((2, 2, 2), (1, 1, 1)), ((1, 1, 1), (0, 0, 0)),
                        ((1, 1, 1), (1, 1, 0))]:
        ref = torch.nn.functional.conv3d(x, w, b, stride, pad).numpy()
        got = co.conv3d(x.numpy(), w.numpy(), b.numpy(), stride, pad)
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5)


def test_box_utils():
    g = load_npz('box_utils.npz')
    val = torch.from_numpy(g['limit_period::val'])
    assert np.array_equal(orc.limit_period(val, 0.5, np.pi).numpy(), g['limit_period::o0.5'])
    assert np.array_equal(orc.limit_period(val, 1, np.pi).numpy(), g['limit_period::o1'])
    assert np.array_equal(orc.limit_period(val, 0, 2 * np.pi).numpy(), g['limit_period::o0'])
    assert np.array_equal(orc.xywhr2xyxyr(torch.from_numpy(g['xywhr::in'])).numpy(), g['xywhr::out'])
    r = orc.rotation_3d_in_axis_z(torch.from_numpy(g['rot::points']), torch.from_numpy(g['rot::angles']))
    np.testing.assert_allclose(r.numpy(), g['rot::axis2'], rtol=1e-6, atol=1e-6)
    d = orc.decode_boxes(torch.from_numpy(g['coder::anchors']), torch.from_numpy(g['coder::deltas']))
    assert np.array_equal(d.numpy(), g['coder::decoded'])


def test_anchor_grid_fullsize_hashes():
    info = load_json('anchors_fullsize.json')
    for name, c in info.items():
        a = orc.grid_anchors(tuple(c['featmap']), c['ranges'], c['sizes'], c['rotations']).numpy()
        assert list(a.shape) == c['shape']
        assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == c['sha256'], name
        assert np.array_equal(a[::9973], np.array(c['strided'], np.float32))


def test_reference_test_vectors_aligned_nms():
    g = load_npz('nms_vectors.npz')
    pick = orc.aligned_3d_nms(torch.from_numpy(g['aligned::boxes']), torch.from_numpy(g['aligned::scores']),
                              torch.from_numpy(g['aligned::classes']), float(g['aligned::thresh']))
    assert np.array_equal(pick.numpy(), g['aligned::pick'])         # tests/test_nms.py expected_pick
    for i in range(4):
        p = f'aligned_rand{i}::'
        pick = orc.aligned_3d_nms(torch.from_numpy(g[p + 'boxes']), torch.from_numpy(g[p + 'scores']),
                                  torch.from_numpy(g[p + 'classes']), 0.25)
        assert np.array_equal(pick.numpy(), g[p + 'pick'])


def test_reference_test_vectors_rotated_overlap():
    """tests/test_box3d.py::test_boxes3d_overlaps known answers (rtol 1e-4, atol 1e-7 as in the reference):
    iou3d = bev_overlap * h_overlap / (v1 + v2 - overlap)   (base_box3d.py:419-443)."""
    g = load_npz('nms_vectors.npz')
    b1, b2 = torch.from_numpy(g['overlaps::boxes1_tensor']), torch.from_numpy(g['overlaps::boxes2_tensor'])

    def bev(b):
        return orc.xywhr2xyxyr(b[:, [0, 1, 3, 4, 6]])

    ov = torch.from_numpy(co.boxes_overlap_bev(bev(b1).numpy(), bev(b2).numpy()))
    top1, bot1 = (b1[:, 2] + b1[:, 5]).view(-1, 1), b1[:, 2].view(-1, 1)
    top2, bot2 = (b2[:, 2] + b2[:, 5]).view(1, -1), b2[:, 2].view(1, -1)
    oh = torch.clamp(torch.min(top1, top2) - torch.max(bot1, bot2), min=0)
    o3 = ov * oh
    v1 = (b1[:, 3] * b1[:, 4] * b1[:, 5]).view(-1, 1)
    v2 = (b2[:, 3] * b2[:, 4] * b2[:, 5]).view(1, -1)
    iou = o3 / torch.clamp(v1 + v2 - o3, min=1e-8)
    iof = o3 / torch.clamp(v1, min=1e-8)
    assert torch.allclose(torch.from_numpy(g['overlaps::expected_iou_tensor']), iou, rtol=1e-4, atol=1e-7)
    assert torch.allclose(torch.from_numpy(g['overlaps::expected_iof_tensor']), iof, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('name', ['kitti', 'nus'])
def test_anchor_head_tail(name):
    g = load_npz('anchor_head.npz')
    p = name + '::'
    sd = sd_from(g, p + 'sd::')
    x = torch.from_numpy(g[p + 'x'])
    with torch.no_grad():
        cls, reg, dr = orc.anchor_head_forward(x, sd, prefix='')
    np.testing.assert_allclose(cls.numpy(), g[p + 'cls'], rtol=1e-5, atol=1e-6)
    anchors = orc.grid_anchors(cls.shape[-2:], g[p + 'ranges'].tolist(), g[p + 'sizes'].tolist(), [0, 1.57])
    assert np.array_equal(anchors.numpy(), g[p + 'anchors'])
    cfg = json.loads(str(g[p + 'test_cfg']))
    for b in range(2):
        # feed the reference's own head outputs so the tail is compared in isolation
        boxes, scores, labels = orc.anchor_head_get_bboxes_single(
            torch.from_numpy(g[p + 'cls'][b]), torch.from_numpy(g[p + 'reg'][b]), torch.from_numpy(g[p + 'dir'][b]),
            anchors, 1, cfg)
        assert np.array_equal(boxes.numpy(), g[p + f'boxes{b}'])
        assert np.array_equal(scores.numpy(), g[p + f'scores{b}'])
        assert np.array_equal(labels.numpy(), g[p + f'labels{b}'])


def test_e2e_small_orchestration():
    g = load_npz('e2e_small.npz')
    sd = sd_from(g, 'sd::')
    cfg = json.loads(str(g['test_cfg']))
    fpn0 = g['fpn0'].reshape((2, 1) + g['fpn0'].shape[1:])
    vols, valids = [], []
    for b in range(2):
        m = sub(g, f'meta{b}::')
        meta = meta_from_case(m)
        v, ok = orc.extract_volume(fpn0[b], meta, g['n_voxels'], g['voxel_size'])
        vols.append(torch.from_numpy(v))
        valids.append(ok)
    assert np.array_equal(np.stack(valids), g['valids'])
    with torch.no_grad():
        y = orc.kitti_neck(torch.stack(vols), sd, 'neck_3d.')[0]
        np.testing.assert_allclose(y.numpy(), g['neck_out'], rtol=1e-4, atol=1e-5)
        cls, reg, dr = orc.anchor_head_forward(y, sd)
        anchors = orc.grid_anchors(cls.shape[-2:], g['ranges'].tolist(), [[1.6, 3.9, 1.56]], [0, 1.57])
        for b in range(2):
            boxes, scores, labels = orc.anchor_head_get_bboxes_single(cls[b], reg[b], dr[b], anchors, 1, cfg)
            np.testing.assert_allclose(boxes.numpy(), g[f'res{b}::boxes'], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(scores.numpy(), g[f'res{b}::scores'], rtol=1e-5, atol=1e-6)
            assert np.array_equal(labels.numpy(), g[f'res{b}::labels'])


@pytest.mark.parametrize('name', ['scannet_v2', 'sunrgbd_v2', 'scannet_v1'])
def test_indoor_heads(name):
    g = load_npz('indoor_heads.npz')
    p = name + '::'
    sd = sd_from(g, p + 'sd::')
    kw = json.loads(str(g[p + 'kw']))
    cfg = json.loads(str(g[p + 'test_cfg']))
    xs = [torch.from_numpy(g[p + f'x{l}']) for l in range(3)]
    with torch.no_grad():
        cs, bs, ss = orc.fcos_head_forward(xs, sd, kw['n_reg_outs'], kw.get('n_convs', 0))
    for l in range(3):
        np.testing.assert_allclose(cs[l].numpy(), g[p + f'centerness{l}'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(bs[l].numpy(), g[p + f'bbox_pred{l}'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ss[l].numpy(), g[p + f'cls{l}'], rtol=1e-5, atol=1e-6)
    valid = torch.from_numpy(g[p + 'valid'])
    for b in range(2):
        boxes, scores, labels = orc.fcos_get_bboxes_single(
            [torch.from_numpy(g[p + f'centerness{l}'][b]) for l in range(3)], [torch.from_numpy(g[p + f'bbox_pred{l}'][b]) for l in range(3)],
            [torch.from_numpy(g[p + f'cls{l}'][b]) for l in range(3)], valid[b], g[p + f'origin{b}'], (.16, .16, .16), kw['n_reg_outs'], cfg)
        assert np.array_equal(labels.numpy(), g[p + f'labels{b}'])
        np.testing.assert_allclose(scores.numpy(), g[p + f'scores{b}'], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(boxes.numpy(), g[p + f'boxes{b}'], rtol=1e-5, atol=1e-6)


def test_dcn_restatement_reduces_to_plain_conv():
    """Sanity pin of the (unpinned) DCNv2 restatement: zero offsets and mask logits -> 0.5 * ordinary convolution;
    integer offsets -> the shifted convolution tap."""
    g = torch.Generator().manual_seed(70)
    x = torch.randn(2, 6, 9, 11, generator=g)
    w = torch.randn(5, 6, 3, 3, generator=g)
    for stride in (1, 2):
        Ho, Wo = (9 + 2 - 3) // stride + 1, (11 + 2 - 3) // stride + 1
        om = torch.zeros(2, 27, Ho, Wo)
        ref = 0.5 * torch.nn.functional.conv2d(x, w, None, stride, 1)
        np.testing.assert_allclose(orc.modulated_deform_conv2d(x, om, w, stride, 1, 1).numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    om = torch.zeros(2, 27, 9, 11)
    om[:, 18:] = 20.0                      # mask ~ 1
    om[:, 0:18:2] = 1.0                    # every tap shifted one row down
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (1, 1, 0, 2)), w, None, 1, 0)   # rows h..h+2 instead of h-1..h+1
    np.testing.assert_allclose(orc.modulated_deform_conv2d(x, om, w, 1, 1, 1).numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
