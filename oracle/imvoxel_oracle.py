"""numpy / torch-fp32-CPU restatement of the ImVoxelNet forward path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference lines (SamsungLabs/imvoxelnet) it restates.  Geometry / index work is
delegated to the plain-C oracle (c_oracle); dense convolutions use
torch.nn.functional on the CPU in fp32 -- the same library the reference's CPU
path executes -- with parameters addressed by the reference's state-dict names.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import c_oracle as co

BN_EPS = 1e-5


# --------------------------------------------------------------------------
# small helpers
def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))


def bn_eval(x, sd, prefix, eps=BN_EPS):
    """nn.BatchNorm{2,3}d in eval mode (running statistics)."""
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                        sd[prefix + '.weight'], sd[prefix + '.bias'], False, 0.0, eps)


def conv3(x, sd, name, stride=1, padding=1):
    return F.conv3d(x, sd[name + '.weight'], sd.get(name + '.bias'), stride, padding)


# --------------------------------------------------------------------------
# unprojection (detectors/imvoxelnet.py:58-76, 114-160)
def get_points(n_voxels, voxel_size, origin):
    return co.get_points(n_voxels, voxel_size, origin)


def compute_projection(img_meta, stride=4):
    """detectors/imvoxelnet.py:114-129 (angles=None branch)."""
    l2i = img_meta['lidar2img']
    ratio = img_meta['ori_shape'][0] / (img_meta['img_shape'][0] / stride)
    return co.compute_projection(np.asarray(l2i['intrinsic'], np.float32), l2i['extrinsic'], ratio)


def backproject(features, points, projection, height=None, width=None):
    return co.backproject(features, points, projection, height, width)


def extract_volume(feature, img_meta, n_voxels, voxel_size, stride=4):
    """One sample of the loop at detectors/imvoxelnet.py:58-76.
    feature: [V,C,H/4,W/4] (full, uncropped FPN level-0 map)."""
    P = compute_projection(img_meta, stride)
    pts = get_points(n_voxels, voxel_size, img_meta['lidar2img']['origin'])
    h = img_meta['img_shape'][0] // stride
    w = img_meta['img_shape'][1] // stride
    return co.backproject_mean(np.asarray(feature, np.float32), pts, P, h, w)


# --------------------------------------------------------------------------
# 3-D necks (necks/imvoxelnet.py)
def basic_block3d(x, sd, p):
    """BasicBlock3d.forward, necks/imvoxelnet.py:209-230 (drop=0, no downsample)."""
    out = F.relu(bn_eval(conv3(x, sd, p + '.conv1'), sd, p + '.bn1'))
    out = bn_eval(conv3(out, sd, p + '.conv2'), sd, p + '.bn2')
    return F.relu(out + x)


def conv_bn_relu3d(x, sd, p, stride, padding):
    """_get_conv, necks/imvoxelnet.py:108-113: Conv3d(bias) -> BN -> ReLU (Sequential idx 0,1,2)."""
    return F.relu(bn_eval(conv3(x, sd, p + '.0', stride, padding), sd, p + '.1'))


def kitti_neck(x, sd, prefix=''):
    """KittiImVoxelNeck.forward, necks/imvoxelnet.py:94-120."""
    m = prefix + 'model.'
    x = basic_block3d(x, sd, m + '0')
    x = conv_bn_relu3d(x, sd, m + '1', (1, 1, 2), (1, 1, 1))
    x = basic_block3d(x, sd, m + '2')
    x = conv_bn_relu3d(x, sd, m + '3', (1, 1, 2), (1, 1, 1))
    x = basic_block3d(x, sd, m + '4')
    x = conv_bn_relu3d(x, sd, m + '5', 1, 0)
    assert x.shape[-1] == 1
    return [x[..., 0].transpose(-1, -2)]


def nuscenes_neck(x, sd, prefix=''):
    """NuScenesImVoxelNeck.forward, necks/imvoxelnet.py:126-151."""
    m = prefix + 'model.'
    x = basic_block3d(x, sd, m + '0')
    x = conv_bn_relu3d(x, sd, m + '1', 2, 1)
    x = basic_block3d(x, sd, m + '2')
    x = conv_bn_relu3d(x, sd, m + '3', (1, 1, 2), (1, 1, 1))
    x = basic_block3d(x, sd, m + '4')
    x = conv_bn_relu3d(x, sd, m + '5', 1, (1, 1, 0))
    assert x.shape[-1] == 1
    return [x[..., 0].transpose(-1, -2)]


def basic_block3d_v2(x, sd, p, stride):
    """BasicBlock3dV2.forward, necks/imvoxelnet.py:233-260."""
    out = F.relu(bn_eval(conv3(x, sd, p + '.conv1', stride, 1), sd, p + '.norm1'))
    out = bn_eval(conv3(out, sd, p + '.conv2', 1, 1), sd, p + '.norm2')
    idt = x
    if stride != 1:
        idt = bn_eval(F.conv3d(x, sd[p + '.downsample.0.weight'], None, stride, 0), sd, p + '.downsample.1')
    return F.relu(out + idt)


def fast_indoor_neck(x, sd, n_blocks=(1, 1, 1), prefix=''):
    """FastIndoorImVoxelNeck.forward, necks/imvoxelnet.py:8-64."""
    n_scales = len(n_blocks)
    down = []
    for i in range(n_scales):
        stride = 1 if i == 0 else 2
        for j in range(n_blocks[i]):
            x = basic_block3d_v2(x, sd, f'{prefix}down_layer_{i}.{j}', stride if j == 0 else 1)
        down.append(x)
    outs = []
    for i in range(n_scales - 1, -1, -1):
        if i < n_scales - 1:
            p = f'{prefix}up_block_{i + 1}'
            x = F.conv_transpose3d(x, sd[p + '.0.weight'], None, 2)
            x = F.relu(bn_eval(x, sd, p + '.1'))
            x = F.relu(bn_eval(conv3(x, sd, p + '.3'), sd, p + '.4'))
            x = down[i] + x
        p = f'{prefix}out_block_{i}'
        outs.append(F.relu(bn_eval(conv3(x, sd, p + '.0'), sd, p + '.1')))
    return outs[::-1]


def atlas_neck(x, sd, channels, layers_down, layers_up, prefix=''):
    """ImVoxelNeck / EncoderDecoder (cond_proj=False), necks/imvoxelnet.py:70-91, 297-372."""
    m = prefix + 'model.'
    xs = []
    for i in range(len(channels)):
        p = f'{m}layers_down.{i}'
        k = 0
        if i > 0:
            x = F.relu(bn_eval(conv3(x, sd, p + '.0', 2, 1), sd, p + '.1'))
            k = 4  # conv, norm, dropout, relu precede the blocks
        for j in range(layers_down[i]):
            x = basic_block3d(x, sd, f'{p}.{k + j}')
        xs.append(x)
    xs = xs[::-1]
    out = []
    for i in range(len(channels) - 1):
        x = F.interpolate(x, scale_factor=2, mode='trilinear', align_corners=False)
        x = F.conv3d(x, sd[f'{m}layers_up_conv.{i}.weight'])
        pj = f'{m}proj.{i}'
        y = F.relu(bn_eval(F.conv3d(xs[i + 1], sd[pj + '.conv.weight']), sd, pj + '.norm'))
        x = (x + y) / 2
        for j in range(layers_up[i]):
            x = basic_block3d(x, sd, f'{m}layers_up_res.{i}.{j}')
        out.append(x)
    out = out[::-1]
    res = []
    for i in range(len(out)):
        p = f'{prefix}conv_blocks.{i}'
        res.append(F.relu(bn_eval(conv3(out[i], sd, p + '.0'), sd, p + '.1')))
    return res


# --------------------------------------------------------------------------
# 2-D trunk: ResNet-50 (style='pytorch', norm_eval) + FPN  -- PARITY UNPINNED
# (mmdet 2.10.0 / torchvision sources are absent; restated from the public
#  architecture; call sites detectors/imvoxelnet.py:48,50, cfg imvoxelnet_kitti.py:4-17)
RESNET50_BLOCKS = (3, 4, 6, 3)


def _bn2(x, sd, p):
    return bn_eval(x, sd, p)


def modulated_deform_conv2d(x, raw_offset_mask, weight, stride=1, pad=1, dil=1):
    """DCNv2 (mmcv-full 1.2.7 ModulatedDeformConv2dPack, deform_groups=1) restated in torch: PARITY UNPINNED (mmcv is
    not in the reference tree).  x [B,C,H,W]; raw_offset_mask [B,3*K,Ho,Wo] = conv_offset(x): channels 2k / 2k+1 are
    (dh, dw) of tap k, channels 2K+k the mask logits; weight [Co,C,kh,kw]."""
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    K = kh * kw
    Ho, Wo = raw_offset_mask.shape[-2:]
    mask = torch.sigmoid(raw_offset_mask[:, 2 * K:3 * K])
    hs = torch.arange(Ho, dtype=torch.float32).view(1, Ho, 1) * stride - pad
    ws = torch.arange(Wo, dtype=torch.float32).view(1, 1, Wo) * stride - pad
    xf = x.reshape(B, C, H * W)
    out = torch.zeros(B, Co, Ho, Wo)
    for k in range(K):
        i, j = k // kw, k % kw
        h_im = hs + i * dil + raw_offset_mask[:, 2 * k]
        w_im = ws + j * dil + raw_offset_mask[:, 2 * k + 1]
        inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        h_low, w_low = torch.floor(h_im), torch.floor(w_im)
        lh, lw = h_im - h_low, w_im - w_low
        val = torch.zeros(B, C, Ho, Wo)
        for (hy, wx, wt, ok) in ((h_low, w_low, (1 - lh) * (1 - lw), (h_low >= 0) & (w_low >= 0)),
                                 (h_low, w_low + 1, (1 - lh) * lw, (h_low >= 0) & (w_low + 1 <= W - 1)),
                                 (h_low + 1, w_low, lh * (1 - lw), (h_low + 1 <= H - 1) & (w_low >= 0)),
                                 (h_low + 1, w_low + 1, lh * lw, (h_low + 1 <= H - 1) & (w_low + 1 <= W - 1))):
            ok = ok & inside
            lin = (hy.clamp(0, H - 1) * W + wx.clamp(0, W - 1)).long().view(B, 1, Ho * Wo).expand(B, C, Ho * Wo)
            g = torch.gather(xf, 2, lin).view(B, C, Ho, Wo)
            val = val + g * (wt * ok.float()).unsqueeze(1)
        val = val * mask[:, k:k + 1]
        out = out + torch.einsum('oc,bchw->bohw', weight[:, :, i, j], val)
    return out


def resnet50(x, sd, prefix='backbone.'):
    x = F.relu(_bn2(F.conv2d(x, sd[prefix + 'conv1.weight'], None, 2, 3), sd, prefix + 'bn1'))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, nb in enumerate(RESNET50_BLOCKS):
        for bi in range(nb):
            p = f'{prefix}layer{li + 1}.{bi}'
            stride = 2 if (bi == 0 and li > 0) else 1
            idt = x
            o = F.relu(_bn2(F.conv2d(x, sd[p + '.conv1.weight']), sd, p + '.bn1'))
            if p + '.conv2.conv_offset.weight' in sd:      # DCNv2 stage (nuScenes reference backbone)
                om = F.conv2d(o, sd[p + '.conv2.conv_offset.weight'], sd[p + '.conv2.conv_offset.bias'], stride, 1)
                o = F.relu(_bn2(modulated_deform_conv2d(o, om, sd[p + '.conv2.weight'], stride, 1, 1), sd, p + '.bn2'))
            else:
                o = F.relu(_bn2(F.conv2d(o, sd[p + '.conv2.weight'], None, stride, 1), sd, p + '.bn2'))
            o = _bn2(F.conv2d(o, sd[p + '.conv3.weight']), sd, p + '.bn3')
            if bi == 0:
                idt = _bn2(F.conv2d(x, sd[p + '.downsample.0.weight'], None, stride), sd, p + '.downsample.1')
            x = F.relu(o + idt)
        outs.append(x)
    return outs


def fpn_level0(feats, sd, prefix='neck.'):
    """FPN (mmdet): lateral 1x1 -> top-down nearest x2 add -> 3x3; only level 0 is
    consumed by the path (detectors/imvoxelnet.py:50)."""
    lat = [F.conv2d(f, sd[f'{prefix}lateral_convs.{i}.conv.weight'], sd[f'{prefix}lateral_convs.{i}.conv.bias'])
           for i, f in enumerate(feats)]
    for i in range(len(lat) - 1, 0, -1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
    return F.conv2d(lat[0], sd[prefix + 'fpn_convs.0.conv.weight'], sd[prefix + 'fpn_convs.0.conv.bias'], 1, 1)


# --------------------------------------------------------------------------
# anchors / coder / box utils
def anchors_single_range(feature_size, anchor_range, sizes, rotations, scale=1):
    """Anchor3DRangeGenerator.anchors_single_range, core/anchor/anchor_3d_generator.py:145-209.
    Returns [D, H, W, n_sizes, n_rot, 7] fp32 (x, y, z, w, l, h, r)."""
    if len(feature_size) == 2:
        feature_size = [1, feature_size[0], feature_size[1]]
    r = torch.tensor(anchor_range, dtype=torch.float32)
    zc = torch.linspace(r[2], r[5], feature_size[0])
    yc = torch.linspace(r[1], r[4], feature_size[1])
    xc = torch.linspace(r[0], r[3], feature_size[2])
    sz = torch.tensor(sizes, dtype=torch.float32).reshape(-1, 3) * scale
    rot = torch.tensor(rotations, dtype=torch.float32)
    D, H, W, S, R = len(zc), len(yc), len(xc), sz.shape[0], len(rot)
    out = torch.empty(D, H, W, S, R, 7)
    out[..., 0] = xc.view(1, 1, W, 1, 1)
    out[..., 1] = yc.view(1, H, 1, 1, 1)
    out[..., 2] = zc.view(D, 1, 1, 1, 1)
    out[..., 3:6] = sz.view(1, 1, 1, S, 1, 3)
    out[..., 6] = rot.view(1, 1, 1, 1, R)
    return out


def grid_anchors(featmap_size, ranges, sizes, rotations, size_per_range=True):
    """grid_anchors / single_level_grid_anchors (:82-143) for one level, reshape_out=True."""
    if size_per_range:
        if len(sizes) != len(ranges):
            ranges = ranges * len(sizes)
        parts = [anchors_single_range(featmap_size, rg, sz, rotations) for rg, sz in zip(ranges, sizes)]
        a = torch.cat(parts, dim=-3)
    else:
        a = anchors_single_range(featmap_size, ranges[0], sizes, rotations)
    return a.reshape(-1, a.shape[-1])


def decode_boxes(anchors, deltas):
    """DeltaXYZWLHRBBoxCoder.decode, core/bbox/coders/delta_xyzwhlr_bbox_coder.py:56-90."""
    xa, ya, za, wa, la, ha, ra = [anchors[:, i] for i in range(7)]
    xt, yt, zt, wt, lt, ht, rt = [deltas[:, i] for i in range(7)]
    za = za + ha / 2
    diag = torch.sqrt(la ** 2 + wa ** 2)
    xg = xt * diag + xa
    yg = yt * diag + ya
    zg = zt * ha + za
    lg = torch.exp(lt) * la
    wg = torch.exp(wt) * wa
    hg = torch.exp(ht) * ha
    rg = rt + ra
    zg = zg - hg / 2
    return torch.stack([xg, yg, zg, wg, lg, hg, rg], dim=-1)


def limit_period(val, offset=0.5, period=np.pi):
    """core/bbox/structures/utils.py:5-18."""
    return val - torch.floor(val / period + offset) * period


def xywhr2xyxyr(b):
    """core/bbox/structures/utils.py:64-82."""
    out = torch.zeros_like(b)
    hw, hh = b[:, 2] / 2, b[:, 3] / 2
    out[:, 0] = b[:, 0] - hw
    out[:, 1] = b[:, 1] - hh
    out[:, 2] = b[:, 0] + hw
    out[:, 3] = b[:, 1] + hh
    out[:, 4] = b[:, 4]
    return out


def rotation_3d_in_axis_z(points, angles):
    """core/bbox/structures/utils.py:21-61 with axis=2: points [N,M,3], angles [N]."""
    s, c = torch.sin(angles), torch.cos(angles)
    x = points[..., 0] * c[:, None] + points[..., 1] * s[:, None]
    y = -points[..., 0] * s[:, None] + points[..., 1] * c[:, None]
    return torch.stack([x, y, points[..., 2]], dim=-1)


# --------------------------------------------------------------------------
# NMS
def nms_gpu(boxes, scores, thresh):
    """ops/iou3d/iou3d_utils.py:25-50 -> iou3d.cpp:95-147 (restated in C)."""
    order = scores.sort(0, descending=True)[1]
    keep = co.nms_sorted(boxes[order].numpy(), thresh, rotated=True)
    return order[torch.from_numpy(keep)].contiguous()


def nms_normal_gpu(boxes, scores, thresh):
    """ops/iou3d/iou3d_utils.py:53-71."""
    order = scores.sort(0, descending=True)[1]
    keep = co.nms_sorted(boxes[order].numpy(), thresh, rotated=False)
    return order[torch.from_numpy(keep)].contiguous()


def box3d_multiclass_nms(bboxes, bboxes_for_nms, scores, score_thr, max_num, use_rotate_nms, nms_thr,
                         dir_scores=None):
    """core/post_processing/box3d_nms.py:8-88."""
    ncls = scores.shape[1] - 1
    ob, os_, ol, od = [], [], [], []
    for i in range(ncls):
        m = scores[:, i] > score_thr
        if not m.any():
            continue
        s = scores[m, i]
        fn = nms_gpu if use_rotate_nms else nms_normal_gpu
        sel = fn(bboxes_for_nms[m], s, nms_thr)
        ob.append(bboxes[m][sel])
        os_.append(s[sel])
        ol.append(torch.full((len(sel),), i, dtype=torch.long))
        if dir_scores is not None:
            od.append(dir_scores[m][sel])
    if ob:
        b, s, l = torch.cat(ob), torch.cat(os_), torch.cat(ol)
        d = torch.cat(od) if dir_scores is not None else None
        if b.shape[0] > max_num:
            inds = s.sort(descending=True)[1][:max_num]
            b, s, l = b[inds], s[inds], l[inds]
            if d is not None:
                d = d[inds]
    else:
        b = scores.new_zeros((0, bboxes.size(-1)))
        s = scores.new_zeros((0,))
        l = scores.new_zeros((0,), dtype=torch.long)
        d = scores.new_zeros((0,))
    return b, s, l, d


def aligned_3d_nms(boxes, scores, classes, thresh):
    """core/post_processing/box3d_nms.py:91-138."""
    order = torch.argsort(scores)
    pick = co.aligned_3d_nms(boxes.numpy(), scores.numpy(), classes.numpy(), order.numpy(), thresh)
    return torch.from_numpy(pick)


# --------------------------------------------------------------------------
# Anchor3DHead (KITTI / nuScenes tail)
def anchor_head_forward(x, sd, prefix='bbox_head.'):
    """Anchor3DHead.forward_single, dense_heads/anchor3d_head.py:138-153."""
    cls = F.conv2d(x, sd[prefix + 'conv_cls.weight'], sd[prefix + 'conv_cls.bias'])
    reg = F.conv2d(x, sd[prefix + 'conv_reg.weight'], sd[prefix + 'conv_reg.bias'])
    dr = F.conv2d(x, sd[prefix + 'conv_dir_cls.weight'], sd[prefix + 'conv_dir_cls.bias'])
    return cls, reg, dr


def anchor_head_candidates(cls_score, bbox_pred, dir_cls_pred, anchors, num_classes, nms_pre):
    """First half of get_bboxes_single (:452-492): permute, sigmoid, argmax, top-k, decode.
    Inputs are single-sample [A*ncls,H,W], [A*7,H,W], [A*2,H,W]."""
    dir_cls = dir_cls_pred.permute(1, 2, 0).reshape(-1, 2)
    dir_score = torch.max(dir_cls, dim=-1)[1]
    scores = cls_score.permute(1, 2, 0).reshape(-1, num_classes).sigmoid()
    reg = bbox_pred.permute(1, 2, 0).reshape(-1, 7)
    topk = None
    if nms_pre > 0 and scores.shape[0] > nms_pre:
        mx, _ = scores.max(dim=1)
        _, topk = mx.topk(nms_pre)
        anchors, reg, scores, dir_score = anchors[topk], reg[topk], scores[topk], dir_score[topk]
    boxes = decode_boxes(anchors, reg)
    return boxes, scores, dir_score, topk


def anchor_head_get_bboxes_single(cls_score, bbox_pred, dir_cls_pred, anchors, num_classes, cfg,
                                  dir_offset=0.0, dir_limit_offset=1.0):
    """Anchor3DHead.get_bboxes_single, dense_heads/anchor3d_head.py:428-517 (one level, sigmoid cls).
    cfg keys: nms_pre, score_thr, max_num, nms_thr, use_rotate_nms.  Returns raw [n,7] boxes
    (LiDAR bottom-centre convention, i.e. exactly the tensor wrapped by box_type_3d), scores, labels."""
    boxes, scores, dir_score, _ = anchor_head_candidates(cls_score, bbox_pred, dir_cls_pred, anchors,
                                                         num_classes, cfg.get('nms_pre', -1))
    bev = boxes[:, [0, 1, 3, 4, 6]]                      # LiDARInstance3DBoxes.bev, lidar_box3d.py:86-90
    for_nms = xywhr2xyxyr(bev)
    scores_p = torch.cat([scores, scores.new_zeros(scores.shape[0], 1)], dim=1)
    b, s, l, d = box3d_multiclass_nms(boxes, for_nms, scores_p, cfg.get('score_thr', 0), cfg['max_num'],
                                      cfg['use_rotate_nms'], cfg['nms_thr'], dir_score)
    if b.shape[0] > 0:
        dir_rot = limit_period(b[..., 6] - dir_offset, dir_limit_offset, np.pi)
        b[..., 6] = dir_rot + dir_offset + np.pi * d.to(b.dtype)
    return b, s, l


# --------------------------------------------------------------------------
# whole-path restatement for the anchor-head (KITTI / nuScenes) configuration
def simple_test_anchor(img, img_metas, sd, cfg):
    """ImVoxelNet.simple_test (detectors/imvoxelnet.py:93-106) for ResNet-50 + FPN +
    {Kitti,NuScenes}ImVoxelNeck + Anchor3DHead.  img [B,V,3,H,W] fp32 torch.  cfg keys:
    n_voxels, voxel_size, neck ('kitti'|'nuscenes'), anchor (ranges,sizes,rotations), test_cfg,
    num_classes, dir_offset / dir_limit_offset (anchor3d_head.py:61-62,511-515, defaults 0 / 1).  Returns list of (boxes[n,7], scores[n], labels[n]) and intermediates."""
    B = img.shape[0]
    x = img.reshape([-1] + list(img.shape[2:]))
    with torch.no_grad():
        feats = resnet50(x, sd)
        f0 = fpn_level0(feats, sd)
        f0 = f0.reshape([B, -1] + list(f0.shape[1:]))
        vols, valids = [], []
        for b in range(B):
            v, m = extract_volume(f0[b].numpy(), img_metas[b], cfg['n_voxels'], cfg['voxel_size'])
            vols.append(torch.from_numpy(v))
            valids.append(torch.from_numpy(m))
        vol = torch.stack(vols)
        neck = kitti_neck if cfg.get('neck', 'kitti') == 'kitti' else nuscenes_neck
        y = neck(vol, sd, 'neck_3d.')[0]
        cls, reg, dr = anchor_head_forward(y, sd)
        a = cfg['anchor']
        anchors = grid_anchors(cls.shape[-2:], a['ranges'], a['sizes'], a['rotations'])
        res = [anchor_head_get_bboxes_single(cls[b], reg[b], dr[b], anchors, cfg.get('num_classes', 1),
                                             cfg['test_cfg'], cfg.get('dir_offset', 0.0),
                                             cfg.get('dir_limit_offset', 1.0)) for b in range(B)]
    return res, dict(fpn0=f0, volume=vol, valids=torch.stack(valids), neck=y, cls=cls, reg=reg, dir=dr)


# --------------------------------------------------------------------------
# synthetic parameters (shared by tests, smoke and bench so HIP and oracle see the same numbers)
def _kaiming(gen, shape):
    fan_in = int(np.prod(shape[1:]))
    return torch.randn(shape, generator=gen) * math.sqrt(2.0 / fan_in)


def _bn_params(gen, sd, p, c):
    sd[p + '.weight'] = torch.rand(c, generator=gen) + 0.5
    sd[p + '.bias'] = torch.randn(c, generator=gen) * 0.1
    sd[p + '.running_mean'] = torch.randn(c, generator=gen) * 0.1
    sd[p + '.running_var'] = torch.rand(c, generator=gen) + 0.5


# --------------------------------------------------------------------------
# anchor-free indoor heads (dense_heads/imvoxel_head_v2.py, imvoxel_head.py)
def fcos_head_forward(xs, sd, n_reg, n_convs=0, prefix=''):
    """forward / forward_single (v2:57-58,305-313,444-449; v1:78-79,326-336,454-461): per level
    (centerness, exp(scale*reg[:6]) [+ raw angle], cls)."""
    cs, bs, ss = [], [], []
    for lvl, x in enumerate(xs):
        r, c = x, x
        for i in range(n_convs):
            r = F.relu(bn_eval(F.conv3d(r, sd[f'{prefix}reg_convs.{i}.0.weight'], None, 1, 1), sd, f'{prefix}reg_convs.{i}.1'))
            c = F.relu(bn_eval(F.conv3d(c, sd[f'{prefix}cls_convs.{i}.0.weight'], None, 1, 1), sd, f'{prefix}cls_convs.{i}.1'))
        reg = F.conv3d(r, sd[prefix + 'reg_conv.weight'], None, 1, 1)
        d = torch.exp(reg[:, :6] * sd[f'{prefix}scales.{lvl}.scale'])
        cs.append(F.conv3d(r, sd[prefix + 'centerness_conv.weight'], None, 1, 1))
        bs.append(d if n_reg == 6 else torch.cat((d, reg[:, 6:7]), dim=1))
        ss.append(F.conv3d(c, sd[prefix + 'cls_conv.weight'], sd[prefix + 'cls_conv.bias'], 1, 1))
    return cs, bs, ss


def fcos_get_bboxes_single(cs, bs, ss, valid, origin, voxel_size, n_reg, cfg, return_candidates=False):
    """get_bboxes + _get_bboxes_single for ONE sample (v2:216-285): cs/bs/ss are per-level [C,nx,ny,nz] tensors,
    valid [1,X,Y,Z] float.  Returns (boxes [n, 7], scores, labels) with the box tensor as the box object holds it.
    return_candidates: additionally (cand_boxes [m,R], cand_scores [m,ncls], cand_index [m] = level * 2^32 + flat voxel
    index) -- the concatenated per-level top-k lists the NMS works on (index-parity tests)."""
    n_classes = ss[0].shape[0]
    mb, ms, mi = [], [], []
    for lvl, (c, b, s) in enumerate(zip(cs, bs, ss)):
        shape = c.shape[-3:]
        v = F.interpolate(valid[None], size=tuple(shape), mode='trilinear', align_corners=False)[0].round().bool()
        pts = torch.from_numpy(get_points(list(shape), np.asarray(voxel_size, np.float32) * np.float32(2 ** lvl), origin))
        pts = pts.reshape(3, -1).transpose(0, 1)
        ctr = c.permute(1, 2, 3, 0).reshape(-1).sigmoid()
        bp = b.permute(1, 2, 3, 0).reshape(-1, n_reg)
        sc = s.permute(1, 2, 3, 0).reshape(-1, n_classes).sigmoid()
        vf = v.permute(1, 2, 3, 0).reshape(-1)
        sc = sc * ctr[:, None] * vf[:, None]
        mx, _ = sc.max(dim=1)
        ids = torch.arange(len(sc))
        if len(sc) > cfg['nms_pre'] > 0:
            _, ids = mx.topk(cfg['nms_pre'])
            bp, sc, pts = bp[ids], sc[ids], pts[ids]
        mi.append(ids.to(torch.int64) + (lvl << 32))
        if n_reg == 6:      # ScanNet (v2:547-555)
            box = torch.stack([pts[:, 0] - bp[:, 0], pts[:, 1] - bp[:, 2], pts[:, 2] - bp[:, 4],
                               pts[:, 0] + bp[:, 1], pts[:, 1] + bp[:, 3], pts[:, 2] + bp[:, 5]], -1)
        else:               # SUN RGB-D (v2:419-438)
            shift = torch.stack(((bp[:, 1] - bp[:, 0]) / 2, (bp[:, 3] - bp[:, 2]) / 2, (bp[:, 5] - bp[:, 4]) / 2), dim=-1).view(-1, 1, 3)
            shift = rotation_3d_in_axis_z(shift, bp[:, 6])[:, 0, :]
            size = torch.stack((bp[:, 0] + bp[:, 1], bp[:, 2] + bp[:, 3], bp[:, 4] + bp[:, 5]), dim=-1)
            box = torch.cat((pts + shift, size, bp[:, 6:7]), dim=-1)
        mb.append(box)
        ms.append(sc)
    boxes, scores = torch.cat(mb), torch.cat(ms)
    cand = (boxes.clone(), scores.clone(), torch.cat(mi))
    if n_reg == 6:          # ScanNet _nms (v2:528-545)
        scores, labels = scores.max(dim=1)
        ids = scores > cfg['score_thr']
        boxes, scores, labels = boxes[ids], scores[ids], labels[ids]
        ids = aligned_3d_nms(boxes, scores, labels, cfg['iou_thr'])
        boxes = boxes[ids]
        boxes = torch.stack(((boxes[:, 0] + boxes[:, 3]) / 2., (boxes[:, 1] + boxes[:, 4]) / 2., (boxes[:, 2] + boxes[:, 5]) / 2.,
                             boxes[:, 3] - boxes[:, 0], boxes[:, 4] - boxes[:, 1], boxes[:, 5] - boxes[:, 2]), dim=1)
        boxes = torch.cat((boxes, boxes.new_zeros(boxes.shape[0], 1)), dim=-1)       # box_dim 6 -> fake yaw (base_box3d.py:52-58)
        scores, labels = scores[ids], labels[ids]
    else:                   # SUN RGB-D _nms (v2:397-417)
        scores_p = torch.cat([scores, scores.new_zeros(scores.shape[0], 1)], dim=1)
        for_nms = torch.stack((boxes[:, 0] - boxes[:, 3] / 2, boxes[:, 1] - boxes[:, 4] / 2,
                               boxes[:, 0] + boxes[:, 3] / 2, boxes[:, 1] + boxes[:, 4] / 2, boxes[:, 6]), dim=1)
        boxes, scores, labels, _ = box3d_multiclass_nms(boxes, for_nms, scores_p, cfg['score_thr'], cfg['nms_pre'],
                                                        cfg['use_rotate_nms'], cfg['nms_thr'])
    boxes = boxes.clone()
    boxes[:, 2] = boxes[:, 2] - boxes[:, 5] * 0.5                                   # origin (.5,.5,.5) -> (.5,.5,0) (base_box3d.py:63-66)
    if return_candidates:
        return boxes, scores, labels, cand
    return boxes, scores, labels
