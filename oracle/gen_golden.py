"""Generate tests/golden/*.npz from the REAL reference imported on CPU.

Run in the build container only:  python -m oracle.gen_golden
(needs /root/reference; see oracle/ref_import.py).  The fixtures are data:
seeded inputs and the outputs the reference code produced for them, plus the
vectors held by the reference's own tests (captured by executing those tests
with the function under test wrapped, or by reading the literals with `ast`).

The reference's CUDA op `nms_gpu` cannot run here; wherever a fixture's
outputs depend on it the C restatement (oracle/ivx_oracle.c) is injected and
the fixture says so in `nms_source`.
"""
import ast
import hashlib
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import, imvoxel_oracle as orc  # noqa: E402

warnings.filterwarnings('ignore')
GOLD = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(GOLD, exist_ok=True)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sd_to_np(sd, prefix='sd::'):
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items() if 'num_batches_tracked' not in k}


def randomize_bn(module, gen):
    for m in module.modules():
        if isinstance(m, (torch.nn.BatchNorm3d, torch.nn.BatchNorm2d)):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)


# ------------------------------------------------------------------ cameras
def look_at(eye, target, up=(0, 0, 1)):
    """world->camera 4x4 (camera looks along +z, x right, y down)."""
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, up)
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    R = np.stack([r, d, f])
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = -R @ eye
    return E.astype(np.float32)


def kitti_like_meta(img_shape, ori_shape, t=(0.05, 0.1, 0.27)):
    K = np.array([[721.5377, 0, 609.5593, 0], [0, 721.5377, 172.854, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    E = np.array([[0, -1, 0, t[0]], [0, 0, -1, t[1]], [1, 0, 0, t[2]], [0, 0, 0, 1]], np.float32)
    return dict(img_shape=img_shape, ori_shape=ori_shape,
                lidar2img=dict(intrinsic=K, extrinsic=[E], origin=np.array([34.56, 0, -1], np.float32)))


def backproject_cases():
    cases = {}
    # A: KITTI-like single view, padded crop (img_shape < feature map), real-image ratio
    K = np.array([[60., 0, 15.5, 0], [0, 60., 11.2, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    E = np.array([[0, -1, 0, 0.02], [0, 0, -1, 0.4], [1, 0, 0, 0.3], [0, 0, 0, 1]], np.float32)
    cases['A'] = dict(meta=dict(img_shape=(88, 120, 3), ori_shape=(44, 60, 3),
                                lidar2img=dict(intrinsic=K, extrinsic=[E], origin=np.array([3.0, 0, 0.2], np.float32))),
                      feat_hw=(24, 32), C=8, n_voxels=(16, 12, 6), voxel_size=(.4, .4, .4), seed=0)
    # B: two indoor views, one camera inside the grid (points behind the camera, w<=0)
    Kb = np.array([[28.9, 0, 15.5, 0], [0, 28.9, 11.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    Eb = [look_at((2.0, 0.1, 0.6), (0, 0, 0.5)), look_at((0.2, 0.3, 0.5), (-2, 0.5, 0.4))]
    cases['B'] = dict(meta=dict(img_shape=(96, 128, 3), ori_shape=(96, 128, 3),
                                lidar2img=dict(intrinsic=Kb, extrinsic=Eb, origin=np.array([0, 0, .5], np.float32))),
                      feat_hw=(24, 32), C=8, n_voxels=(12, 12, 6), voxel_size=(.25, .25, .25), seed=1)
    # C: six cameras on a ring looking outwards (nuScenes-like; intrinsic=eye, K folded into extrinsic)
    Kc = np.array([[20., 0, 16, 0], [0, 20., 12, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    Ec = []
    for yaw in np.deg2rad([0, 55, 110, 180, -110, -55]):
        eye = np.array([0.3 * np.cos(yaw), 0.3 * np.sin(yaw), 0.5])
        Ec.append((Kc.astype(np.float64) @ look_at(eye, eye + np.array([np.cos(yaw), np.sin(yaw), -0.05])).astype(np.float64)).astype(np.float32))
    cases['C'] = dict(meta=dict(img_shape=(96, 128, 3), ori_shape=(96, 128, 3),
                                lidar2img=dict(intrinsic=np.eye(4, dtype=np.float32), extrinsic=Ec,
                                               origin=np.array([0, 0, -.2], np.float32))),
                      feat_hw=(24, 32), C=8, n_voxels=(14, 14, 4), voxel_size=(.5, .5, .5), seed=2)
    # D: dyadic intrinsics / exact .5 pixel ties -> round-half-even, bit-exact projection
    Kd = np.array([[8., 0, 8.5, 0], [0, 8., 6.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    Ed = np.array([[0, -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]], np.float32)
    cases['D'] = dict(meta=dict(img_shape=(48, 64, 3), ori_shape=(48, 64, 3),
                                lidar2img=dict(intrinsic=Kd, extrinsic=[Ed], origin=np.array([2.0, 0, 0], np.float32))),
                      feat_hw=(12, 16), C=4, n_voxels=(8, 16, 8), voxel_size=(.5, .25, .25), seed=3)
    # E: camera plane passes exactly through voxel corners (w == 0 -> inf / nan), 3 views, C=5 (odd)
    Ke = np.array([[16., 0, 8, 0], [0, 16., 6, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    Ee = [np.array([[0, -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]], np.float32),
          np.array([[0, -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, -0.5], [0, 0, 0, 1]], np.float32),
          np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 1.0], [0, 0, 0, 1]], np.float32)]
    cases['E'] = dict(meta=dict(img_shape=(48, 64, 3), ori_shape=(48, 64, 3),
                                lidar2img=dict(intrinsic=Ke, extrinsic=Ee, origin=np.array([0, 0, 0], np.float32))),
                      feat_hw=(12, 16), C=5, n_voxels=(8, 8, 4), voxel_size=(.5, .5, .5), seed=4)
    return cases


def gen_backproject(ns):
    det = ns.detector
    out = {}
    for name, c in backproject_cases().items():
        meta = c['meta']
        V = len(meta['lidar2img']['extrinsic'])
        g = torch.Generator().manual_seed(c['seed'])
        feat = torch.randn((V, c['C']) + c['feat_hw'], generator=g)
        P = det.ImVoxelNet._compute_projection(meta, 4, None)
        pts = det.get_points(torch.tensor(c['n_voxels']), torch.tensor(c['voxel_size']),
                             torch.tensor(meta['lidar2img']['origin']))
        h, w = meta['img_shape'][0] // 4, meta['img_shape'][1] // 4
        vol, valid = det.backproject(feat[:, :, :h, :w], pts, P)
        # index maps exactly as the reference computes them (:148-153)
        p = pts.view(1, 3, -1).expand(V, 3, -1)
        p = torch.cat((p, torch.ones_like(p[:, :1])), dim=1)
        p23 = torch.bmm(P, p)
        xi = (p23[:, 0] / p23[:, 2]).round().long()
        yi = (p23[:, 1] / p23[:, 2]).round().long()
        # view mean (:70-74)
        vs = vol.sum(dim=0)
        cnt = valid.sum(dim=0)
        mean = vs / cnt
        ok = cnt > 0
        mean[:, ~ok[0]] = .0
        pre = f'{name}::'
        out.update({pre + 'feat': feat.numpy(), pre + 'intrinsic': meta['lidar2img']['intrinsic'],
                    pre + 'extrinsic': np.stack(meta['lidar2img']['extrinsic']),
                    pre + 'origin': meta['lidar2img']['origin'],
                    pre + 'img_shape': np.array(meta['img_shape']), pre + 'ori_shape': np.array(meta['ori_shape']),
                    pre + 'n_voxels': np.array(c['n_voxels']), pre + 'voxel_size': np.array(c['voxel_size'], np.float32),
                    pre + 'projection': P.numpy(), pre + 'points': pts.numpy(),
                    pre + 'xi': xi.numpy(), pre + 'yi': yi.numpy(), pre + 'valid': valid.numpy(),
                    pre + 'volume': vol.numpy(), pre + 'mean': mean.numpy(), pre + 'mean_valid': ok.numpy()})
        print('backproject', name, 'V', V, 'valid frac', float(valid.float().mean()), 'novalid', int((~ok).sum()))
    np.savez_compressed(os.path.join(GOLD, 'backproject_cases.npz'), **out)


def gen_fullsize_kitti(ns):
    """Full-size KITTI unprojection: only hashes / sums are stored."""
    det = ns.detector
    meta = kitti_like_meta((384, 1280, 3), (384, 1280, 3), t=(0.0, 0.0, 0.0))
    g = torch.Generator().manual_seed(1234)
    feat = torch.randn((1, 64, 96, 320), generator=g)
    P = det.ImVoxelNet._compute_projection(meta, 4, None)
    pts = det.get_points(torch.tensor((216, 248, 12)), torch.tensor((.32, .32, .32)),
                         torch.tensor(meta['lidar2img']['origin']))
    vol, valid = det.backproject(feat, pts, P)
    vs = vol.sum(0)
    cnt = valid.sum(0)
    mean = vs / cnt
    ok = cnt > 0
    mean[:, ~ok[0]] = .0
    info = dict(seed=1234, feat_shape=[1, 64, 96, 320], n_voxels=[216, 248, 12], voxel_size=[.32, .32, .32],
                intrinsic=meta['lidar2img']['intrinsic'].tolist(), extrinsic=[meta['lidar2img']['extrinsic'][0].tolist()],
                origin=[34.56, 0, -1], valid_frac=float(ok.float().mean()), valid_sha256=sha(ok.numpy()),
                mean_sha256=sha(mean.numpy()), mean_sum=float(mean.double().sum()), mean_absmax=float(mean.abs().max()),
                projection=P.numpy().tolist(), feat_sha256=sha(feat.numpy()))
    with open(os.path.join(GOLD, 'kitti_fullsize_backproject.json'), 'w') as f:
        json.dump(info, f, indent=1)
    print('fullsize kitti valid frac', info['valid_frac'])


def gen_necks(ns):
    nk = ns.necks
    out = {}
    g = torch.Generator().manual_seed(10)

    def run(name, module, x):
        torch.manual_seed(hash(name) % 1000)
        randomize_bn(module, g)
        # un-zero the zero-initialised residual BN gammas of the Atlas net so the test has signal
        module.eval()
        with torch.no_grad():
            ys = module(x)
        out.update(sd_to_np(module.state_dict(), f'{name}::sd::'))
        out[f'{name}::x'] = x.numpy()
        for i, y in enumerate(ys):
            out[f'{name}::y{i}'] = y.numpy()
        print('neck', name, [tuple(y.shape) for y in ys], 'absmax', [float(y.abs().max()) for y in ys])

    torch.manual_seed(11)
    run('kitti', nk.KittiImVoxelNeck(4, 8), torch.randn(2, 4, 7, 9, 12, generator=g))
    torch.manual_seed(12)
    run('nuscenes', nk.NuScenesImVoxelNeck(4, 8), torch.randn(1, 4, 10, 12, 12, generator=g))
    torch.manual_seed(13)
    run('fast', nk.FastIndoorImVoxelNeck(4, [1, 1, 1], 8), torch.randn(1, 4, 8, 8, 4, generator=g))
    torch.manual_seed(14)
    run('atlas', nk.ImVoxelNeck([4, 8, 16], 4, [1, 2, 2], [2, 1], False), torch.randn(1, 4, 8, 8, 8, generator=g))
    np.savez_compressed(os.path.join(GOLD, 'necks.npz'), **out)


def gen_anchor_head(ns):
    ah = ns.anchor_head

    class Cfg(dict):
        __getattr__ = dict.get

    ns.nms.nms_gpu = orc.nms_gpu            # CUDA op replaced by the C restatement
    ns.nms.nms_normal_gpu = orc.nms_normal_gpu
    out = {}
    for name, (H, W, ncls, ranges, sizes, test_cfg) in {
        'kitti': (10, 12, 1, [[0, -39.68, -1.78, 69.12 - .32, 39.68 - .32, -1.78]], [[1.6, 3.9, 1.56]],
                  dict(use_rotate_nms=True, nms_across_levels=False, nms_thr=0.01, score_thr=0.1, min_bbox_size=0,
                       nms_pre=100, max_num=50)),
        'nus': (9, 9, 1, [[-49.92, -49.92, -1.8, 49.92 - .64, 49.92 - .64, -1.8]], [[1.95, 4.60, 1.73]],
                dict(use_rotate_nms=True, nms_across_levels=False, nms_thr=0.2, score_thr=0.05, min_bbox_size=0,
                     nms_pre=60, max_num=20)),
    }.items():
        torch.manual_seed(20 if name == 'kitti' else 21)
        head = ah.Anchor3DHead(num_classes=ncls, in_channels=16, train_cfg=None, test_cfg=Cfg(test_cfg),
                               feat_channels=16, use_direction_classifier=True,
                               anchor_generator=dict(type='Anchor3DRangeGenerator', ranges=ranges, sizes=sizes,
                                                     rotations=[0, 1.57], reshape_out=True),
                               diff_rad_by_sin=True, bbox_coder=dict(type='DeltaXYZWLHRBBoxCoder'),
                               loss_cls=dict(type='FocalLoss', use_sigmoid=True))
        with torch.no_grad():
            head.conv_cls.weight.normal_(0, 0.5)
            head.conv_cls.bias.fill_(-1.0)
            head.conv_reg.weight.normal_(0, 0.05)
            head.conv_dir_cls.weight.normal_(0, 0.3)
        x = torch.randn(2, 16, H, W)
        with torch.no_grad():
            cls, reg, dr = head([x])
        metas = [dict(box_type_3d=ns.lidar.LiDARInstance3DBoxes)] * 2
        with torch.no_grad():
            res = head.get_bboxes(cls, reg, dr, None, metas)
        anchors = head.anchor_generator.grid_anchors([cls[0].shape[-2:]], device='cpu')[0]
        pre = f'{name}::'
        out.update(sd_to_np(head.state_dict(), pre + 'sd::'))
        out.update({pre + 'x': x.numpy(), pre + 'cls': cls[0].numpy(), pre + 'reg': reg[0].numpy(),
                    pre + 'dir': dr[0].numpy(), pre + 'anchors': anchors.numpy(),
                    pre + 'ranges': np.array(ranges, np.float64), pre + 'sizes': np.array(sizes, np.float64),
                    pre + 'test_cfg': np.array(json.dumps(test_cfg))})
        for b, (boxes, scores, labels) in enumerate(res):
            out[pre + f'boxes{b}'] = boxes.tensor.numpy()
            out[pre + f'scores{b}'] = scores.numpy()
            out[pre + f'labels{b}'] = labels.numpy()
            print('anchor head', name, b, 'kept', len(scores))
        out[pre + 'nms_source'] = np.array('oracle/ivx_oracle.c (reference CUDA op iou3d_cuda cannot run here)')
    np.savez_compressed(os.path.join(GOLD, 'anchor_head.npz'), **out)

    # full-size KITTI / nuScenes anchor grids: hashes + strided samples
    info = {}
    for name, (fm, ranges, sizes) in {
        'kitti': ((246, 214), [[0, -39.68, -1.78, 69.12 - .32, 39.68 - .32, -1.78]], [[1.6, 3.9, 1.56]]),
        'nuscenes': ((156, 156), [[-49.92, -49.92, -1.8, 49.92 - .64, 49.92 - .64, -1.8]], [[1.95, 4.60, 1.73]]),
    }.items():
        gen = ns.anchor.Anchor3DRangeGenerator(ranges=ranges, sizes=sizes, rotations=[0, 1.57], reshape_out=True)
        a = gen.grid_anchors([fm], device='cpu')[0].numpy()
        info[name] = dict(featmap=list(fm), ranges=ranges, sizes=sizes, rotations=[0, 1.57], shape=list(a.shape),
                          sha256=sha(a), first=a[0].tolist(), last=a[-1].tolist(),
                          strided=a[::9973].tolist())
    with open(os.path.join(GOLD, 'anchors_fullsize.json'), 'w') as f:
        json.dump(info, f, indent=1)


def gen_box_utils(ns):
    g = torch.Generator().manual_seed(30)
    out = {}
    val = torch.randn(64, generator=g) * 6
    out['limit_period::val'] = val.numpy()
    out['limit_period::o0.5'] = ns.utils.limit_period(val, 0.5, np.pi).numpy()
    out['limit_period::o1'] = ns.utils.limit_period(val, 1, np.pi).numpy()
    out['limit_period::o0'] = ns.utils.limit_period(val, 0, 2 * np.pi).numpy()
    b = torch.randn(32, 5, generator=g)
    out['xywhr::in'] = b.numpy()
    out['xywhr::out'] = ns.utils.xywhr2xyxyr(b).numpy()
    pts = torch.randn(16, 5, 3, generator=g)
    ang = torch.randn(16, generator=g) * 2
    out['rot::points'] = pts.numpy()
    out['rot::angles'] = ang.numpy()
    out['rot::axis2'] = ns.utils.rotation_3d_in_axis(pts, ang, axis=2).numpy()
    anchors = torch.rand(40, 7, generator=g) * 3 + 0.5
    deltas = torch.randn(40, 7, generator=g) * 0.3
    out['coder::anchors'] = anchors.numpy()
    out['coder::deltas'] = deltas.numpy()
    out['coder::decoded'] = ns.coder.DeltaXYZWLHRBBoxCoder.decode(anchors, deltas).numpy()
    # box ctor origin shift (base_box3d.py:63-66) + gravity centre
    t = torch.rand(12, 7, generator=g) * 4
    out['boxes::in'] = t.numpy()
    out['boxes::depth_origin_555'] = ns.depth.DepthInstance3DBoxes(t, origin=(.5, .5, .5)).tensor.numpy()
    out['boxes::lidar_bev'] = ns.lidar.LiDARInstance3DBoxes(t).bev.numpy()
    out['boxes::lidar_gravity'] = ns.lidar.LiDARInstance3DBoxes(t).gravity_center.numpy()
    np.savez_compressed(os.path.join(GOLD, 'box_utils.npz'), **out)


def gen_nms_vectors(ns):
    """Vectors held by the reference's own tests: tests/test_nms.py (aligned_3d_nms,
    captured by running the test with the function wrapped) and
    tests/test_box3d.py::test_boxes3d_overlaps (literals read with ast; CUDA-only test)."""
    out = {}
    cap = {}
    real = ns.nms.aligned_3d_nms

    def spy(boxes, scores, classes, thresh):
        r = real(boxes, scores, classes, thresh)
        cap.update(boxes=boxes.numpy(), scores=scores.numpy(), classes=classes.numpy(), thresh=thresh, pick=r.numpy())
        return r

    sys.modules['mmdet3d.core.post_processing'].aligned_3d_nms = spy
    src = open(os.path.join(ref_import.REF, 'tests/test_nms.py')).read()
    scope = {}
    exec(compile(src, 'test_nms.py', 'exec'), scope)
    scope['test_aligned_3d_nms']()      # asserts pick == expected inside
    sys.modules['mmdet3d.core.post_processing'].aligned_3d_nms = real
    for k, v in cap.items():
        out['aligned::' + k] = np.asarray(v)

    tree = ast.parse(open(os.path.join(ref_import.REF, 'tests/test_box3d.py')).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'test_boxes3d_overlaps'][0]
    want = {'boxes1_tensor', 'boxes2_tensor', 'expected_iou_tensor', 'expected_iof_tensor'}
    for node in ast.walk(fn):
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and node.targets[0].id in want:
            name = node.targets[0].id
            if name in out:
                continue
            lit = node.value.args[0]
            out['overlaps::' + name] = np.array(ast.literal_eval(lit), np.float32)
            if len([k for k in out if k.startswith('overlaps::')]) == 4:
                break
    # extra: random aligned-nms sets with distinct scores through the real reference
    g = torch.Generator().manual_seed(40)
    for i, n in enumerate([1, 7, 64, 300]):
        c = torch.rand(n, 3, generator=g) * 4
        s = torch.rand(n, 3, generator=g) * 1.5 + 0.1
        boxes = torch.cat([c - s / 2, c + s / 2], 1)
        scores = torch.rand(n, generator=g)
        cls = torch.randint(0, 3, (n,), generator=g)
        out[f'aligned_rand{i}::boxes'] = boxes.numpy()
        out[f'aligned_rand{i}::scores'] = scores.numpy()
        out[f'aligned_rand{i}::classes'] = cls.numpy()
        out[f'aligned_rand{i}::pick'] = real(boxes, scores, cls, 0.25).numpy()
    np.savez_compressed(os.path.join(GOLD, 'nms_vectors.npz'), **out)
    print('nms vectors', sorted(out))


def gen_e2e_small(ns):
    """Reference ImVoxelNet.simple_test end to end (orchestration pin): a toy stride-4
    backbone/FPN stand-in feeds the REAL extract_feat -> KittiImVoxelNeck -> Anchor3DHead
    -> get_bboxes -> bbox3d2result.  The toy trunk's level-0 output is stored as an input."""
    from torch import nn
    R = ns.registries

    class ToyBackbone(nn.Module):
        def __init__(self):
            super().__init__()
            self.c = nn.Conv2d(3, 8, 4, 4)

        def forward(self, x):
            y = self.c(x)
            return (y, y[..., ::2, ::2], y[..., ::4, ::4], y[..., ::8, ::8])

        def init_weights(self, pretrained=None):
            pass

    class ToyFPN(nn.Module):
        def forward(self, xs):
            return [torch.tanh(x) for x in xs]

        def init_weights(self):
            pass

    R['DET'].module_dict  # noqa
    sys.modules['mmdet.models'].BACKBONES.register_module()(ToyBackbone)
    R['NECKS'].register_module()(ToyFPN)
    ns.nms.nms_gpu = orc.nms_gpu
    ns.nms.nms_normal_gpu = orc.nms_normal_gpu

    class Cfg(dict):
        __getattr__ = dict.get

    test_cfg = dict(use_rotate_nms=True, nms_across_levels=False, nms_thr=0.01, score_thr=0.1, min_bbox_size=0,
                    nms_pre=100, max_num=50)
    n_voxels, vs = (20, 24, 12), (.32, .32, .32)
    # anchor ranges follow the kitti config rule: grid extent shrunk by one voxel (imvoxelnet_kitti.py:30)
    ox, oy = 3.2 + 0.5, 0.0
    ranges = [[ox - 3.2, oy - 3.84, -1.78, ox + 3.2 - .32, oy + 3.84 - .32, -1.78]]
    torch.manual_seed(50)

    def super_init(self, pretrained=None):
        return None
    nn.Module.init_weights = super_init  # BaseDetector.init_weights stand-in (mmdet, logging only)
    model = ns.detector.ImVoxelNet(
        backbone=dict(type='ToyBackbone'), neck=dict(type='ToyFPN'),
        neck_3d=dict(type='KittiImVoxelNeck', in_channels=8, out_channels=16),
        bbox_head=dict(type='Anchor3DHead', num_classes=1, in_channels=16, feat_channels=16,
                       use_direction_classifier=True,
                       anchor_generator=dict(type='Anchor3DRangeGenerator', ranges=ranges, sizes=[[1.6, 3.9, 1.56]],
                                             rotations=[0, 1.57], reshape_out=True),
                       diff_rad_by_sin=True, bbox_coder=dict(type='DeltaXYZWLHRBBoxCoder'),
                       loss_cls=dict(type='FocalLoss', use_sigmoid=True)),
        n_voxels=n_voxels, voxel_size=vs, train_cfg=None, test_cfg=Cfg(test_cfg))
    del nn.Module.init_weights
    g = torch.Generator().manual_seed(51)
    randomize_bn(model, g)
    with torch.no_grad():
        model.bbox_head.conv_cls.weight.normal_(0, 0.3)
        model.bbox_head.conv_cls.bias.fill_(-0.5)
        model.bbox_head.conv_reg.weight.normal_(0, 0.05)
        model.bbox_head.conv_dir_cls.weight.normal_(0, 0.3)
    model.eval()
    B, H, W = 2, 96, 160
    img = torch.randn(B, 1, 3, H, W, generator=g)
    K = np.array([[36., 0, 40, 0], [0, 36., 22, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    metas = []
    for b in range(B):
        E = np.array([[0, -1, 0, 0.03 * b], [0, 0, -1, 0.2], [1, 0, 0, 0.1], [0, 0, 0, 1]], np.float32)
        metas.append(dict(img_shape=(H - 8 * b, W - 12 * b, 3), ori_shape=((H - 8 * b) // 2, (W - 12 * b) // 2, 3),
                          box_type_3d=ns.lidar.LiDARInstance3DBoxes,
                          lidar2img=dict(intrinsic=K, extrinsic=[E], origin=np.array([ox, oy, -1.0], np.float32))))
    with torch.no_grad():
        x, valids, _ = model.extract_feat(img, metas, 'test')
        res = model.simple_test(img, metas)
        f = model.backbone(img.reshape(-1, 3, H, W))
        fpn0 = model.neck(f)[0]
    out = {}
    out.update(sd_to_np(model.state_dict(), 'sd::'))
    out.update({'fpn0': fpn0.numpy(), 'neck_out': x[0].numpy(), 'valids': valids.numpy(),
                'n_voxels': np.array(n_voxels), 'voxel_size': np.array(vs, np.float32),
                'ranges': np.array(ranges), 'test_cfg': np.array(json.dumps(test_cfg)),
                'nms_source': np.array('oracle/ivx_oracle.c')})
    for b in range(B):
        m = metas[b]
        out[f'meta{b}::img_shape'] = np.array(m['img_shape'])
        out[f'meta{b}::ori_shape'] = np.array(m['ori_shape'])
        out[f'meta{b}::intrinsic'] = m['lidar2img']['intrinsic']
        out[f'meta{b}::extrinsic'] = np.stack(m['lidar2img']['extrinsic'])
        out[f'meta{b}::origin'] = m['lidar2img']['origin']
        out[f'res{b}::boxes'] = res[b]['boxes_3d'].tensor.numpy()
        out[f'res{b}::scores'] = res[b]['scores_3d'].numpy()
        out[f'res{b}::labels'] = res[b]['labels_3d'].numpy()
        print('e2e sample', b, 'valid frac', float(valids[b].float().mean()), 'dets', len(res[b]['scores_3d']))
    np.savez_compressed(os.path.join(GOLD, 'e2e_small.npz'), **out)


def gen_indoor_heads(ns):
    """ImVoxelHeadV2 (ScanNet, SUN RGB-D) and V1 (ScanNet with a 1-conv tower): forward + get_bboxes through the
    real reference classes.  SUN RGB-D uses rotated NMS -> the C restatement is injected (see nms_source)."""
    class Cfg(dict):
        __getattr__ = dict.get
    ns.nms.nms_gpu = orc.nms_gpu
    ns.nms.nms_normal_gpu = orc.nms_normal_gpu
    ns.head_v2.box3d_multiclass_nms = ns.nms.box3d_multiclass_nms
    out = {}
    cases = {
        'scannet_v2': (ns.head_v2.ScanNetImVoxelHeadV2, dict(n_classes=5, n_channels=8, n_reg_outs=6, n_scales=3, limit=27, centerness_topk=18),
                       dict(nms_pre=40, iou_thr=.25, score_thr=.01), 6),
        'sunrgbd_v2': (ns.head_v2.SunRgbdImVoxelHeadV2, dict(n_classes=4, n_channels=8, n_reg_outs=7, n_scales=3, limit=27, centerness_topk=18),
                       dict(nms_pre=40, nms_thr=.15, use_rotate_nms=True, score_thr=.0), 7),
        'scannet_v1': (ns.head_v1.ScanNetImVoxelHead, dict(n_classes=5, n_channels=8, n_convs=1, n_reg_outs=6),
                       dict(nms_pre=40, iou_thr=.15, score_thr=.01), 6),
    }
    for name, (cls, kw, tcfg, R) in cases.items():
        torch.manual_seed({'scannet_v2': 60, 'sunrgbd_v2': 61, 'scannet_v1': 62}[name])
        head = cls(train_cfg=None, test_cfg=Cfg(tcfg), **kw)
        head.voxel_size = (.16, .16, .16)
        g = torch.Generator().manual_seed(63)
        randomize_bn(head, g)
        with torch.no_grad():
            head.centerness_conv.weight.normal_(0, 0.2)
            head.reg_conv.weight.normal_(0, 0.05)
            head.cls_conv.weight.normal_(0, 0.2)
            head.cls_conv.bias.fill_(-0.5)
            for i, sc in enumerate(head.scales):
                sc.scale.fill_(1.0 + 0.25 * i)
        head.eval()
        B = 2
        xs = [torch.randn(B, 8, 8 >> l, 8 >> l, 4 >> l, generator=g) for l in range(3)]
        valid = (torch.rand(B, 1, 8, 8, 4, generator=g) > 0.35).float()
        box_t = ns.depth.DepthInstance3DBoxes
        metas = [dict(box_type_3d=box_t, lidar2img=dict(origin=np.array([0.1 * b, 3.0, -1.0 + 0.5 * b], np.float32))) for b in range(B)]
        with torch.no_grad():
            cs, bs, ss = head(xs)
            res = head.get_bboxes(cs, bs, ss, valid, metas)
        pre = name + '::'
        out.update(sd_to_np(head.state_dict(), pre + 'sd::'))
        out[pre + 'valid'] = valid.numpy()
        out[pre + 'test_cfg'] = np.array(json.dumps(tcfg))
        out[pre + 'kw'] = np.array(json.dumps(kw))
        for l in range(3):
            out[pre + f'x{l}'] = xs[l].numpy()
            out[pre + f'centerness{l}'] = cs[l].numpy()
            out[pre + f'bbox_pred{l}'] = bs[l].numpy()
            out[pre + f'cls{l}'] = ss[l].numpy()
        for b in range(B):
            out[pre + f'origin{b}'] = metas[b]['lidar2img']['origin']
            boxes, scores, labels = res[b]
            out[pre + f'boxes{b}'] = boxes.tensor.numpy()
            out[pre + f'scores{b}'] = scores.numpy()
            out[pre + f'labels{b}'] = labels.numpy()
            print('indoor head', name, b, 'kept', len(scores))
        out[pre + 'nms_source'] = np.array('reference aligned_3d_nms' if R == 6 else 'oracle/ivx_oracle.c rotated NMS')
    np.savez_compressed(os.path.join(GOLD, 'indoor_heads.npz'), **out)


def gen_e2e_indoor(ns):
    """Reference ImVoxelNet.simple_test end to end on the two INDOOR families (orchestration pin of the anchor-free path): the toy
    stride-4 trunk of gen_e2e_small feeds the REAL extract_feat (multi-view unprojection, detectors/imvoxelnet.py:45-80) ->
    FastIndoorImVoxelNeck -> ScanNetImVoxelHeadV2 / SunRgbdImVoxelHeadV2 forward + get_bboxes (aligned 3-D NMS / rotated multi-class
    NMS) -> bbox3d2result.  The trunk's level-0 output is stored as the input of the handle (ivx_model_cfg.with_trunk = 0).
    Written LAST by main() so the other fixtures' random streams are untouched."""
    from torch import nn

    class Cfg(dict):
        __getattr__ = dict.get
    ns.nms.nms_gpu = orc.nms_gpu
    ns.nms.nms_normal_gpu = orc.nms_normal_gpu
    ns.head_v2.box3d_multiclass_nms = ns.nms.box3d_multiclass_nms
    out = {}
    cases = {
        'scannet': ('ScanNetImVoxelHeadV2', dict(n_classes=5, n_channels=16, n_reg_outs=6, n_scales=3, limit=27, centerness_topk=18),
                    dict(nms_pre=50, iou_thr=.25, score_thr=.01), 3, (0.0, 0.0, 0.5)),
        'sunrgbd': ('SunRgbdImVoxelHeadV2', dict(n_classes=4, n_channels=16, n_reg_outs=7, n_scales=3, limit=27, centerness_topk=18),
                    dict(nms_pre=50, nms_thr=.15, use_rotate_nms=True, score_thr=.0), 1, (0.0, 3.0, -1.0)),
    }
    n_voxels, vs = (16, 16, 8), (.16, .16, .16)
    for name, (head_type, hkw, tcfg, V, origin) in cases.items():
        torch.manual_seed({'scannet': 70, 'sunrgbd': 71}[name])

        def super_init(self, pretrained=None):
            return None
        nn.Module.init_weights = super_init            # BaseDetector.init_weights stand-in (mmdet, logging only)
        model = ns.detector.ImVoxelNet(
            backbone=dict(type='ToyBackbone'), neck=dict(type='ToyFPN'),
            neck_3d=dict(type='FastIndoorImVoxelNeck', in_channels=8, out_channels=16, n_blocks=[1, 1, 1]),
            bbox_head=dict(type=head_type, **hkw), n_voxels=n_voxels, voxel_size=vs, train_cfg=None, test_cfg=Cfg(tcfg))
        del nn.Module.init_weights
        g = torch.Generator().manual_seed(72)
        randomize_bn(model, g)
        with torch.no_grad():
            model.bbox_head.centerness_conv.weight.normal_(0, 0.1)
            model.bbox_head.reg_conv.weight.normal_(0, 0.02)
            model.bbox_head.cls_conv.weight.normal_(0, 0.1)
            model.bbox_head.cls_conv.bias.fill_(-1.0)
            for i, sc in enumerate(model.bbox_head.scales):
                sc.scale.fill_(1.0 + 0.25 * i)
        model.eval()
        B, H, W = 2, 96, 128
        img = torch.randn(B, V, 3, H, W, generator=g)
        K = np.array([[70., 0, 63.5, 0], [0, 70., 47.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
        metas = []
        for b in range(B):
            E = [look_at((2.0 * np.cos(2 * np.pi * (i + 0.3 * b) / max(V, 3)) + origin[0], 2.0 * np.sin(2 * np.pi * (i + 0.3 * b) / max(V, 3)) + origin[1],
                          origin[2] + 0.9), origin) for i in range(V)]
            metas.append(dict(img_shape=(H - 8 * b, W - 12 * b, 3), ori_shape=((H - 8 * b) // 2, (W - 12 * b) // 2, 3),
                              box_type_3d=ns.depth.DepthInstance3DBoxes,
                              lidar2img=dict(intrinsic=K, extrinsic=E, origin=np.array(origin, np.float32))))
        with torch.no_grad():
            x, valids, _ = model.extract_feat(img, metas, 'test')
            res = model.simple_test(img, metas)
            f = model.backbone(img.reshape(-1, 3, H, W))
            fpn0 = model.neck(f)[0]
        pre = name + '::'
        out.update(sd_to_np(model.state_dict(), pre + 'sd::'))
        out.update({pre + 'fpn0': fpn0.numpy(), pre + 'valids': valids.numpy(), pre + 'n_voxels': np.array(n_voxels),
                    pre + 'voxel_size': np.array(vs, np.float32), pre + 'test_cfg': np.array(json.dumps(tcfg)),
                    pre + 'head_kw': np.array(json.dumps(hkw)), pre + 'views': np.array(V), pre + 'hw': np.array([H, W]),
                    pre + 'nms_source': np.array('reference aligned_3d_nms' if hkw['n_reg_outs'] == 6 else 'oracle/ivx_oracle.c rotated NMS')})
        for l in range(3):
            out[pre + f'level{l}'] = x[l].numpy()
        for b in range(B):
            m = metas[b]
            out[pre + f'meta{b}::img_shape'] = np.array(m['img_shape'])
            out[pre + f'meta{b}::ori_shape'] = np.array(m['ori_shape'])
            out[pre + f'meta{b}::intrinsic'] = m['lidar2img']['intrinsic']
            out[pre + f'meta{b}::extrinsic'] = np.stack(m['lidar2img']['extrinsic'])
            out[pre + f'meta{b}::origin'] = m['lidar2img']['origin']
            out[pre + f'res{b}::boxes'] = res[b]['boxes_3d'].tensor.numpy()
            out[pre + f'res{b}::scores'] = res[b]['scores_3d'].numpy()
            out[pre + f'res{b}::labels'] = res[b]['labels_3d'].numpy()
            print('e2e indoor', name, 'sample', b, 'valid frac', float(valids[b].float().mean()), 'dets', len(res[b]['scores_3d']))
    np.savez_compressed(os.path.join(GOLD, 'e2e_indoor.npz'), **out)


def gen_indoor_eval(ns):
    """Reference indoor_eval (core/evaluation/indoor_eval.py) on synthetic scenes.  Its 3-D IoU goes through
    BaseInstance3DBoxes.overlaps -> iou3d_cuda.boxes_overlap_bev_gpu (CUDA): the C restatement is injected and
    Tensor.cuda() is made a no-op for the duration so the reference's own overlaps() code runs on the CPU."""
    import types
    from oracle import c_oracle as co
    import importlib.util
    sys.modules['mmcv.utils'] = types.ModuleType('mmcv.utils')
    sys.modules['mmcv.utils'].print_log = lambda *a, **k: None
    tt = types.ModuleType('terminaltables')

    class AsciiTable:
        def __init__(self, data):
            self.table = ''
    tt.AsciiTable = AsciiTable
    sys.modules['terminaltables'] = tt
    spec = importlib.util.spec_from_file_location('ref_indoor_eval', os.path.join(ref_import.REF, 'mmdet3d/core/evaluation/indoor_eval.py'))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)

    class FakeIou3d:
        @staticmethod
        def boxes_overlap_bev_gpu(a, b, out):
            out.copy_(torch.from_numpy(co.boxes_overlap_bev(a.numpy(), b.numpy())))
    ns.base.iou3d_cuda = FakeIou3d
    Depth = ns.depth.DepthInstance3DBoxes
    Depth.convert_to = lambda self, dst, rt_mat=None: self          # boxes are already in depth mode
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        rng = np.random.RandomState(80)
        n_scenes, n_cls = 6, 4
        gt_annos, dt_annos, store = [], [], {}
        for sidx in range(n_scenes):
            n = rng.randint(0, 7) if sidx != 2 else 0
            ctr = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
            size = rng.uniform(0.4, 1.6, (n, 3)).astype(np.float32)
            yaw = rng.uniform(-1.5, 1.5, (n, 1)).astype(np.float32)
            gtb = np.concatenate([ctr, size, yaw], 1)
            cls = rng.randint(0, n_cls, (n,)).astype(np.int64)
            gt_annos.append(dict(gt_num=n, gt_boxes_upright_depth=gtb, **{'class': cls}))
            # detections: jittered copies of most gts (+ duplicates) and some false positives
            keep = rng.rand(n) < 0.8
            det = gtb[keep] + rng.normal(0, 0.08, (keep.sum(), 7)).astype(np.float32)
            dcls = cls[keep].copy()
            flip = rng.rand(len(dcls)) < 0.15
            dcls[flip] = rng.randint(0, n_cls, flip.sum())
            nfp = rng.randint(1, 5)
            fpb = np.concatenate([rng.uniform(-3, 3, (nfp, 3)), rng.uniform(0.4, 1.6, (nfp, 3)), rng.uniform(-1.5, 1.5, (nfp, 1))], 1).astype(np.float32)
            det = np.concatenate([det, det[:1], fpb]) if len(det) else fpb
            dcls = np.concatenate([dcls, dcls[:1], rng.randint(0, n_cls, nfp)]) if len(dcls) else rng.randint(0, n_cls, nfp)
            scores = rng.uniform(0.05, 1.0, len(det)).astype(np.float32)
            det_bc = det.copy()
            det_bc[:, 2] -= det_bc[:, 5] * 0.5            # detections are stored bottom-centred (as simple_test returns them)
            dt_annos.append(dict(boxes_3d=Depth(torch.from_numpy(det_bc)), scores_3d=torch.from_numpy(scores),
                                 labels_3d=torch.from_numpy(dcls.astype(np.int64))))
            store[f's{sidx}::gt_boxes'] = gtb
            store[f's{sidx}::gt_class'] = cls
            store[f's{sidx}::det_boxes'] = det_bc
            store[f's{sidx}::det_scores'] = scores
            store[f's{sidx}::det_labels'] = dcls.astype(np.int64)
        label2cat = {i: f'c{i}' for i in range(n_cls)}
        res = ev.indoor_eval(gt_annos, dt_annos, [0.25, 0.5], label2cat, box_type_3d=Depth, box_mode_3d=2)
    finally:
        torch.Tensor.cuda = real_cuda
    store['result'] = np.array(json.dumps(res))
    store['n_scenes'] = np.array(n_scenes)
    np.savez_compressed(os.path.join(GOLD, 'indoor_eval.npz'), **store)
    print('indoor_eval', {k: round(v, 4) for k, v in res.items() if k.startswith('m')})


def synth_kitti_annos(seed=90, n_img=60):
    """Synthetic KITTI-format annotation / detection dicts (camera frame) that exercise every branch of the AP code:
    all three classes plus Van / Person_sitting / DontCare, heights around the per-difficulty minimum, occlusion and
    truncation levels around the limits, missed objects, duplicates, class flips and false positives."""
    rng = np.random.RandomState(seed)
    names = ['Car', 'Pedestrian', 'Cyclist', 'Van', 'Person_sitting', 'DontCare']
    dims_by = {'Car': (3.9, 1.56, 1.6), 'Van': (5.0, 2.1, 1.9), 'Pedestrian': (0.8, 1.73, 0.6), 'Person_sitting': (0.8, 1.2, 0.6),
               'Cyclist': (1.76, 1.73, 0.6), 'DontCare': (1.0, 1.0, 1.0)}
    gts, dts = [], []
    for i in range(n_img):
        n = rng.randint(1, 9)
        nm = rng.choice(names, n, p=[0.4, 0.2, 0.15, 0.1, 0.05, 0.1])
        z = rng.uniform(6, 55, n)
        x = rng.uniform(-12, 12, n)
        y = rng.uniform(1.4, 1.9, n)
        dims = np.array([dims_by[k] for k in nm]) * rng.uniform(0.85, 1.15, (n, 3))
        ry = rng.uniform(-np.pi, np.pi, n)
        h2d = 721.5 * dims[:, 1] / z * rng.uniform(0.9, 1.1, n)           # pixel height: crosses 25 / 40 px with depth
        w2d = h2d * rng.uniform(0.6, 2.2, n)
        cx, cy = 609.5 + 721.5 * x / z, 172.8 + rng.uniform(-10, 30, n)
        bbox = np.stack([cx - w2d / 2, cy - h2d / 2, cx + w2d / 2, cy + h2d / 2], 1)
        gt = dict(name=np.array(nm), truncated=rng.choice([0.0, 0.1, 0.2, 0.4, 0.6], n, p=[.5, .2, .1, .1, .1]),
                  occluded=rng.choice([0, 1, 2, 3], n, p=[.5, .25, .15, .1]), alpha=rng.uniform(-np.pi, np.pi, n),
                  bbox=bbox, dimensions=dims, location=np.stack([x, y, z], 1), rotation_y=ry)
        real = nm != 'DontCare'
        keep = real & (rng.rand(n) < 0.8)
        k = int(keep.sum())
        jit = rng.normal(0, 1, (k, 3)) * np.array([[0.08, 0.02, 0.12]]) * rng.choice([1.0, 3.0], (k, 1), p=[0.7, 0.3])
        dloc = gt['location'][keep] + jit
        ddim = dims[keep] * rng.uniform(0.96, 1.04, (k, 3))
        dry = ry[keep] + rng.normal(0, 0.05, k)
        dbox = bbox[keep] + rng.normal(0, 2.0, (k, 4))
        dname = nm[keep].copy()
        dname[np.isin(dname, ['Van'])] = 'Car'
        dname[np.isin(dname, ['Person_sitting'])] = 'Pedestrian'
        flip = rng.rand(k) < 0.1
        dname[flip] = rng.choice(['Car', 'Pedestrian', 'Cyclist'], int(flip.sum()))
        dalpha = gt['alpha'][keep] + rng.normal(0, 0.2, k)
        # duplicates of the first kept object and a few false positives (some on DontCare regions, some tiny)
        nfp = rng.randint(0, 4)
        if k:
            dloc = np.concatenate([dloc, dloc[:1] + 0.05]); ddim = np.concatenate([ddim, ddim[:1]]); dry = np.concatenate([dry, dry[:1]])
            dbox = np.concatenate([dbox, dbox[:1] + 1.0]); dname = np.concatenate([dname, dname[:1]]); dalpha = np.concatenate([dalpha, dalpha[:1]])
        fz = rng.uniform(6, 55, nfp)
        fl = np.stack([rng.uniform(-12, 12, nfp), rng.uniform(1.4, 1.9, nfp), fz], 1)
        fd = np.array([dims_by['Car']] * nfp).reshape(nfp, 3) * rng.uniform(0.8, 1.2, (nfp, 3))
        fh = rng.uniform(15, 90, nfp)
        fcx, fcy = rng.uniform(100, 1100, nfp), rng.uniform(150, 220, nfp)
        fb = np.stack([fcx - fh, fcy - fh / 2, fcx + fh, fcy + fh / 2], 1)
        dc = bbox[nm == 'DontCare']
        if len(dc) and nfp:
            fb[0] = dc[0] + rng.normal(0, 1.0, 4)
        dloc = np.concatenate([dloc.reshape(-1, 3), fl]); ddim = np.concatenate([ddim.reshape(-1, 3), fd]); dry = np.concatenate([dry, rng.uniform(-3, 3, nfp)])
        dbox = np.concatenate([dbox.reshape(-1, 4), fb]); dname = np.concatenate([dname, rng.choice(['Car', 'Pedestrian', 'Cyclist'], nfp)])
        dalpha = np.concatenate([dalpha, rng.uniform(-3, 3, nfp)])
        nd = len(dname)
        dt = dict(name=np.array(dname), truncated=np.zeros(nd), occluded=np.zeros(nd, dtype=np.int64), alpha=dalpha, bbox=dbox,
                  dimensions=ddim, location=dloc, rotation_y=dry, score=rng.uniform(0.05, 1.0, nd))
        gts.append(gt)
        dts.append(dt)
    return gts, dts


def gen_kitti_eval(ns):
    """Reference kitti_eval (core/evaluation/kitti_utils/eval.py) on synthetic annotations.  numba is absent, so
    numba.jit is the identity and the reference's numba-CUDA *device functions* of rotate_iou.py (rbbox_to_corners,
    quadrilateral_intersection, sort_vertex_in_convex_polygon, area, inter, devRotateIoUEval) run as plain Python with
    cuda.local.array -> numpy float32 arrays; only the kernel launcher rotate_iou_gpu_eval is replaced by a pair loop
    that calls the reference's devRotateIoUEval with the launcher's argument order (query box first)."""
    import types
    import importlib.util
    nb = sys.modules['numba']
    nb.float32 = np.float32
    nb.prange = range
    cuda = types.ModuleType('numba.cuda')

    def cjit(*a, **k):
        if a and callable(a[0]):
            return a[0]
        return lambda f: f
    cuda.jit = cjit
    cuda.local = types.SimpleNamespace(array=lambda shape, dtype: np.zeros(shape, dtype=dtype))
    nb.cuda = cuda
    sys.modules['numba.cuda'] = cuda
    pkg = types.ModuleType('ref_kitti_utils')
    pkg.__path__ = [os.path.join(ref_import.REF, 'mmdet3d/core/evaluation/kitti_utils')]
    sys.modules['ref_kitti_utils'] = pkg
    mods = {}
    for name in ('rotate_iou', 'eval'):
        spec = importlib.util.spec_from_file_location(f'ref_kitti_utils.{name}', os.path.join(pkg.__path__[0], name + '.py'))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f'ref_kitti_utils.{name}'] = m
        spec.loader.exec_module(m)
        mods[name] = m
    riou, ev = mods['rotate_iou'], mods['eval']

    def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
        boxes = boxes.astype(np.float32)
        query_boxes = query_boxes.astype(np.float32)
        out = np.zeros((boxes.shape[0], query_boxes.shape[0]), dtype=np.float32)
        for i in range(boxes.shape[0]):
            for j in range(query_boxes.shape[0]):
                out[i, j] = riou.devRotateIoUEval(query_boxes[j], boxes[i], criterion)
        return out
    riou.rotate_iou_gpu_eval = rotate_iou_gpu_eval

    store = {}
    # (1) rotated IoU vectors: random pairs + touching / identical / contained / disjoint cases
    rng = np.random.RandomState(91)
    a = np.concatenate([rng.uniform(-4, 4, (40, 2)), rng.uniform(0.5, 5, (40, 2)), rng.uniform(-3.2, 3.2, (40, 1))], 1).astype(np.float32)
    b = np.concatenate([a[:20, :2] + rng.normal(0, 0.8, (20, 2)), a[:20, 2:4] * rng.uniform(0.7, 1.3, (20, 2)),
                        a[:20, 4:] + rng.normal(0, 0.5, (20, 1))], 1).astype(np.float32)
    b = np.concatenate([b, a[:3], np.array([[0, 0, 2, 2, 0], [0.5, 0, 1, 1, 0.3], [50, 50, 1, 1, 0]], dtype=np.float32)])
    store['riou::boxes'], store['riou::query'] = a, b
    for crit in (-1, 0, 1, 2):
        store[f'riou::out{crit}'] = rotate_iou_gpu_eval(a, b, crit)
    # (2) whole evaluation
    gts, dts = synth_kitti_annos()
    for tag, annos in (('gt', gts), ('dt', dts)):       # the inputs travel with the fixture (flat arrays + per-image counts)
        store[f'anno::{tag}::count'] = np.array([len(a['name']) for a in annos], dtype=np.int64)
        for key in annos[0]:
            store[f'anno::{tag}::{key}'] = np.concatenate([np.asarray(a[key]) for a in annos], 0)
    res_str, res = ev.kitti_eval(gts, dts, ['Car', 'Pedestrian', 'Cyclist'], eval_types=['bbox', 'bev', '3d'])
    res1_str, res1 = ev.kitti_eval(gts, dts, ['Car'], eval_types=['bbox', 'bev', '3d'])
    store['eval::result_str'] = np.array(res_str)
    store['eval::result'] = np.array(json.dumps({k: float(v) for k, v in res.items()}))
    store['eval::car_only_str'] = np.array(res1_str)
    store['eval::car_only'] = np.array(json.dumps({k: float(v) for k, v in res1.items()}))
    # per-image 3-D / BEV overlap matrices of the first images (dt x gt), for the overlap restatement
    ov3, _, _, _ = ev.calculate_iou_partly(dts[:6], gts[:6], 2, 6)
    ovb, _, _, _ = ev.calculate_iou_partly(dts[:6], gts[:6], 1, 6)
    for i in range(6):
        store[f'eval::ov3d{i}'] = ov3[i]
        store[f'eval::ovbev{i}'] = ovb[i]
    np.savez_compressed(os.path.join(GOLD, 'kitti_eval.npz'), **store)
    print('kitti_eval', res_str)


def gen_kitti_format(ns):
    """Reference KittiDataset.bbox2result_kitti / convert_valid_bboxes (datasets/kitti_dataset.py:360-472,587-674) on
    synthetic LiDAR-frame detections.  The dataset module is loaded with stand-ins for mmcv / mmdet / Custom3DDataset
    (none of which the two methods touch besides mmcv.track_iter_progress); the box classes are the reference's."""
    import types
    import importlib.util
    base = 'mmdet3d/core/bbox/structures/'
    cam = ref_import._load('mmdet3d.core.bbox.structures.cam_box3d', base + 'cam_box3d.py')
    mode = ref_import._load('mmdet3d.core.bbox.structures.box_3d_mode', base + 'box_3d_mode.py')
    mmcv = sys.modules['mmcv']
    mmcv.track_iter_progress = lambda x: x
    mmcv.mkdir_or_exist = lambda p: os.makedirs(p, exist_ok=True)
    mu = types.ModuleType('mmcv.utils')
    mu.print_log = lambda *a, **k: None
    sys.modules['mmcv.utils'] = mu

    class _Reg:
        def register_module(self, *a, **k):
            return lambda c: c
    md = types.ModuleType('mmdet.datasets')
    md.DATASETS = _Reg()
    sys.modules['mmdet.datasets'] = md
    core = sys.modules['mmdet3d.core']
    core.show_result = None
    cb = sys.modules['mmdet3d.core.bbox']
    cb.Box3DMode, cb.CameraInstance3DBoxes, cb.Coord3DMode = mode.Box3DMode, cam.CameraInstance3DBoxes, None   # Coord3DMode: show() only
    cb.points_cam2img = ns.utils.points_cam2img
    cb.LiDARInstance3DBoxes = ns.lidar.LiDARInstance3DBoxes
    pk = types.ModuleType('mmdet3d.datasets')
    pk.__path__ = [os.path.join(ref_import.REF, 'mmdet3d/datasets')]
    sys.modules['mmdet3d.datasets'] = pk
    c3 = types.ModuleType('mmdet3d.datasets.custom_3d')
    c3.Custom3DDataset = object
    sys.modules['mmdet3d.datasets.custom_3d'] = c3
    spec = importlib.util.spec_from_file_location('mmdet3d.datasets.kitti_dataset', os.path.join(ref_import.REF, 'mmdet3d/datasets/kitti_dataset.py'))
    kd = importlib.util.module_from_spec(spec)
    sys.modules['mmdet3d.datasets.kitti_dataset'] = kd
    spec.loader.exec_module(kd)
    # LiDARInstance3DBoxes.convert_to needs Box3DMode from its own module namespace
    ns.lidar.LiDARInstance3DBoxes.convert_to = lambda self, dst, rt_mat=None: mode.Box3DMode.convert(self, mode.Box3DMode.LIDAR, dst, rt_mat)

    rng = np.random.RandomState(95)
    rect = np.eye(4); rect[:3, :3] = [[0.9999, 0.0098, -0.0074], [-0.0099, 0.9999, -0.0043], [0.0074, 0.0044, 0.9999]]
    trv2c = np.eye(4); trv2c[:3] = [[0.0075, -0.99997, -0.0006, -0.0041], [0.0148, 0.0007, -0.9999, -0.0763], [0.9999, 0.0075, 0.0148, -0.2718]]
    p2 = np.eye(4); p2[:3] = [[721.5377, 0, 609.5593, 44.857], [0, 721.5377, 172.854, 0.2163], [0, 0, 1, 0.0027]]
    infos, outs, store = [], [], {}
    for i in range(5):
        n = [10, 0, 16, 4, 24][i]
        xyz = np.stack([rng.uniform(-5, 80, n), rng.uniform(-45, 45, n), rng.uniform(-3.3, 0.3, n)], 1)
        wlh = np.stack([rng.uniform(0.5, 2, n), rng.uniform(0.6, 4.5, n), rng.uniform(1.2, 2, n)], 1)
        yaw = rng.uniform(-4, 4, (n, 1))
        b = np.concatenate([xyz, wlh, yaw], 1).astype(np.float32)
        sc = rng.uniform(0.1, 1, n).astype(np.float32)
        lb = rng.randint(0, 3, n).astype(np.int64)
        info = dict(image=dict(image_idx=100 + i, image_shape=np.array([375, 1242], dtype=np.int32)),
                    calib=dict(R0_rect=rect, Tr_velo_to_cam=trv2c, P2=p2))
        infos.append(info)
        outs.append(dict(boxes_3d=ns.lidar.LiDARInstance3DBoxes(torch.from_numpy(b.copy())), scores_3d=torch.from_numpy(sc), labels_3d=torch.from_numpy(lb)))
        store[f'in{i}::boxes'], store[f'in{i}::scores'], store[f'in{i}::labels'] = b, sc, lb
    fake = types.SimpleNamespace(data_infos=infos, pcd_limit_range=[0, -40, -3, 70.4, 40, 0.0])
    fake.convert_valid_bboxes = lambda box_dict, info: kd.KittiDataset.convert_valid_bboxes(fake, box_dict, info)
    annos = kd.KittiDataset.bbox2result_kitti(fake, outs, ['Pedestrian', 'Cyclist', 'Car'])
    for i, a in enumerate(annos):
        for k, v in a.items():
            store[f'out{i}::{k}'] = np.asarray(v)
    store['calib::R0_rect'], store['calib::Tr_velo_to_cam'], store['calib::P2'] = rect, trv2c, p2
    np.savez_compressed(os.path.join(GOLD, 'kitti_format.npz'), **store)
    print('kitti_format kept', [len(a['score']) for a in annos])


def gen_layout_head(ns):
    """Reference LayoutHead (dense_heads/layout_head.py) forward + get_extrinsics / _compute_projection with predicted
    angles (detectors/imvoxelnet.py:114-129,164-187) on seeded weights and a random C5 map."""
    sys.modules['mmdet3d.core.bbox.structures'].limit_period = ns.utils.limit_period
    lh = ref_import._load('mmdet3d.models.dense_heads.layout_head', 'mmdet3d/models/dense_heads/layout_head.py')
    torch.manual_seed(97)
    head = lh.LayoutHead(n_channels=256, linear_size=64, dropout=0.0).eval()
    with torch.no_grad():
        for p in head.parameters():
            p.mul_(6.0)            # wider outputs so limit_period actually wraps some angles
        x = torch.randn(3, 256, 6, 8) + 0.3
        angles, layouts = head.forward(x, [None] * 3)
    store = {'x': x.numpy()}
    for k, v in head.state_dict().items():
        store['sd::' + k] = v.numpy()
    store['angles'] = torch.stack(angles).numpy()
    store['layouts'] = torch.stack(layouts).numpy()
    store['extrinsics'] = torch.stack([ns.detector.get_extrinsics(a) for a in angles]).numpy()
    K = np.eye(4, dtype=np.float32)
    K[:3, :3] = [[529.5, 0, 365.0], [0, 529.5, 265.0], [0, 0, 1]]
    meta = dict(img_shape=(480, 640, 3), ori_shape=(530, 730, 3), lidar2img=dict(intrinsic=K, extrinsic=[np.eye(4, dtype=np.float32)]))
    store['projection'] = ns.detector.ImVoxelNet._compute_projection(meta, 4, [angles[1]]).numpy()
    store['intrinsic'] = K
    np.savez_compressed(os.path.join(GOLD, 'layout_head.npz'), **store)
    print('layout_head angles', store['angles'].round(3).tolist())


def gen_input_side(ns):
    """Input side of the path: the reference's own get_data_info bodies of the four multi-view
    datasets (datasets/{kitti,nuscenes,scannet,sunrgbd}_monocular_dataset.py) and its MultiViewPipeline / KittiSetOrigin /
    SunRgbdSetOrigin (datasets/pipelines/multi_view.py) run on synthetic calibration records.  The dataset base classes
    (mmdet / the LiDAR datasets: file loading, annotation parsing) are replaced by empty stand-ins -- the methods called
    here touch only `self.data_infos`, `self.data_root`, `self.test_mode` and numpy."""
    import types

    class _Reg:
        def register_module(self, *a, **k):
            return lambda c: c

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class NuBase:       # NuScenesDataset.get_data_info: returns the prepared per-sample record
        def get_data_info(self, index):
            return dict(self.data_infos[index])

    stub('mmdet.datasets', DATASETS=_Reg())
    stub('mmdet.datasets.builder', PIPELINES=_Reg(), DATASETS=_Reg())

    class Compose:      # mmdet Compose: apply the callables in order
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, data):
            for t in self.transforms:
                data = t(data)
            return data
    stub('mmdet.datasets.pipelines', Compose=Compose, RandomFlip=type('RandomFlip', (), {}), LoadImageFromFile=type('LoadImageFromFile', (), {}))
    stub('mmcv.utils', print_log=lambda *a, **k: None)
    stub('mmdet3d.datasets')
    sys.modules['mmdet3d.datasets'].__path__ = []
    stub('mmdet3d.datasets.kitti_dataset', KittiDataset=type('KittiDataset', (), {}))
    stub('mmdet3d.datasets.nuscenes_dataset', NuScenesDataset=NuBase)
    stub('mmdet3d.datasets.custom_3d', Custom3DDataset=type('Custom3DDataset', (), {}))
    stub('mmdet3d.datasets.scannet_dataset', ScanNetDataset=type('ScanNetDataset', (), {'CLASSES': ('cabinet',)}))
    stub('mmdet3d.datasets.sunrgbd_dataset', SUNRGBDDataset=type('SUNRGBDDataset', (), {}))
    stub('mmdet3d.datasets.dataset_wrappers', MultiViewMixin=type('MultiViewMixin', (), {}))
    sys.modules['mmdet3d.core.bbox'].DepthInstance3DBoxes = ns.depth.DepthInstance3DBoxes
    kd = ref_import._load('mmdet3d.datasets.kitti_monocular_dataset', 'mmdet3d/datasets/kitti_monocular_dataset.py')
    nd = ref_import._load('mmdet3d.datasets.nuscenes_monocular_dataset', 'mmdet3d/datasets/nuscenes_monocular_dataset.py')
    sd = ref_import._load('mmdet3d.datasets.scannet_monocular_dataset', 'mmdet3d/datasets/scannet_monocular_dataset.py')
    ud = ref_import._load('mmdet3d.datasets.sunrgbd_monocular_dataset', 'mmdet3d/datasets/sunrgbd_monocular_dataset.py')
    mv = ref_import._load('mmdet3d.datasets.pipelines.multi_view', 'mmdet3d/datasets/pipelines/multi_view.py')

    rng = np.random.RandomState(17)

    def rot(seed):
        q, _ = np.linalg.qr(np.random.RandomState(seed).randn(3, 3))
        return q

    def self_of(cls, infos):
        o = cls.__new__(cls)
        o.data_infos, o.data_root, o.test_mode = infos, '/data', True
        return o

    out = {}
    # KITTI: P2 with a baseline term, rectification, velodyne -> camera (float64 on disk, as the info pkl holds them)
    P2 = np.eye(4); P2[:3, :3] = [[721.5377, 0, 609.5593], [0, 721.5377, 172.854], [0, 0, 1]]; P2[:3, 3] = [44.85728, 0.2163791, 0.002745884]
    R0 = np.eye(4); R0[:3, :3] = rot(1) @ np.diag([1, 1, 1.0])
    Tr = np.eye(4); Tr[:3, :3] = rot(2); Tr[:3, 3] = [0.01, -0.07, -0.27]
    info = dict(image=dict(image_idx=7, image_path='training/image_2/000007.png'), calib=dict(R0_rect=R0, Tr_velo_to_cam=Tr, P2=P2))
    r = kd.KittiMultiViewDataset.get_data_info(self_of(kd.KittiMultiViewDataset, [info]), 0)
    out.update({'kitti::P2': P2, 'kitti::R0_rect': R0, 'kitti::Tr_velo_to_cam': Tr, 'kitti::extrinsic': r['lidar2img']['extrinsic'][0],
                'kitti::intrinsic': r['lidar2img']['intrinsic']})
    pcr = [0, -39.68, -3, 69.12, 39.68, 1]
    r = mv.KittiSetOrigin(pcr)(dict(lidar2img=dict(r['lidar2img'])))
    out.update({'kitti::point_cloud_range': np.array(pcr, np.float64), 'kitti::origin': r['lidar2img']['origin']})
    # nuScenes: six lidar2img matrices with K folded in
    l2i = [rng.randn(4, 4) * 50 for _ in range(6)]
    rec = dict(sample_idx='tok', img_filename=[f'cam{i}.jpg' for i in range(6)], lidar2img=l2i)
    r = nd.NuScenesMultiViewDataset.get_data_info(self_of(nd.NuScenesMultiViewDataset, [rec]), 0)
    out.update({'nuscenes::lidar2img': np.stack(l2i), 'nuscenes::extrinsic': np.stack(r['lidar2img']['extrinsic']),
                'nuscenes::intrinsic': r['lidar2img']['intrinsic']})
    # ScanNet: axis alignment, 5 camera-to-world poses, shared intrinsics
    aam = np.eye(4); aam[:3, :3] = rot(3); aam[:3, 3] = [1.2, -0.7, 0.1]
    poses = []
    for i in range(5):
        m = np.eye(4); m[:3, :3] = rot(10 + i); m[:3, 3] = rng.randn(3)
        poses.append(m)
    K = np.array([[577.87, 0, 319.5, 0], [0, 577.87, 239.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    info = dict(annos=dict(axis_align_matrix=aam), img_paths=[f'f{i}.jpg' for i in range(5)], extrinsics=poses, intrinsics=K)
    r = sd.ScanNetMultiViewDataset.get_data_info(self_of(sd.ScanNetMultiViewDataset, [info]), 0)
    out.update({'scannet::axis_align_matrix': aam, 'scannet::poses': np.stack(poses), 'scannet::K': K,
                'scannet::extrinsic': np.stack(r['lidar2img']['extrinsic']), 'scannet::intrinsic': r['lidar2img']['intrinsic'],
                'scannet::origin': r['lidar2img']['origin']})
    # SUN RGB-D: K stored column-major, Rt
    Ksun = np.array([[529.5, 0, 365.0], [0, 529.5, 265.0], [0, 0, 1.0]]).T.reshape(-1)
    Rt = rot(4)
    info = dict(image=dict(image_path='sunrgbd_trainval/image/000001.jpg'), calib=dict(K=Ksun, Rt=Rt))
    r = ud.SunRgbdMultiViewDataset.get_data_info(self_of(ud.SunRgbdMultiViewDataset, [info]), 0)
    out.update({'sunrgbd::K': Ksun, 'sunrgbd::Rt': Rt, 'sunrgbd::extrinsic': r['lidar2img']['extrinsic'][0],
                'sunrgbd::intrinsic': r['lidar2img']['intrinsic'], 'sunrgbd::origin': r['lidar2img']['origin']})
    res = dict(lidar2img=dict(intrinsic=r['lidar2img']['intrinsic'].copy(), extrinsic=[r['lidar2img']['extrinsic'][0].copy()]),
               ori_shape=(530, 730, 3))
    res = mv.SunRgbdSetOrigin()(res)
    out['sunrgbd::set_origin'] = np.asarray(res['lidar2img']['origin'])
    # MultiViewPipeline: 7 views; draw 4 (no replacement), 7, and 10 (with replacement); the transform tags each view
    def tag(res):
        i = res['img_info']['idx']
        return dict(res, img=np.full((2, 2), i, np.float32), img_shape=(10 + i, 20 + i, 3), ori_shape=(100 + i, 200, 3), pad_shape=(32, 32, 3))
    for n in (4, 7, 10):
        results = dict(img_prefix=[None] * 7, img_info=[dict(idx=i) for i in range(7)],
                       lidar2img=dict(extrinsic=[np.full((4, 4), i, np.float32) for i in range(7)], intrinsic=np.eye(4, dtype=np.float32)))
        np.random.seed(100 + n)
        r = mv.MultiViewPipeline([tag], n)(results)
        out[f'mvp{n}::ids'] = np.array([int(e[0, 0]) for e in r['lidar2img']['extrinsic']], np.int64)
        out[f'mvp{n}::img_ids'] = np.array([int(im[0, 0]) for im in r['img']], np.int64)
        out[f'mvp{n}::img_shape'] = np.array(r['img_shape'], np.int64)
        out[f'mvp{n}::ori_shape'] = np.array(r['ori_shape'], np.int64)
    np.savez_compressed(os.path.join(GOLD, 'input_side.npz'), **out)


def main():
    ns = ref_import.load()
    gen_backproject(ns)
    gen_fullsize_kitti(ns)
    gen_necks(ns)
    gen_box_utils(ns)
    gen_nms_vectors(ns)
    gen_anchor_head(ns)
    gen_e2e_small(ns)
    gen_indoor_heads(ns)
    gen_indoor_eval(ns)
    gen_kitti_eval(ns)
    gen_kitti_format(ns)
    gen_layout_head(ns)
    gen_input_side(ns)
    gen_e2e_indoor(ns)
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == '__main__':
    main()
