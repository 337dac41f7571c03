"""KITTI AP evaluation -- the metric behind the "mAP within 0.1 of reference" clause.

Mirrors mmdet3d/core/evaluation/kitti_utils/eval.py (`kitti_eval`, `kitti_eval_coco_style`, same annotation dicts,
same result string and result-dict keys) with the work placed where it belongs on this machine:
  * rotated BEV / 3-D overlaps (rotate_iou.py:340-378, a numba-CUDA kernel in the reference): the device kernel
    behind ivx_boxes_overlap_bev -- the same geometry the NMS uses -- on all boxes of a chunk of images at once;
  * the per-image matching loops the reference jit-compiles with numba (eval.py:161-338): host C++ in
    libimvoxel_hip.so (csrc/kitti_eval.cpp: ivx_kitti_collect_scores / ivx_kitti_fused_statistics);
  * the bookkeeping (difficulty filter, 41-point recall sampling, 11-point AP, report) stays in Python.
Annotations: list of dicts with 'name' [n] str, 'truncated', 'occluded', 'alpha', 'bbox' [n,4], 'dimensions' [n,3]
(l,h,w), 'location' [n,3] (camera x,y,z), 'rotation_y' [n]; detections add 'score' [n].
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

CLASS_NAMES = ('car', 'pedestrian', 'cyclist')
MIN_HEIGHT = (40, 25, 25)
MAX_OCCLUSION = (0, 1, 2)
MAX_TRUNCATION = (0.15, 0.3, 0.5)
N_SAMPLE_PTS = 41
CLASS_TO_NAME = {0: 'Car', 1: 'Pedestrian', 2: 'Cyclist', 3: 'Van', 4: 'Person_sitting'}


def _dp(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------------------------- overlaps
def image_box_overlap(boxes, query_boxes, criterion=-1):
    """Axis-aligned 2-D IoU [N,K] (eval.py:83-112); host C++."""
    b = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 4)
    q = np.ascontiguousarray(query_boxes, dtype=np.float64).reshape(-1, 4)
    out = np.zeros((b.shape[0], q.shape[0]), dtype=np.float64)
    check(_lib.lib().ivx_kitti_image_box_overlap(_dp(b), b.shape[0], _dp(q), q.shape[0], int(criterion), _dp(out)),
          'ivx_kitti_image_box_overlap')
    return out


def _device_intersection(a_xyxyr, b_xyxyr):
    """Rotated-rectangle intersection areas [N,K] on the MI355X (no CPU fallback)."""
    import torch
    from . import ops
    if not torch.cuda.is_available():
        raise RuntimeError('rotate_iou_eval needs a HIP device (pass overlap_fn= to kitti_eval to inject another backend)')
    a = torch.from_numpy(np.ascontiguousarray(a_xyxyr, dtype=np.float32)).cuda()
    b = torch.from_numpy(np.ascontiguousarray(b_xyxyr, dtype=np.float32)).cuda()
    rows = [ops.boxes_overlap_bev(a[i:i + 32768].contiguous(), b, iou=False) for i in range(0, a.shape[0], 32768)]
    return torch.cat(rows, 0).cpu().numpy()


def rotate_iou_eval(boxes, query_boxes, criterion=-1, overlap_fn=None):
    """rotate_iou_gpu_eval (rotate_iou.py:340-378): boxes [N,5] / query [K,5] = (cx, cy, x_d, y_d, angle) -> [N,K] fp32.
    criterion -1: IoU; 0: inter / area(query); 1: inter / area(box); 2: inter  (the reference's kernel passes the query
    box as rbox1, rotate_iou.py:334-336).  The rectangle convention (x' = cos*x + sin*y, y' = -sin*x + cos*y around the
    centre, rotate_iou.py:204-227) is the one of the NMS geometry kernel, so the intersection is that kernel's."""
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 5)
    query_boxes = np.asarray(query_boxes, dtype=np.float32).reshape(-1, 5)
    n, k = boxes.shape[0], query_boxes.shape[0]
    if n == 0 or k == 0:
        return np.zeros((n, k), dtype=np.float32)

    def xyxyr(b):
        return np.stack([b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2, b[:, 4]], 1)

    inter = (overlap_fn or _device_intersection)(xyxyr(boxes), xyxyr(query_boxes)).astype(np.float32)
    area_b = (boxes[:, 2] * boxes[:, 3])[:, None]
    area_q = (query_boxes[:, 2] * query_boxes[:, 3])[None, :]
    if criterion == -1:
        return inter / (area_q + area_b - inter)
    if criterion == 0:
        return inter / area_q
    if criterion == 1:
        return inter / area_b
    return inter


def bev_box_overlap(boxes, qboxes, criterion=-1, overlap_fn=None):
    return rotate_iou_eval(boxes, qboxes, criterion, overlap_fn)


def d3_box_overlap(boxes, qboxes, criterion=-1, overlap_fn=None):
    """3-D IoU of camera-frame boxes (x, y, z, l, h, w, ry), y pointing down and (x,y,z) the bottom centre
    (eval.py:121-158): BEV intersection over (x, z, l, w, ry) times the overlap of the [y-h, y] intervals."""
    boxes = np.asarray(boxes).reshape(-1, 7)
    qboxes = np.asarray(qboxes).reshape(-1, 7)
    rinc = rotate_iou_eval(boxes[:, [0, 2, 3, 5, 6]], qboxes[:, [0, 2, 3, 5, 6]], 2, overlap_fn).astype(boxes.dtype)
    if rinc.size == 0:
        return rinc
    ih = np.minimum(boxes[:, None, 1], qboxes[None, :, 1]) - np.maximum(boxes[:, None, 1] - boxes[:, None, 4],
                                                                         qboxes[None, :, 1] - qboxes[None, :, 4])
    vol1 = (boxes[:, 3] * boxes[:, 4] * boxes[:, 5])[:, None]
    vol2 = (qboxes[:, 3] * qboxes[:, 4] * qboxes[:, 5])[None, :]
    inc = ih * rinc
    if criterion == -1:
        ua = vol1 + vol2 - inc
    elif criterion == 0:
        ua = np.broadcast_to(vol1, inc.shape)
    elif criterion == 1:
        ua = np.broadcast_to(vol2, inc.shape)
    else:
        ua = inc
    hit = (rinc > 0) & (ih > 0)
    with np.errstate(divide='ignore', invalid='ignore'):
        out = np.where(hit, inc / ua, 0.0)
    return out.astype(boxes.dtype)


# ---------------------------------------------------------------------------------------------- bookkeeping
def get_thresholds(scores, num_gt, num_sample_pts=N_SAMPLE_PTS):
    """Score thresholds at which recall crosses the 41 sample points (eval.py:7-25)."""
    scores = np.sort(np.asarray(scores, dtype=np.float64))[::-1]
    current_recall = 0.0
    thresholds = []
    n = len(scores)
    for i, score in enumerate(scores):
        l_recall = (i + 1) / num_gt
        r_recall = (i + 2) / num_gt if i < n - 1 else l_recall
        if (r_recall - current_recall) < (current_recall - l_recall) and i < n - 1:
            continue
        thresholds.append(score)
        current_recall += 1 / (num_sample_pts - 1.0)
    return thresholds


def clean_data(gt_anno, dt_anno, current_class, difficulty):
    """Difficulty / class filter (eval.py:28-80): 0 = counted, 1 = ignored (neighbouring class or too hard),
    -1 = other class.  Returns (num_valid_gt, ignored_gt, ignored_dt, dontcare boxes)."""
    cls = CLASS_NAMES[current_class]
    ignored_gt, ignored_dt, dc = [], [], []
    num_valid = 0
    for i in range(len(gt_anno['name'])):
        bbox = gt_anno['bbox'][i]
        name = gt_anno['name'][i].lower()
        height = bbox[3] - bbox[1]
        if name == cls:
            valid_class = 1
        elif (cls == 'pedestrian' and name == 'person_sitting') or (cls == 'car' and name == 'van'):
            valid_class = 0
        else:
            valid_class = -1
        ignore = (gt_anno['occluded'][i] > MAX_OCCLUSION[difficulty] or gt_anno['truncated'][i] > MAX_TRUNCATION[difficulty]
                  or height <= MIN_HEIGHT[difficulty])
        if valid_class == 1 and not ignore:
            ignored_gt.append(0)
            num_valid += 1
        elif valid_class == 0 or (ignore and valid_class == 1):
            ignored_gt.append(1)
        else:
            ignored_gt.append(-1)
        if gt_anno['name'][i] == 'DontCare':
            dc.append(gt_anno['bbox'][i])
    for i in range(len(dt_anno['name'])):
        height = abs(dt_anno['bbox'][i, 3] - dt_anno['bbox'][i, 1])
        if height < MIN_HEIGHT[difficulty]:
            ignored_dt.append(1)
        elif dt_anno['name'][i].lower() == cls:
            ignored_dt.append(0)
        else:
            ignored_dt.append(-1)
    return num_valid, ignored_gt, ignored_dt, dc


def _metric_boxes(annos, metric):
    if metric == 0:
        return np.concatenate([a['bbox'] for a in annos], 0)
    cols = [0, 2] if metric == 1 else [0, 1, 2]
    loc = np.concatenate([a['location'][:, cols] for a in annos], 0)
    dims = np.concatenate([a['dimensions'][:, cols] for a in annos], 0)
    rots = np.concatenate([a['rotation_y'] for a in annos], 0)
    return np.concatenate([loc, dims, rots[..., np.newaxis]], axis=1)


def calculate_overlaps(gt_annos, dt_annos, metric, chunk=64, overlap_fn=None):
    """Per-image overlap matrices [num_dt, num_gt] (float64, contiguous), computed chunk-of-images at a time so the
    device sees a few large launches (eval.py:341-416 does the same with `num_parts`)."""
    out = []
    for s in range(0, len(gt_annos), chunk):
        g, d = gt_annos[s:s + chunk], dt_annos[s:s + chunk]
        gb, db = _metric_boxes(g, metric), _metric_boxes(d, metric)
        if metric == 0:
            ov = image_box_overlap(db, gb)
        elif metric == 1:
            ov = bev_box_overlap(db, gb, -1, overlap_fn).astype(np.float64)
        elif metric == 2:
            ov = d3_box_overlap(db, gb, -1, overlap_fn).astype(np.float64)
        else:
            raise ValueError('unknown metric')
        gi = di = 0
        for ga, da in zip(g, d):
            ng, nd = len(ga['name']), len(da['name'])
            out.append(np.ascontiguousarray(ov[di:di + nd, gi:gi + ng]))
            gi += ng
            di += nd
    return out


def _prepare_data(gt_annos, dt_annos, current_class, difficulty):
    gt_datas, dt_datas, ign_gt, ign_dt, dcs, dc_nums = [], [], [], [], [], []
    total_valid = 0
    for g, d in zip(gt_annos, dt_annos):
        nv, ig, idt, dc = clean_data(g, d, current_class, difficulty)
        total_valid += nv
        ign_gt.append(np.array(ig, dtype=np.int64))
        ign_dt.append(np.array(idt, dtype=np.int64))
        dc = np.stack(dc, 0).astype(np.float64) if len(dc) else np.zeros((0, 4), dtype=np.float64)
        dcs.append(dc)
        dc_nums.append(dc.shape[0])
        gt_datas.append(np.concatenate([g['bbox'], g['alpha'][..., np.newaxis]], 1).astype(np.float64).reshape(-1, 5))
        dt_datas.append(np.concatenate([d['bbox'], d['alpha'][..., np.newaxis], d['score'][..., np.newaxis]], 1)
                        .astype(np.float64).reshape(-1, 6))
    return gt_datas, dt_datas, ign_gt, ign_dt, dcs, np.array(dc_nums, dtype=np.int32), total_valid


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False, overlap_fn=None):
    """precision / recall / orientation-similarity curves [class, difficulty, min_overlap, 41] (eval.py:450-568)."""
    assert len(gt_annos) == len(dt_annos)
    L = _lib.lib()
    n_img = len(gt_annos)
    overlaps = calculate_overlaps(gt_annos, dt_annos, metric, overlap_fn=overlap_fn)
    ov_ptrs = (C.c_void_p * max(n_img, 1))(*[o.ctypes.data for o in overlaps])
    gt_nums = np.array([len(a['name']) for a in gt_annos], dtype=np.int32)
    dt_nums = np.array([len(a['name']) for a in dt_annos], dtype=np.int32)
    shape = [len(current_classes), len(difficultys), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, current_class in enumerate(current_classes):
        for idx_l, difficulty in enumerate(difficultys):
            gt_datas, dt_datas, ign_gt, ign_dt, dcs, dc_nums, total_valid = _prepare_data(gt_annos, dt_annos, current_class, difficulty)
            gt_all = np.ascontiguousarray(np.concatenate(gt_datas, 0)) if n_img else np.zeros((0, 5))
            dt_all = np.ascontiguousarray(np.concatenate(dt_datas, 0)) if n_img else np.zeros((0, 6))
            ig_all = np.ascontiguousarray(np.concatenate(ign_gt, 0)) if n_img else np.zeros((0,), np.int64)
            id_all = np.ascontiguousarray(np.concatenate(ign_dt, 0)) if n_img else np.zeros((0,), np.int64)
            dc_all = np.ascontiguousarray(np.concatenate(dcs, 0)) if n_img else np.zeros((0, 4))
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                scores = np.zeros((max(int(gt_nums.sum()), 1),), dtype=np.float64)
                n_sc = C.c_int64(0)
                check(L.ivx_kitti_collect_scores(ov_ptrs, n_img, _dp(gt_nums), _dp(dt_nums), _dp(gt_all), _dp(dt_all), _dp(ig_all),
                                                 _dp(id_all), int(metric), C.c_double(float(min_overlap)), _dp(scores), C.byref(n_sc)),
                      'ivx_kitti_collect_scores')
                thresholds = np.array(get_thresholds(scores[:n_sc.value], total_valid), dtype=np.float64)
                pr = np.zeros([len(thresholds), 4], dtype=np.float64)
                check(L.ivx_kitti_fused_statistics(ov_ptrs, n_img, _dp(gt_nums), _dp(dt_nums), _dp(dc_nums), _dp(gt_all), _dp(dt_all),
                                                   _dp(dc_all), _dp(ig_all), _dp(id_all), int(metric), C.c_double(float(min_overlap)),
                                                   _dp(thresholds), len(thresholds), int(bool(compute_aos)), _dp(pr)),
                      'ivx_kitti_fused_statistics')
                nt = len(thresholds)
                with np.errstate(divide='ignore', invalid='ignore'):
                    recall[m, idx_l, k, :nt] = pr[:, 0] / (pr[:, 0] + pr[:, 2])
                    precision[m, idx_l, k, :nt] = pr[:, 0] / (pr[:, 0] + pr[:, 1])
                    if compute_aos:
                        aos[m, idx_l, k, :nt] = pr[:, 3] / (pr[:, 0] + pr[:, 1])
                for i in range(nt):     # monotone envelope over the remaining points (incl. the zero tail)
                    precision[m, idx_l, k, i] = np.max(precision[m, idx_l, k, i:], axis=-1)
                    recall[m, idx_l, k, i] = np.max(recall[m, idx_l, k, i:], axis=-1)
                    if compute_aos:
                        aos[m, idx_l, k, i] = np.max(aos[m, idx_l, k, i:], axis=-1)
    return {'recall': recall, 'precision': precision, 'orientation': aos}


def get_mAP(prec):
    """11-point interpolated AP in percent: every 4th of the 41 recall samples (eval.py:571-575)."""
    sums = 0
    for i in range(0, prec.shape[-1], 4):
        sums = sums + prec[..., i]
    return sums / 11 * 100


# The three overlap metrics of the KITTI protocol: (eval_types name, metric index of eval_class, key infix of the result dict, report label)
_METRICS = (('bbox', 0, '2D', 'bbox'), ('bev', 1, 'BEV', 'bev '), ('3d', 2, '3D', '3d  '))
_LEVELS = ('easy', 'moderate', 'hard')


def ap_tables(gt_annos, dt_annos, class_ids, min_overlaps, eval_types=('bbox', 'bev', '3d'), overlap_fn=None):
    """11-point AP tables [class, difficulty, overlap set] per requested metric: {'bbox' | 'bev' | '3d' | 'aos': array}.
    One eval_class pass per metric; 'aos' rides on the bbox pass (orientation similarity needs the 2-D matches)."""
    want = set(eval_types)
    tables = {}
    for name, metric, _, _ in _METRICS:
        if name not in want:
            continue
        with_aos = name == 'bbox' and 'aos' in want
        curves = eval_class(gt_annos, dt_annos, class_ids, [0, 1, 2], metric, min_overlaps, compute_aos=with_aos, overlap_fn=overlap_fn)
        tables[name] = get_mAP(curves['precision'])
        if with_aos:
            tables['aos'] = get_mAP(curves['orientation'])
    return tables


def do_eval(gt_annos, dt_annos, current_classes, min_overlaps, eval_types=('bbox', 'bev', '3d'), overlap_fn=None):
    """(mAP_bbox, mAP_bev, mAP_3d, mAP_aos) in the reference's order (eval.py:578-625); None for a metric that was not asked for."""
    t = ap_tables(gt_annos, dt_annos, current_classes, min_overlaps, eval_types, overlap_fn)
    return t.get('bbox'), t.get('bev'), t.get('3d'), t.get('aos')


def _class_ids(current_classes):
    name_to_class = {v: n for n, v in CLASS_TO_NAME.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    return [name_to_class[c] if isinstance(c, str) else c for c in current_classes]


def _ap_lines(tables, pick, with_aos):
    """The 'bbox AP:..' / 'bev  AP:..' / '3d   AP:..' [/ 'aos  AP:..'] lines of one report block; pick(table) -> the three difficulty values."""
    lines = ['{} AP:{:.4f}, {:.4f}, {:.4f}\n'.format(label, *pick(tables[name])) for name, _, _, label in _METRICS if name in tables]
    if with_aos:
        lines.append('aos  AP:{:.2f}, {:.2f}, {:.2f}\n'.format(*pick(tables['aos'])))
    return ''.join(lines)


def kitti_eval(gt_annos, dt_annos, current_classes, eval_types=('bbox', 'bev', '3d'), overlap_fn=None):
    """-> (report string, dict) with the reference's format and keys (eval.py:643-773): per class one block per overlap set (strict:
    0.7 / 0.5 / 0.5 per Car / Pedestrian / Cyclist, loose: 0.5 / 0.25 / 0.25 in BEV and 3-D), then the class mean of the strict set.
    overlap_fn(a_xyxyr, b_xyxyr) -> intersection areas [N,K] replaces the device kernel (used by the CPU-only tests)."""
    eval_types = list(eval_types)
    if not eval_types:
        raise AssertionError('must contain at least one evaluation type')
    if 'aos' in eval_types and 'bbox' not in eval_types:
        raise AssertionError('must evaluate bbox when evaluating aos')
    per_class = np.array([0.7, 0.5, 0.5, 0.7, 0.5])                       # strict 3-D / BEV / 2-D overlap per class id
    loose = np.array([0.5, 0.25, 0.25, 0.5, 0.25])
    class_ids = _class_ids(current_classes)
    min_overlaps = np.stack([np.stack([per_class] * 3), np.stack([per_class, loose, loose])])[:, :, class_ids]    # [set, metric, class]
    # orientation similarity is reported whenever the detections carry alpha and the ground truth has it (alpha = -10 marks "none")
    with_aos = any(a['alpha'].shape[0] != 0 for a in dt_annos) and any(a['alpha'][0] != -10 for a in gt_annos)
    if with_aos and 'aos' not in eval_types:
        eval_types.append('aos')
    tables = ap_tables(gt_annos, dt_annos, class_ids, min_overlaps, eval_types, overlap_fn)
    report, values = [], {}
    for j, cid in enumerate(class_ids):
        cname = CLASS_TO_NAME[cid]
        for s_, tag in enumerate(('strict', 'loose')):
            report.append('{} AP@{:.2f}, {:.2f}, {:.2f}:\n'.format(cname, *min_overlaps[s_, :, j]))
            report.append(_ap_lines(tables, lambda t: t[j, :, s_], with_aos))
            for d, level in enumerate(_LEVELS):
                for name, _, infix, _ in reversed(_METRICS):                 # key order of the reference: 3D, BEV, 2D
                    if name in tables:
                        values[f'KITTI/{cname}_{infix}_{level}_{tag}'] = tables[name][j, d, s_]
    if len(class_ids) > 1:
        mean = {k: v.mean(axis=0) for k, v in tables.items()}
        report.append('\nOverall AP@{}, {}, {}:\n'.format(*_LEVELS))
        report.append(_ap_lines(mean, lambda t: t[:, 0], with_aos))
        for d, level in enumerate(_LEVELS):
            for name, _, infix, _ in reversed(_METRICS):
                if name in mean:
                    values[f'KITTI/Overall_{infix}_{level}'] = mean[name][d, 0]
    return ''.join(report), values


def kitti_eval_coco_style(gt_annos, dt_annos, current_classes, overlap_fn=None):
    """COCO-style AP over 10 overlap thresholds per class (eval.py:776-845) -> report string."""
    class_to_range = {0: [0.5, 0.95, 10], 1: [0.25, 0.7, 10], 2: [0.25, 0.7, 10], 3: [0.5, 0.95, 10], 4: [0.25, 0.7, 10]}
    current_classes = _class_ids(current_classes)
    overlap_ranges = np.zeros([3, 3, len(current_classes)])
    for i, c in enumerate(current_classes):
        overlap_ranges[:, :, i] = np.array(class_to_range[c])[:, np.newaxis]
    compute_aos = False
    for anno in dt_annos:
        if anno['alpha'].shape[0] != 0:
            compute_aos = anno['alpha'][0] != -10
            break
    min_overlaps = np.zeros([10, *overlap_ranges.shape[1:]])
    for i in range(overlap_ranges.shape[1]):
        for j in range(overlap_ranges.shape[2]):
            lo, hi, num = overlap_ranges[:, i, j]
            min_overlaps[:, i, j] = np.linspace(lo, hi, int(num))
    # (the reference hands its compute_aos bool to do_eval's eval_types parameter, eval.py:633-635, which raises a
    # TypeError there -- the function is dead code upstream; here it evaluates every metric, AOS when flagged)
    types = ['bbox', 'bev', '3d'] + (['aos'] if compute_aos else [])
    mAPbbox, mAPbev, mAP3d, mAPaos = do_eval(gt_annos, dt_annos, current_classes, min_overlaps, types, overlap_fn)
    mAPbbox, mAPbev, mAP3d = mAPbbox.mean(-1), mAPbev.mean(-1), mAP3d.mean(-1)
    if mAPaos is not None:
        mAPaos = mAPaos.mean(-1)
    result = ''
    for j, c in enumerate(current_classes):
        o_range = np.array(class_to_range[c])[[0, 2, 1]]
        o_range[1] = (o_range[2] - o_range[0]) / (o_range[1] - 1)
        result += f'{CLASS_TO_NAME[c]} ' + 'coco AP@{:.2f}:{:.2f}:{:.2f}:\n'.format(*o_range)
        result += f'bbox AP:{mAPbbox[j, 0]:.2f}, {mAPbbox[j, 1]:.2f}, {mAPbbox[j, 2]:.2f}\n'
        result += f'bev  AP:{mAPbev[j, 0]:.2f}, {mAPbev[j, 1]:.2f}, {mAPbev[j, 2]:.2f}\n'
        result += f'3d   AP:{mAP3d[j, 0]:.2f}, {mAP3d[j, 1]:.2f}, {mAP3d[j, 2]:.2f}\n'
        if compute_aos:
            result += f'aos  AP:{mAPaos[j, 0]:.2f}, {mAPaos[j, 1]:.2f}, {mAPaos[j, 2]:.2f}\n'
    return result


# ---------------------------------------------------------------------------------------------- detections -> annotations
def _limit_period(val, offset=0.5, period=np.pi):
    return val - np.floor(val / period + offset) * period


def convert_valid_bboxes(boxes_lidar, scores, labels, info, pcd_limit_range=(0, -40, -3, 70.4, 40, 0.0)):
    """LiDAR-frame detections of one sample -> camera-frame boxes + projected 2-D boxes, restricted to boxes that touch
    the image and whose bottom centre lies inside `pcd_limit_range` (datasets/kitti_dataset.py:587-674).
    boxes_lidar [n,7] (x,y,z,w,l,h,yaw) fp32 with z the bottom; info: {'image': {'image_idx', 'image_shape'},
    'calib': {'R0_rect','Tr_velo_to_cam','P2'}} as in the reference's info files."""
    boxes = np.array(boxes_lidar, dtype=np.float32).reshape(-1, 7)
    scores = np.asarray(scores).reshape(-1)
    labels = np.asarray(labels).reshape(-1)
    sample_idx = info['image']['image_idx']
    empty = dict(bbox=np.zeros([0, 4]), box3d_camera=np.zeros([0, 7]), box3d_lidar=np.zeros([0, 7]), scores=np.zeros([0]),
                 label_preds=np.zeros([0, 4]), sample_idx=sample_idx)
    boxes[:, 6] = _limit_period(boxes[:, 6] - np.float32(np.pi), 0.5, np.float32(np.pi * 2))       # :616-617
    if len(boxes) == 0:
        return empty
    rect = info['calib']['R0_rect'].astype(np.float32)
    trv2c = info['calib']['Tr_velo_to_cam'].astype(np.float32)
    p2 = info['calib']['P2'].astype(np.float32)
    rt = rect @ trv2c
    xyz1 = np.concatenate([boxes[:, :3], np.ones((len(boxes), 1), dtype=np.float32)], 1)
    cam_xyz = (xyz1 @ rt.T)[:, :3]
    cam = np.concatenate([cam_xyz, boxes[:, [4, 5, 3]], boxes[:, 6:7]], 1)          # sizes (l, h, w): box_3d_mode.py:104-108
    # camera-box corners, origin (0.5, 1, 0.5), rotated about the y axis (cam_box3d.py:99-139)
    cn = np.stack(np.unravel_index(np.arange(8), [2] * 3), axis=1)[[0, 1, 3, 2, 4, 5, 7, 6]].astype(np.float32)
    cn = cn - np.array([0.5, 1, 0.5], dtype=np.float32)
    corners = cam[:, None, 3:6] * cn[None]
    s, c = np.sin(cam[:, 6]), np.cos(cam[:, 6])
    x = corners[..., 0] * c[:, None] + corners[..., 2] * s[:, None]
    z = -corners[..., 0] * s[:, None] + corners[..., 2] * c[:, None]
    corners = np.stack([x, corners[..., 1], z], -1) + cam[:, None, :3]
    pts4 = np.concatenate([corners, np.ones(corners.shape[:2] + (1,), dtype=np.float32)], -1)
    uvw = pts4 @ p2.T
    uv = uvw[..., :2] / uvw[..., 2:3]
    box2d = np.concatenate([uv.min(1), uv.max(1)], 1)
    h, w = info['image']['image_shape'][:2]
    valid_cam = (box2d[:, 0] < w) & (box2d[:, 1] < h) & (box2d[:, 2] > 0) & (box2d[:, 3] > 0)
    lim = np.asarray(pcd_limit_range, dtype=np.float32)
    valid_pcd = ((boxes[:, :3] > lim[:3]) & (boxes[:, :3] < lim[3:])).all(-1)
    keep = valid_cam & valid_pcd
    if keep.sum() == 0:
        return empty
    return dict(bbox=box2d[keep], box3d_camera=cam[keep], box3d_lidar=boxes[keep], scores=scores[keep], label_preds=labels[keep],
                sample_idx=sample_idx)


def bbox2result_kitti(net_outputs, data_infos, class_names, pcd_limit_range=(0, -40, -3, 70.4, 40, 0.0), submission_prefix=None):
    """simple_test outputs -> KITTI-format detection annotations for kitti_eval (datasets/kitti_dataset.py:360-472).
    net_outputs: list of dict(boxes_3d, scores_3d, labels_3d) (boxes_3d: LiDARInstance3DBoxes or an [n,7] array)."""
    import os
    assert len(net_outputs) == len(data_infos), 'invalid list length of network outputs'
    if submission_prefix is not None:
        os.makedirs(submission_prefix, exist_ok=True)
    det_annos = []
    for pred, info in zip(net_outputs, data_infos):
        b = pred['boxes_3d']
        b = b.tensor if hasattr(b, 'tensor') else b
        b = b.detach().cpu().numpy() if hasattr(b, 'detach') else np.asarray(b)
        sc = pred['scores_3d']
        sc = sc.detach().cpu().numpy() if hasattr(sc, 'detach') else np.asarray(sc)
        lb = pred['labels_3d']
        lb = lb.detach().cpu().numpy() if hasattr(lb, 'detach') else np.asarray(lb)
        image_shape = np.asarray(info['image']['image_shape'][:2])
        d = convert_valid_bboxes(b, sc, lb, info, pcd_limit_range)
        n = len(d['bbox'])
        if n > 0:
            bbox = d['bbox'].copy()
            bbox[:, 2:] = np.minimum(bbox[:, 2:], image_shape[::-1])
            bbox[:, :2] = np.maximum(bbox[:, :2], 0)
            cam, lid = d['box3d_camera'], d['box3d_lidar']
            anno = dict(name=np.array([class_names[int(k)] for k in d['label_preds']]), truncated=np.zeros(n), occluded=np.zeros(n, dtype=np.int64),
                        alpha=-np.arctan2(-lid[:, 1], lid[:, 0]) + cam[:, 6], bbox=bbox, dimensions=cam[:, 3:6], location=cam[:, :3],
                        rotation_y=cam[:, 6], score=d['scores'])
        else:
            anno = dict(name=np.array([]), truncated=np.array([]), occluded=np.array([]), alpha=np.array([]), bbox=np.zeros([0, 4]),
                        dimensions=np.zeros([0, 3]), location=np.zeros([0, 3]), rotation_y=np.array([]), score=np.array([]))
        if submission_prefix is not None:
            with open(f"{submission_prefix}/{info['image']['image_idx']:06d}.txt", 'w') as f:
                for i in range(n):      # KITTI text format: dims written as h w l
                    f.write('{} -1 -1 {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f}\n'.format(
                        anno['name'][i], anno['alpha'][i], *anno['bbox'][i], anno['dimensions'][i][1], anno['dimensions'][i][2],
                        anno['dimensions'][i][0], *anno['location'][i], anno['rotation_y'][i], anno['score'][i]))
        anno['sample_idx'] = np.array([info['image']['image_idx']] * n, dtype=np.int64)
        det_annos.append(anno)
    return det_annos
