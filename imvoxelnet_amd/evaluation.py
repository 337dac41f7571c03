"""Indoor detection evaluation (mAP / mAR at IoU thresholds) with the reference's names and result keys
(mmdet3d/core/evaluation/indoor_eval.py:7-309) -- SURVEY.md section 8(f) rank 1: the AP code the
"mAP within 0.1 of the reference" claim is measured with.  Host-side numpy bookkeeping; the 3-D IoU matrix comes from
BaseInstance3DBoxes.overlaps, i.e. the rotated-BEV overlap kernel (ivx_boxes_overlap_bev) x height overlap.
"""
import numpy as np
import torch


def average_precision(recalls, precisions, mode='area'):
    """indoor_eval.py:7-52: area under the monotone precision envelope ('area') or 11-point interpolation."""
    if recalls.ndim == 1:
        recalls, precisions = recalls[np.newaxis, :], precisions[np.newaxis, :]
    assert recalls.shape == precisions.shape and recalls.ndim == 2
    n = recalls.shape[0]
    ap = np.zeros(n, dtype=np.float32)
    if mode == 'area':
        mrec = np.hstack((np.zeros((n, 1), recalls.dtype), recalls, np.ones((n, 1), recalls.dtype)))
        mpre = np.hstack((np.zeros((n, 1), recalls.dtype), precisions, np.zeros((n, 1), recalls.dtype)))
        mpre = np.maximum.accumulate(mpre[:, ::-1], axis=1)[:, ::-1]
        for i in range(n):
            ind = np.where(mrec[i, 1:] != mrec[i, :-1])[0]
            ap[i] = np.sum((mrec[i, ind + 1] - mrec[i, ind]) * mpre[i, ind + 1])
    elif mode == '11points':
        for i in range(n):
            for thr in np.arange(0, 1 + 1e-3, 0.1):
                precs = precisions[i, recalls[i, :] >= thr]
                ap[i] += precs.max() if precs.size > 0 else 0
            ap /= 11      # (sic) inside the loop, as the reference has it
    else:
        raise ValueError('Unrecognized mode, only "area" and "11points" are supported')
    return ap


def eval_det_cls(pred, gt, iou_thr=None):
    """indoor_eval.py:55-164: precision / recall / AP of ONE class.  pred {img_id: [(box, score), ...]},
    gt {img_id: [box, ...]}; greedy matching of score-sorted detections to the max-IoU ground truth per image."""
    recs, npos = {}, 0
    for img_id, boxes in gt.items():
        if len(boxes) != 0:
            bbox = boxes[0].new_box(torch.stack([b.tensor.reshape(-1) for b in boxes]).to(torch.float32))
        else:
            bbox = boxes
        npos += len(bbox)
        recs[img_id] = dict(bbox=bbox, det=[[False] * len(bbox) for _ in iou_thr])
    image_ids, confidence, ious = [], [], []
    for img_id, dets in pred.items():
        if len(dets) == 0:
            continue
        cur = dets[0][0].new_box(torch.stack([b.tensor.reshape(-1) for b, _ in dets]).to(torch.float32))
        image_ids += [img_id] * len(dets)
        confidence += [s for _, s in dets]
        gt_cur = recs[img_id]['bbox']
        if len(gt_cur) > 0:
            iou_cur = cur.overlaps(cur, gt_cur)
            ious += [iou_cur[i] for i in range(len(dets))]
        else:
            ious += [np.zeros(1) for _ in dets]
    order = np.argsort(-np.array(confidence))
    image_ids = [image_ids[x] for x in order]
    ious = [ious[x] for x in order]
    nd = len(image_ids)
    tp = [np.zeros(nd) for _ in iou_thr]
    fp = [np.zeros(nd) for _ in iou_thr]
    for d in range(nd):
        R = recs[image_ids[d]]
        iou_max, jmax = -np.inf, -1
        for j in range(len(R['bbox'])):
            if ious[d][j] > iou_max:
                iou_max, jmax = ious[d][j], j
        for t, thresh in enumerate(iou_thr):
            if iou_max > thresh and not R['det'][t][jmax]:
                tp[t][d] = 1.
                R['det'][t][jmax] = 1
            else:
                fp[t][d] = 1.
    out = []
    for t in range(len(iou_thr)):
        fpc, tpc = np.cumsum(fp[t]), np.cumsum(tp[t])
        recall = tpc / float(npos)
        precision = tpc / np.maximum(tpc + fpc, np.finfo(np.float64).eps)
        out.append((recall, precision, average_precision(recall, precision)))
    return out


def eval_map_recall(pred, gt, ovthresh=None):
    """indoor_eval.py:167-205: per-class results for every threshold; classes without predictions score zeros."""
    vals = {c: eval_det_cls(pred[c], gt[c], ovthresh) for c in gt if c in pred}
    recall, precision, ap = ([{} for _ in ovthresh] for _ in range(3))
    for c in gt:
        for t in range(len(ovthresh)):
            if c in pred:
                recall[t][c], precision[t][c], ap[t][c] = vals[c][t]
            else:
                recall[t][c], precision[t][c], ap[t][c] = np.zeros(1), np.zeros(1), np.zeros(1)
    return recall, precision, ap


def indoor_eval(gt_annos, dt_annos, metric, label2cat, logger=None, box_type_3d=None, box_mode_3d=None):
    """indoor_eval.py:208-309.  Same arguments and result keys ('<cat>_AP_0.25', 'mAP_0.25', '<cat>_rec_0.25',
    'mAR_0.25', ...).  Boxes must already be in box_mode_3d (mode conversion is outside the built path)."""
    assert len(dt_annos) == len(gt_annos)
    pred, gt = {}, {}
    for img_id, (det, gta) in enumerate(zip(dt_annos, gt_annos)):
        labels = det['labels_3d'].numpy()
        scores = det['scores_3d'].numpy()
        for i in range(len(labels)):
            label = int(labels[i])
            pred.setdefault(label, {}).setdefault(img_id, [])
            gt.setdefault(label, {}).setdefault(img_id, [])
            pred[label][img_id].append((det['boxes_3d'][i], scores[i]))
        if gta['gt_num'] != 0:
            gtb = gta['gt_boxes_upright_depth']
            gt_boxes = box_type_3d(gtb, box_dim=gtb.shape[-1], origin=(0.5, 0.5, 0.5))
            glabels = gta['class']
        else:
            gt_boxes, glabels = box_type_3d(np.array([], dtype=np.float32)), np.array([], dtype=np.int64)
        for i in range(len(glabels)):
            gt.setdefault(glabels[i], {}).setdefault(img_id, []).append(gt_boxes[i])
    rec, prec, ap = eval_map_recall(pred, gt, metric)
    ret = {}
    for t, thr in enumerate(metric):
        for label in ap[t]:
            ret[f'{label2cat[label]}_AP_{thr:.2f}'] = float(ap[t][label][0])
        ret[f'mAP_{thr:.2f}'] = float(np.mean(list(ap[t].values())))
        rl = []
        for label in rec[t]:
            ret[f'{label2cat[label]}_rec_{thr:.2f}'] = float(rec[t][label][-1])
            rl.append(rec[t][label][-1])
        ret[f'mAR_{thr:.2f}'] = float(np.mean(rl))
    if logger is not None:
        getattr(logger, 'info', print)('\n'.join(f'{k}: {v:.4f}' for k, v in ret.items()))
    return ret
