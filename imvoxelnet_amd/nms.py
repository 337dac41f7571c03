"""NMS entry points with the reference's names, running on the device through the C-ABI
(mmdet3d/ops/iou3d/iou3d_utils.py:25-71, mmdet3d/core/post_processing/box3d_nms.py:8-138).
Unlike the reference op nothing is copied to the host inside nms_gpu: the mask, the greedy scan and the
kept-index list stay on the device (the reference does cudaMalloc + a blocking D2H per call,
ops/iou3d/src/iou3d.cpp:110-146).
"""
import torch

from . import ops


def _nms(boxes, scores, thresh, rotated, pre_maxsize=None, post_max_size=None):
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = boxes[order].contiguous().float()
    keep, num = ops.nms_bev_sorted(b, thresh, rotated)
    keep = order[keep[:int(num.item())]].contiguous()
    if post_max_size is not None:
        keep = keep[:post_max_size]
    return keep


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """boxes [N,5] (x1,y1,x2,y2,ry), scores [N] -> kept indices (descending score)."""
    return _nms(boxes, scores, thresh, True, pre_maxsize, post_max_size)


def nms_normal_gpu(boxes, scores, thresh):
    return _nms(boxes, scores, thresh, False)


def boxes_iou_bev(boxes_a, boxes_b):
    return ops.boxes_overlap_bev(boxes_a.contiguous().float(), boxes_b.contiguous().float(), iou=True)


def boxes_overlap_bev(boxes_a, boxes_b):
    return ops.boxes_overlap_bev(boxes_a.contiguous().float(), boxes_b.contiguous().float(), iou=False)


def box3d_multiclass_nms(mlvl_bboxes, mlvl_bboxes_for_nms, mlvl_scores, score_thr, max_num, cfg, mlvl_dir_scores=None):
    """post_processing/box3d_nms.py:8-88 as ONE fused device call (ivx_multiclass_nms_bev: per-class filter + sort +
    rotated/normal NMS for all classes concurrently, class-major concat, final top-max_num) followed by gathers; the
    only host round trip is the survivor count.  Any n up to 65536 candidates (beyond 4096 the library sorts by rank and keeps
    the removal bits in LDS: same kept order); more than 64 classes fall back to the per-class loop over the same device NMS.
    (The single-class anchor-head configs use ops.anchor_head_get_bboxes.)"""
    num_classes = mlvl_scores.shape[1] - 1
    n = mlvl_bboxes.shape[0]
    if n > 65536 or num_classes > 64:
        return _box3d_multiclass_nms_loop(mlvl_bboxes, mlvl_bboxes_for_nms, mlvl_scores, score_thr, max_num, cfg, mlvl_dir_scores)
    if n == 0 or num_classes == 0:
        z = mlvl_scores.new_zeros
        return z((0, mlvl_bboxes.size(-1))), z((0,)), z((0,), dtype=torch.long), z((0,))
    idx, labels, cnt = ops.multiclass_nms_bev(mlvl_bboxes_for_nms.contiguous().float(), mlvl_scores.contiguous().float(), num_classes,
                                              score_thr, cfg['nms_thr'], cfg['use_rotate_nms'], max_num)
    k = int(cnt.item())
    idx, labels = idx[:k], labels[:k]
    bboxes = mlvl_bboxes[idx]
    scores = mlvl_scores[idx, labels]
    dir_scores = mlvl_dir_scores[idx] if mlvl_dir_scores is not None else mlvl_scores.new_zeros((0,))
    if k == 0:
        dir_scores = mlvl_scores.new_zeros((0,))
    return bboxes, scores, labels, dir_scores


def _box3d_multiclass_nms_loop(mlvl_bboxes, mlvl_bboxes_for_nms, mlvl_scores, score_thr, max_num, cfg, mlvl_dir_scores=None):
    """The reference's control flow (host loop over classes) on the device NMS: used when the fused op does not apply (more than
    64 classes, or more than 65536 candidates in total; every per-class call takes up to 65536 boxes above score_thr)."""
    num_classes = mlvl_scores.shape[1] - 1
    bboxes, scores, labels, dir_scores = [], [], [], []
    fn = nms_gpu if cfg['use_rotate_nms'] else nms_normal_gpu
    for i in range(num_classes):
        inds = mlvl_scores[:, i] > score_thr
        if not inds.any():
            continue
        s = mlvl_scores[inds, i]
        sel = fn(mlvl_bboxes_for_nms[inds, :], s, cfg['nms_thr'])
        bboxes.append(mlvl_bboxes[inds, :][sel])
        scores.append(s[sel])
        labels.append(mlvl_bboxes.new_full((len(sel),), i, dtype=torch.long))
        if mlvl_dir_scores is not None:
            dir_scores.append(mlvl_dir_scores[inds][sel])
    if bboxes:
        bboxes, scores, labels = torch.cat(bboxes), torch.cat(scores), torch.cat(labels)
        if mlvl_dir_scores is not None:
            dir_scores = torch.cat(dir_scores)
        if bboxes.shape[0] > max_num:
            inds = scores.sort(descending=True)[1][:max_num]
            bboxes, labels, scores = bboxes[inds, :], labels[inds], scores[inds]
            if mlvl_dir_scores is not None:
                dir_scores = dir_scores[inds]
    else:
        bboxes = mlvl_scores.new_zeros((0, mlvl_bboxes.size(-1)))
        scores = mlvl_scores.new_zeros((0,))
        labels = mlvl_scores.new_zeros((0,), dtype=torch.long)
        dir_scores = mlvl_scores.new_zeros((0,))
    return bboxes, scores, labels, dir_scores


def aligned_3d_nms(boxes, scores, classes, thresh):
    """post_processing/box3d_nms.py:91-138 on the device: boxes [n,6] corners -> picked indices."""
    pick, num = ops.aligned_3d_nms_dev(boxes.contiguous().float(), scores.contiguous().float(),
                                       classes.contiguous().long(), thresh)
    return pick[:int(num.item())]
