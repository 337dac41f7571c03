"""ctypes bindings for oracle/ivx_oracle.c (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE, 'libivx_oracle.so'])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libivx_oracle.so')
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.ivxo_box_overlap.restype = C.c_float
        _LIB.ivxo_iou_bev.restype = C.c_float
        for n in ('ivxo_nms_rotated_sorted', 'ivxo_nms_normal_sorted', 'ivxo_aligned_3d_nms', 'ivxo_num_threads'):
            getattr(_LIB, n).restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def get_points(n_voxels, voxel_size, origin):
    nv = np.ascontiguousarray(n_voxels, dtype=np.int64)
    out = np.empty((3, int(nv[0]), int(nv[1]), int(nv[2])), np.float32)
    lib().ivxo_get_points(_p(nv), _p(_f32(voxel_size)), _p(_f32(origin)), _p(out))
    return out


def compute_projection(intrinsic4, extrinsics, ratio):
    K = _f32(intrinsic4)
    assert K.shape == (4, 4)
    E = _f32(np.stack([np.asarray(e) for e in extrinsics]))
    V = E.shape[0]
    P = np.empty((V, 3, 4), np.float32)
    lib().ivxo_compute_projection(_p(K), _p(E), C.c_int(V), C.c_double(float(ratio)), _p(P))
    return P


def backproject(features, points, projection, height=None, width=None, want_idx=False):
    f = _f32(features)
    V, Cn, FH, FW = f.shape
    height = FH if height is None else int(height)
    width = FW if width is None else int(width)
    pts = _f32(points)
    shape3 = pts.shape[1:]
    N = int(np.prod(shape3))
    P = _f32(projection)
    vol = np.empty((V, Cn, N), np.float32)
    valid = np.empty((V, N), np.uint8)
    xi = np.empty((V, N), np.int64) if want_idx else None
    yi = np.empty((V, N), np.int64) if want_idx else None
    lib().ivxo_backproject(_p(f), V, Cn, FH, FW, height, width, _p(pts), _p(P), C.c_int64(N),
                           _p(vol), _p(valid), _p(xi), _p(yi))
    vol = vol.reshape((V, Cn) + shape3)
    valid = valid.reshape((V, 1) + shape3).astype(bool)
    if want_idx:
        return vol, valid, xi, yi
    return vol, valid


def backproject_mean(features, points, projection, height=None, width=None):
    f = _f32(features)
    V, Cn, FH, FW = f.shape
    height = FH if height is None else int(height)
    width = FW if width is None else int(width)
    pts = _f32(points)
    shape3 = pts.shape[1:]
    N = int(np.prod(shape3))
    P = _f32(projection)
    out = np.empty((Cn, N), np.float32)
    valid = np.empty((N,), np.uint8)
    lib().ivxo_backproject_mean(_p(f), V, Cn, FH, FW, height, width, _p(pts), _p(P), C.c_int64(N),
                                _p(out), _p(valid))
    return out.reshape((Cn,) + shape3), valid.reshape((1,) + shape3).astype(bool)


def conv3d(x, w, bias=None, stride=(1, 1, 1), padding=(0, 0, 0), scale=None, shift=None,
           residual=None, relu=False):
    x = _f32(x)
    w = _f32(w)
    B, Ci, D, H, W = x.shape
    Co, Ci2, kd, kh, kw = w.shape
    assert Ci == Ci2
    sd, sh, sw = stride
    pd, ph, pw = padding
    Do, Ho, Wo = (D + 2 * pd - kd) // sd + 1, (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    out = np.empty((B, Co, Do, Ho, Wo), np.float32)
    b = _f32(bias) if bias is not None else None
    sc = _f32(scale) if scale is not None else None
    sf = _f32(shift) if shift is not None else None
    rs = _f32(residual) if residual is not None else None
    lib().ivxo_conv3d(_p(x), B, Ci, D, H, W, _p(w), Co, kd, kh, kw, sd, sh, sw, pd, ph, pw,
                      _p(b), _p(sc), _p(sf), _p(rs), int(bool(relu)), _p(out))
    return out


def box_overlap(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().ivxo_box_overlap(_p(a), _p(b)))


def boxes_overlap_bev(a, b):
    a, b = _f32(a).reshape(-1, 5), _f32(b).reshape(-1, 5)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().ivxo_boxes_overlap_bev(_p(a), a.shape[0], _p(b), b.shape[0], _p(out))
    return out


def boxes_iou_bev(a, b):
    a, b = _f32(a).reshape(-1, 5), _f32(b).reshape(-1, 5)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().ivxo_boxes_iou_bev(_p(a), a.shape[0], _p(b), b.shape[0], _p(out))
    return out


def nms_sorted(boxes_sorted, thr, rotated=True):
    b = _f32(boxes_sorted).reshape(-1, 5)
    keep = np.empty((max(b.shape[0], 1),), np.int64)
    fn = lib().ivxo_nms_rotated_sorted if rotated else lib().ivxo_nms_normal_sorted
    n = fn(_p(b), b.shape[0], C.c_float(thr), _p(keep))
    return keep[:n].copy()


def aligned_3d_nms(boxes, scores, classes, order, thresh):
    b = _f32(boxes).reshape(-1, 6)
    s = _f32(scores)
    c = np.ascontiguousarray(classes, dtype=np.int64)
    o = np.ascontiguousarray(order, dtype=np.int64)
    pick = np.empty((max(b.shape[0], 1),), np.int64)
    n = lib().ivxo_aligned_3d_nms(_p(b), _p(s), _p(c), _p(o), b.shape[0], C.c_float(thresh), _p(pick))
    return pick[:n].copy()


def num_threads():
    return int(lib().ivxo_num_threads())
