"""TEST INFRASTRUCTURE.  ctypes driver of oracle/_cpuabi/libimvoxel_cpu.so (the product's model-level C-ABI over the CPU restatement
of the op-level entry points): build the handle from an imvoxelnet_amd.ImVoxelNet module's configuration and state dict exactly as
imvoxelnet_amd/engine.py does for the HIP library, and run ivx_model_forward on host buffers.  Used by tests/ and by bench.py's
cpu_baseline leg (SURVEY 8d baseline (ii): "the build's own CPU restatement, same C-ABI, OpenMP, all host cores")."""
import ctypes as C
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    spec = importlib.util.spec_from_file_location('ivx_cpu_abi_build', os.path.join(HERE, 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib_path, _ = mod.build()
    L = C.CDLL(lib_path)
    L.ivx_last_error.restype = C.c_char_p
    for f in ('ivx_model_workspace_bytes', 'ivx_neck3d_workspace_bytes', 'ivx_model_detect_workspace_bytes'):
        getattr(L, f).restype = C.c_int64
    return L


class CpuModel:
    """The anchor-head families (KITTI / nuScenes without DCN) on the CPU restatement: CpuModel(model).forward(img, metas)."""

    def __init__(self, model):
        from imvoxelnet_amd import engine                       # host-only helpers: the configuration struct, no HIP library
        self.L = L = load()
        self.model = model
        self.cfg = cfg = engine.model_cfg(model, with_trunk=True)
        self.family = engine.family(model)
        if self.family is None:
            raise NotImplementedError('the model-level handle does not cover this module (engine.family)')
        self.h = C.c_void_p()
        self._ok(L.ivx_create(C.byref(cfg), C.byref(self.h)), 'ivx_create')
        for key, t in model.state_dict().items():
            if t.dtype.is_floating_point:
                self._load(key, t.detach().cpu().numpy())
        self._ok(L.ivx_weights_finalize(self.h, None), 'ivx_weights_finalize')
        if self.family == 'anchor':
            X, Y, Cn = C.c_int32(), C.c_int32(), C.c_int32()
            self._ok(L.ivx_neck3d_out_dims(self.h, 1, C.byref(X), C.byref(Y), C.byref(Cn)), 'ivx_neck3d_out_dims')
            anc = model.bbox_head.anchor_generator.grid_anchors([(Y.value, X.value)], device='cpu')[0].reshape(-1, 7).contiguous().float().numpy()
            self._load('anchors', anc)

    def _ok(self, rc, what):
        if rc != 0:
            raise RuntimeError(f'{what}: {self.L.ivx_last_error().decode()}')

    def _load(self, key, a):
        a = np.ascontiguousarray(a, np.float32)
        self._ok(self.L.ivx_weights_load(self.h, key.encode(), a.ctypes.data_as(C.c_void_p), (C.c_int64 * max(a.ndim, 1))(*a.shape), a.ndim), key)

    def forward(self, img, img_metas):
        """img [B, V, 3, H, W] float32 (torch or numpy, host) -> list of (boxes [n, 7], scores [n], labels [n]) numpy arrays."""
        vp = C.c_void_p
        x = np.ascontiguousarray(np.asarray(img, dtype=np.float32))
        B, V, _, H, W = x.shape
        proj, new_origin, crop = self.model._camera_setup(img_metas, 4, 'cpu')
        proj, new_origin, crop = (np.ascontiguousarray(t.numpy()) for t in (proj, new_origin, crop))
        n = self.L.ivx_model_workspace_bytes(self.h, B, V, H, W)
        if n < 0:
            raise RuntimeError(self.L.ivx_last_error().decode())
        raw = np.empty(n + 256, np.uint8)
        ws = raw.ctypes.data + (-raw.ctypes.data % 256)
        M = self.cfg.max_num
        boxes, scores = np.empty((B, M, 7), np.float32), np.empty((B, M), np.float32)
        labels, count = np.empty((B, M), np.int64), np.empty((B,), np.int32)
        self._ok(self.L.ivx_model_forward(self.h, x.ctypes.data_as(vp), B, V, H, W, proj.ctypes.data_as(vp), new_origin.ctypes.data_as(vp),
                                          crop.ctypes.data_as(vp), vp(ws), C.c_int64(n), boxes.ctypes.data_as(vp), scores.ctypes.data_as(vp),
                                          labels.ctypes.data_as(vp), count.ctypes.data_as(vp), None, None), 'ivx_model_forward')
        return [(boxes[b, :count[b]].copy(), scores[b, :count[b]].copy(), labels[b, :count[b]].copy()) for b in range(B)]

    def forward_levels(self, img, img_metas):
        """Indoor families: img [B, V, 3, H, W] -> ([levels, channels-last [B, X_l, Y_l, Z_l, C], finest first], valid [B, X, Y, Z] bool)
        = extract_feat through ivx_model_forward_levels."""
        vp = C.c_void_p
        x = np.ascontiguousarray(np.asarray(img, dtype=np.float32))
        B, V, _, H, W = x.shape
        proj, new_origin, crop = self.model._camera_setup(img_metas, 4, 'cpu')
        proj, new_origin, crop = (np.ascontiguousarray(t.numpy()) for t in (proj, new_origin, crop))
        n = self.L.ivx_model_workspace_bytes(self.h, B, V, H, W)
        if n < 0:
            raise RuntimeError(self.L.ivx_last_error().decode())
        raw = np.empty(n + 256, np.uint8)
        ws = raw.ctypes.data + (-raw.ctypes.data % 256)
        dims = ((C.c_int32 * 4) * 3)()
        self._ok(self.L.ivx_neck3d_levels(self.h, B, dims), 'ivx_neck3d_levels')
        outs = [np.empty((B, d[0], d[1], d[2], d[3]), np.float32) for d in dims if d[3] > 0]
        ptrs = (vp * 3)(*([o.ctypes.data for o in outs] + [None] * (3 - len(outs))))
        valid = np.empty((B,) + tuple(self.model.n_voxels), np.uint8)
        self._ok(self.L.ivx_model_forward_levels(self.h, x.ctypes.data_as(vp), B, V, H, W, proj.ctypes.data_as(vp), new_origin.ctypes.data_as(vp),
                                                 crop.ctypes.data_as(vp), vp(ws), C.c_int64(n), ptrs, valid.ctypes.data_as(vp), None),
                 'ivx_model_forward_levels')
        return outs, valid.astype(bool)

    def detect(self, img, img_metas):
        """simple_test through ivx_model_detect (every family with a head): img [B, V, 3, H, W] host array + the reference's img_metas ->
        list of (boxes [n, 7], scores [n], labels [n]) [, (angles [B,2], layouts [B,7]) with a LayoutHead]; valid mask in self.last_valid."""
        from imvoxelnet_amd._lib import SampleMeta
        vp = C.c_void_p
        x = np.ascontiguousarray(np.asarray(img, dtype=np.float32))
        B, V, _, H, W = x.shape
        metas, keep = (SampleMeta * B)(), []
        layout = bool(self.cfg.layout_head)
        for b, meta in enumerate(img_metas):
            K = np.zeros((4, 4), np.float32)
            Ki = np.asarray(meta['lidar2img']['intrinsic'], np.float32)
            K[:Ki.shape[0], :Ki.shape[1]] = Ki
            metas[b].intrinsic[:] = K.reshape(-1).tolist()
            if not layout:
                E = np.zeros((V, 4, 4), np.float32)
                for v, e in enumerate(meta['lidar2img']['extrinsic']):
                    e = np.asarray(e, np.float32)
                    E[v, :e.shape[0], :e.shape[1]] = e
                keep.append(E)
                metas[b].extrinsics = E.ctypes.data
            metas[b].origin[:] = [float(v) for v in np.asarray(meta['lidar2img']['origin'], np.float32)]
            metas[b].img_h, metas[b].img_w, metas[b].ori_h = int(meta['img_shape'][0]), int(meta['img_shape'][1]), int(meta['ori_shape'][0])
        n = self.L.ivx_model_detect_workspace_bytes(self.h, B, V, H, W)
        M = self.L.ivx_model_max_detections(self.h, B, V, H, W)
        if n < 0 or M < 0:
            raise RuntimeError(self.L.ivx_last_error().decode())
        raw = np.empty(n + 256, np.uint8)
        ws = raw.ctypes.data + (-raw.ctypes.data % 256)
        boxes, scores = np.empty((B, M, 7), np.float32), np.empty((B, M), np.float32)
        labels, count = np.empty((B, M), np.int64), np.empty((B,), np.int32)
        valid = np.empty((B,) + tuple(self.model.n_voxels), np.uint8)
        ang, lay = np.zeros((B, 2), np.float32), np.zeros((B, 7), np.float32)
        self._ok(self.L.ivx_model_detect(self.h, x.ctypes.data_as(vp), B, V, H, W, C.cast(metas, vp), vp(ws), C.c_int64(n), boxes.ctypes.data_as(vp),
                                         scores.ctypes.data_as(vp), labels.ctypes.data_as(vp), count.ctypes.data_as(vp), valid.ctypes.data_as(vp),
                                         ang.ctypes.data_as(vp) if layout else None, lay.ctypes.data_as(vp) if layout else None, None), 'ivx_model_detect')
        self.last_valid = valid.astype(bool)
        dets = [(boxes[b, :count[b]].copy(), scores[b, :count[b]].copy(), labels[b, :count[b]].copy()) for b in range(B)]
        return (dets, (ang, lay)) if layout else dets

    def close(self):
        if self.h:
            self.L.ivx_destroy(self.h)
            self.h = C.c_void_p()
