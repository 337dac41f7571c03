"""Anchor3DRangeGenerator and DeltaXYZWLHRBBoxCoder under the reference's registry names
(mmdet3d/core/anchor/anchor_3d_generator.py:7-209, core/bbox/coders/delta_xyzwhlr_bbox_coder.py).
The anchor grid is a per-config constant: it is generated once on the host with the same torch ops the
reference's CPU path uses (linspace centres -- NOT idx*voxel) and cached on the device, instead of being
regenerated on every get_bboxes call (anchor3d_head.py:403-407).
"""
import torch

from .registry import ANCHOR_GENERATORS, BBOX_CODERS


@ANCHOR_GENERATORS.register_module()
class Anchor3DRangeGenerator:
    def __init__(self, ranges, sizes=((1.6, 3.9, 1.56),), scales=(1,), rotations=(0, 1.5707963), custom_values=(),
                 reshape_out=True, size_per_range=True):
        ranges = [list(r) for r in ranges]
        sizes = [list(s) for s in sizes]
        if size_per_range:
            if len(sizes) != len(ranges):
                assert len(ranges) == 1
                ranges = ranges * len(sizes)
            assert len(ranges) == len(sizes)
        else:
            assert len(ranges) == 1
        self.ranges, self.sizes, self.scales = ranges, sizes, list(scales)
        self.rotations, self.custom_values = list(rotations), tuple(custom_values)
        self.reshape_out, self.size_per_range = reshape_out, size_per_range
        self._cache = {}

    @property
    def num_base_anchors(self):
        return len(self.rotations) * torch.tensor(self.sizes).reshape(-1, 3).size(0)

    @property
    def num_levels(self):
        return len(self.scales)

    def anchors_single_range(self, feature_size, anchor_range, scale=1, sizes=((1.6, 3.9, 1.56),),
                             rotations=(0, 1.5707963), device='cpu'):
        """-> [D, H, W, n_sizes, n_rot, 7(+custom)]: (x, y, z, w, l, h, r)."""
        if len(feature_size) == 2:
            feature_size = [1, feature_size[0], feature_size[1]]
        rg = torch.tensor(anchor_range, dtype=torch.float32)
        zc = torch.linspace(rg[2], rg[5], feature_size[0])
        yc = torch.linspace(rg[1], rg[4], feature_size[1])
        xc = torch.linspace(rg[0], rg[3], feature_size[2])
        sz = torch.tensor(sizes, dtype=torch.float32).reshape(-1, 3) * scale
        rot = torch.tensor(rotations, dtype=torch.float32)
        D, H, W, S, R = len(zc), len(yc), len(xc), sz.shape[0], len(rot)
        ncus = len(self.custom_values)
        out = torch.zeros(D, H, W, S, R, 7 + ncus)
        out[..., 0] = xc.view(1, 1, W, 1, 1)
        out[..., 1] = yc.view(1, H, 1, 1, 1)
        out[..., 2] = zc.view(D, 1, 1, 1, 1)
        out[..., 3:6] = sz.view(1, 1, 1, S, 1, 3)
        out[..., 6] = rot.view(1, 1, 1, 1, R)
        return out.to(device)

    def single_level_grid_anchors(self, featmap_size, scale, device='cpu'):
        if not self.size_per_range:
            return self.anchors_single_range(featmap_size, self.ranges[0], scale, self.sizes, self.rotations, device)
        parts = [self.anchors_single_range(featmap_size, r, scale, s, self.rotations, device)
                 for r, s in zip(self.ranges, self.sizes)]
        return torch.cat(parts, dim=-3)

    def grid_anchors(self, featmap_sizes, device='cpu'):
        assert self.num_levels == len(featmap_sizes)
        outs = []
        for i in range(self.num_levels):
            key = (tuple(featmap_sizes[i]), str(device), i)
            if key not in self._cache:
                a = self.single_level_grid_anchors(tuple(featmap_sizes[i]), self.scales[i], device='cpu')
                if self.reshape_out:
                    a = a.reshape(-1, a.size(-1))
                self._cache[key] = a.contiguous().to(device)
            outs.append(self._cache[key])
        return outs


@BBOX_CODERS.register_module()
class DeltaXYZWLHRBBoxCoder:
    def __init__(self, code_size=7):
        self.code_size = code_size

    @staticmethod
    def decode(anchors, deltas):
        """coders/delta_xyzwhlr_bbox_coder.py:56-90 (the device tail applies the same formulas in-kernel)."""
        xa, ya, za, wa, la, ha, ra = torch.split(anchors[..., :7], 1, dim=-1)
        xt, yt, zt, wt, lt, ht, rt = torch.split(deltas[..., :7], 1, dim=-1)
        za = za + ha / 2
        diagonal = torch.sqrt(la ** 2 + wa ** 2)
        xg, yg, zg = xt * diagonal + xa, yt * diagonal + ya, zt * ha + za
        lg, wg, hg = torch.exp(lt) * la, torch.exp(wt) * wa, torch.exp(ht) * ha
        rg = rt + ra
        zg = zg - hg / 2
        extra = [t + a for t, a in zip(torch.split(deltas[..., 7:], 1, dim=-1), torch.split(anchors[..., 7:], 1, dim=-1))] \
            if anchors.shape[-1] > 7 else []
        return torch.cat([xg, yg, zg, wg, lg, hg, rg, *extra], dim=-1)
