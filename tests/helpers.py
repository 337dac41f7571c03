import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def sub(npz, prefix):
    """{key-without-prefix: array} for keys starting with `prefix`."""
    return {k[len(prefix):]: npz[k] for k in npz.files if k.startswith(prefix)}


def sd_from(npz, prefix):
    return {k: torch.from_numpy(v) for k, v in sub(npz, prefix).items()}


def meta_from_case(c):
    return dict(img_shape=tuple(int(v) for v in c['img_shape']), ori_shape=tuple(int(v) for v in c['ori_shape']),
                lidar2img=dict(intrinsic=c['intrinsic'], extrinsic=list(c['extrinsic']), origin=c['origin']))


def kitti_annos_from_golden(g):
    """Rebuild the per-image annotation / detection dicts of tests/golden/kitti_eval.npz."""
    out = []
    for tag in ('gt', 'dt'):
        counts = g[f'anno::{tag}::count']
        keys = [k.split('::')[2] for k in g.files if k.startswith(f'anno::{tag}::') and not k.endswith('::count')]
        annos, o = [], 0
        for n in counts:
            annos.append({k: g[f'anno::{tag}::{k}'][o:o + n] for k in keys})
            o += int(n)
        out.append(annos)
    return out
