"""Parameter containers.  torch.nn.Module is used ONLY to hold tensors under the reference's state-dict
names (so released ImVoxelNet checkpoints load with load_state_dict); none of these modules computes
anything -- the arithmetic is in libimvoxel_hip.so.
"""
import math

import torch
from torch import nn


class ConvParams(nn.Module):
    """Holds `weight` [Cout,Cin,*k] and optional `bias` like nn.Conv{2,3}d."""

    def __init__(self, cin, cout, kernel, bias=False, dims=3):
        super().__init__()
        k = (kernel,) * dims if isinstance(kernel, int) else tuple(kernel)
        self.dims = dims
        self.weight = nn.Parameter(torch.empty((cout, cin) + k), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False) if bias else None
        fan_in = cin * int(torch.tensor(k).prod())
        nn.init.normal_(self.weight, 0.0, math.sqrt(2.0 / fan_in))  # Kaiming-normal (fan_in, ReLU gain)

    def forward(self, *a, **k):
        raise RuntimeError('ConvParams is a parameter container; the convolution runs in libimvoxel_hip.so')


class ConvTransposeParams(nn.Module):
    """Holds `weight` [Cin,Cout,*k] like nn.ConvTranspose3d (bias-free in the reference necks)."""

    def __init__(self, cin, cout, kernel, dims=3):
        super().__init__()
        k = (kernel,) * dims if isinstance(kernel, int) else tuple(kernel)
        self.weight = nn.Parameter(torch.empty((cin, cout) + k), requires_grad=False)
        nn.init.normal_(self.weight, 0.0, math.sqrt(2.0 / cin))

    def forward(self, *a, **k):
        raise RuntimeError('parameter container')


class BNParams(nn.Module):
    """Holds eval-mode BatchNorm statistics under nn.BatchNorm{2,3}d's names."""

    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(c), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c), requires_grad=False)
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def tensors(self):
        return (self.weight, self.bias, self.running_mean, self.running_var)

    def forward(self, *a, **k):
        raise RuntimeError('parameter container')


class ScaleParams(nn.Module):
    """mmcv.cnn.Scale: one learnable scalar `scale`."""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(float(scale)), requires_grad=False)


def randomize_(module, seed=0, residual_gain=0.2):
    """Synthetic weights for benchmarks/tests: Kaiming-normal convs, BN gamma~U(.5,1.5),
    beta~N(0,.1), mean~N(0,.1), var~U(.5,1.5).  The last BN of every residual branch (bn3 of a bottleneck,
    bn2 / norm2 of a 3-D basic block) gets its gamma scaled by `residual_gain`, as trained residual nets
    have, so activations stay O(1) through ~30 residual blocks instead of growing ~2x per block.
    Deterministic in `seed`."""
    g = torch.Generator().manual_seed(seed)
    for name, m in module.named_modules():
        last = name.rsplit('.', 1)[-1]
        if isinstance(m, BNParams) and last in ('bn3', 'norm2') or (isinstance(m, BNParams) and last == 'bn2' and
                                                                   not hasattr(_parent(module, name), 'conv3')):
            m._residual_tail = True
    for m in module.modules():
        if isinstance(m, (ConvParams, ConvTransposeParams)):
            w = m.weight
            fan_in = w[0].numel() if isinstance(m, ConvParams) else w.shape[0]
            w.copy_(torch.randn(w.shape, generator=g) * math.sqrt(2.0 / fan_in))
            if getattr(m, 'bias', None) is not None:
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
        elif isinstance(m, BNParams):
            m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            if getattr(m, '_residual_tail', False):
                m.weight.mul_(residual_gain)
            m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    return module


def _parent(root, name):
    mod = root
    for part in name.split('.')[:-1]:
        mod = getattr(mod, part)
    return mod


def invalidate_packed_on_load(module):
    """The device kernels work on PACKED copies of the parameters made by module.prepare().  Loading weights afterwards
    (load_state_dict, data.load_checkpoint -- on this module or on any parent) must not leave stale packed copies in
    use: the post-hook drops them, and the next forward_cl re-packs from the new parameters (fp32 storage; a model
    prepared in another storage dtype is re-prepared in that dtype by the detector's own hook)."""
    def _drop(mod, incompatible_keys):
        mod._device = None
    module.register_load_state_dict_post_hook(_drop)
    return module
