"""-m gpu: the one-launch head of the 2-D trunk (include/imvoxel.h ivx_stem_pool_fwd_pair, csrc/stem.hip): conv 7x7 s2 p3 + BN + ReLU +
MaxPool2d(3, 2, 1) from the NCHW image to the pair map (mmdet ResNet stem; reference call site mmdet3d/models/detectors/imvoxelnet.py:48).
Checked against torch fp64 and against the three-launch head (layout change, fp32-MFMA stem, pair max-pool) it replaces."""
import pytest
import torch

from gpu_util import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ia():
    import imvoxelnet_amd
    from imvoxelnet_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    return imvoxelnet_amd


def _stem(seed):
    from imvoxelnet_amd.conv import FusedConv
    from imvoxelnet_amd import ops
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(64, 3, 7, 7, generator=g) * (2.0 / 147) ** 0.5
    bn = (0.5 + torch.rand(64, generator=g), 0.2 * torch.randn(64, generator=g), 0.1 * torch.randn(64, generator=g), 0.5 + torch.rand(64, generator=g))
    f = FusedConv(w, bn=bn, stride=2, padding=3, relu=True, dims=2, chain=True).to('cuda')
    fr, sp = ops.stem_pool_pack_filters(w, f._scale_host)
    return f, fr.cuda(), sp.cuda(), w, bn


def _ref64(img, w, bn):
    import torch.nn.functional as F
    gmm, beta, mean, var = (v.double() for v in bn)
    y = F.conv2d(img.double(), w.double(), stride=2, padding=3)
    y = (y - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5) * gmm[None, :, None, None] + beta[None, :, None, None]
    y = F.max_pool2d(F.relu(y), 3, 2, 1)
    return y.permute(0, 2, 3, 1)[:, None]


@pytest.mark.parametrize('N,H,W', [(1, 32, 64), (2, 37, 53), (1, 7, 9), (3, 64, 130), (2, 96, 160), (1, 129, 31)])
def test_stem_pool_vs_fp64_and_three_launch_head(ia, N, H, W):
    from imvoxelnet_amd import ops
    f, fr, sp, w, bn = _stem(H + W)
    img = (torch.randn(N, 3, H, W, generator=torch.Generator().manual_seed(H * W)) * 1.5).cuda()
    got = ops.stem_pool_pair(img, fr, sp, f.shift, f.wbound, f.sbound)
    torch.cuda.synchronize()
    ref = _ref64(img.cpu(), w, bn)
    assert got.shape == tuple(ref.shape)
    y = got.float()
    rng = float(ref.abs().max())
    assert_close(f'one-launch stem vs fp64 {N}x{H}x{W}', y.double().cpu(), ref, rtol=0, atol=2e-5 * rng)
    # the three-launch head: fp32 MFMA stem -> pair max-pool with the same bound rule
    x4 = ops.to_channels_last_amax(img, pad_to=4)
    old = ops.maxpool2d_pair(f(x4), ops.slots_of(x4), f.wbound, f.sbound, 3, 2, 1)
    assert_close('one-launch vs three-launch head', y, old.float(), rtol=0, atol=2e-5 * rng)
    assert got.scale() == old.scale()                       # same bound, same power of two
    assert abs(got.amax() - float(y.abs().max())) <= 1e-6 * rng + 2.0 ** -20 * got.amax()


def test_stem_pool_kitti_size(ia):
    from imvoxelnet_amd import ops
    f, fr, sp, w, bn = _stem(3)
    img = torch.randn(4, 3, 384, 1280, generator=torch.Generator().manual_seed(5)).cuda()
    got = ops.stem_pool_pair(img, fr, sp, f.shift, f.wbound, f.sbound)
    x4 = ops.to_channels_last_amax(img, pad_to=4)
    old = ops.maxpool2d_pair(f(x4), ops.slots_of(x4), f.wbound, f.sbound, 3, 2, 1)
    assert got.shape == (4, 1, 96, 320, 64)
    assert_close('KITTI head: one launch vs three', got.float(), old.float(), rtol=0, atol=2e-5 * float(old.float().abs().max()))


def test_stem_pool_nonfinite_image(ia):
    """an Inf pixel: fixed scales, saturating splits -- outputs away from the pixel's receptive field stay finite and right"""
    from imvoxelnet_amd import ops
    f, fr, sp, w, bn = _stem(9)
    img = torch.randn(1, 3, 64, 96, generator=torch.Generator().manual_seed(2)).cuda()
    img[0, 1, 5, 7] = float('inf')
    y = ops.stem_pool_pair(img, fr, sp, f.shift, f.wbound, f.sbound).float()
    clean = img.clone()
    clean[0, 1, 5, 7] = 0.0
    ref = _ref64(clean.cpu(), w, bn)
    far = torch.ones(ref.shape[2], ref.shape[3], dtype=torch.bool)
    far[:5, :6] = False                                       # pooled pixels whose 7x7 / 3x3 windows can touch image pixel (5, 7)
    assert bool(torch.isfinite(y[0, 0][far.cuda()]).all())
    assert_close('away from the Inf', y[0, 0][far.cuda()].double().cpu(), ref[0, 0][far], rtol=0, atol=5e-3 * float(ref.abs().max()))
