"""-m gpu: the one-launch identity bottleneck (include/imvoxel.h ivx_bottleneck_fwd_pio, csrc/bottleneck.hip) of ResNet-50's stages 1 and 2
(reference: the mmdet ResNet the detector builds at mmdet3d/models/detectors/imvoxelnet.py:22 and runs at :48).  Checked against the
layer-wise pair chain (three ivx_conv_fwd_pio launches: same arithmetic, measured instead of a-priori scales for the two intermediates) and
against torch fp64 on the values the operands stand for, at ragged sizes (partial tiles, maps smaller than a tile) and at KITTI's / ScanNet's."""
import pytest
import torch

from gpu_util import assert_close
from test_gpu_pair_chain import make_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ia():
    import imvoxelnet_amd
    from imvoxelnet_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    return imvoxelnet_amd


def _block(P, seed, device='cuda'):
    """three FusedConv(chain=True) layers of an identity bottleneck with trained-looking BN statistics + their fp64 parameters"""
    from imvoxelnet_amd.conv import FusedConv
    g = torch.Generator().manual_seed(seed)
    C4 = 4 * P

    def bn(c):
        return (0.5 + torch.rand(c, generator=g), 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g), 0.5 + torch.rand(c, generator=g))
    w1 = torch.randn(P, C4, 1, 1, generator=g) * (2.0 / C4) ** 0.5
    w2 = torch.randn(P, P, 3, 3, generator=g) * (2.0 / (9 * P)) ** 0.5
    w3 = torch.randn(C4, P, 1, 1, generator=g) * (2.0 / P) ** 0.5
    bns = [bn(P), bn(P), bn(C4)]
    f1 = FusedConv(w1, bn=bns[0], relu=True, dims=2, chain=True).to(device)
    f2 = FusedConv(w2, bn=bns[1], padding=1, relu=True, dims=2, chain=True).to(device)
    f3 = FusedConv(w3, bn=bns[2], relu=True, dims=2, chain=True).to(device)
    return (f1, f2, f3), (w1, w2, w3), bns


def _ref64(x_cl, ws, bns):
    """fp64 bottleneck on channels-last x [B,1,H,W,C] (values the pair tensor stands for) -> [B,1,H,W,C] fp64"""
    import torch.nn.functional as F
    x = x_cl[:, 0].permute(0, 3, 1, 2).double()

    def bn(y, t):
        gmm, beta, mean, var = (v.double().to(y.device) for v in t)
        return (y - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5) * gmm[None, :, None, None] + beta[None, :, None, None]
    y = F.relu(bn(F.conv2d(x, ws[0].double().to(x.device)), bns[0]))
    y = F.relu(bn(F.conv2d(y, ws[1].double().to(x.device), padding=1), bns[1]))
    y = F.relu(bn(F.conv2d(y, ws[2].double().to(x.device)), bns[2]) + x)
    return y.permute(0, 2, 3, 1)[:, None]


@pytest.mark.parametrize('P,B,H,W', [(64, 1, 8, 16), (64, 2, 13, 37), (64, 1, 5, 9), (128, 1, 8, 16), (128, 2, 11, 21), (64, 2, 24, 80),
                                       (128, 3, 17, 48)])
def test_fused_bottleneck_vs_chain_and_fp64(ia, P, B, H, W):
    from imvoxelnet_amd import ops
    (f1, f2, f3), ws, bns = _block(P, 100 + P + H)
    g = torch.Generator().manual_seed(H * W + P)
    x = torch.relu(torch.randn(B, 1, H, W, 4 * P, generator=g) * torch.logspace(-1.5, 0.5, 4 * P)).cuda()
    xp = make_pair(x)
    xv = xp.float()                                     # the values the pair tensor stands for
    got = ops.bottleneck_fwd_pio(xp, f1, f2, f3)
    torch.cuda.synchronize()
    assert isinstance(got, ops.PairTensor) and got.shape == x.shape
    y = got.float()
    # layer-wise chain: the same kernels' arithmetic in three launches
    ch = f3(f2(f1(xp, out_pair=True), out_pair=True), res=xp, out_pair=True)
    yc = ch.float()
    ref = _ref64(xv.cpu(), ws, bns)
    rng = float(ref.abs().max())
    assert_close(f'fused vs layer-wise chain P{P} {B}x{H}x{W}', y, yc, rtol=0, atol=2e-5 * rng)
    assert_close(f'fused vs fp64 P{P} {B}x{H}x{W}', y.double().cpu(), ref, rtol=0, atol=2e-5 * rng)
    e_f = float((y.double().cpu() - ref).abs().max())
    e_c = float((yc.double().cpu() - ref).abs().max())
    print(f'max error vs fp64: fused {e_f:.3e}  chain {e_c:.3e}  (range {rng:.3e})')
    assert e_f <= 2.0 * e_c + 1e-6 * rng, 'the fused block must be as accurate as the layer-wise chain'
    # the scalar block: the recorded maximum is the maximum of the stored values; the scale is a power of two that keeps s * bound < 2^15
    amax = got.amax()
    assert abs(amax - float(y.abs().max())) <= 1e-6 * rng + 2.0 ** -20 * amax
    s = got.scale()
    assert s > 0 and (s * amax) < 32768.0 and float(torch.tensor(s).log2()) == round(float(torch.tensor(s).log2()))


def test_fused_bottleneck_chain_of_two(ia):
    """two blocks in a row: the second reads the first's output, scale and recorded maximum straight from the device"""
    from imvoxelnet_amd import ops
    blk_a, ws_a, bns_a = _block(64, 1)
    blk_b, ws_b, bns_b = _block(64, 2)
    x = torch.relu(torch.randn(2, 1, 19, 33, 256, generator=torch.Generator().manual_seed(5))).cuda()
    xp = make_pair(x)
    y1 = ops.bottleneck_fwd_pio(xp, *blk_a)
    y2 = ops.bottleneck_fwd_pio(y1, *blk_b)
    ref = _ref64(_ref64(xp.float().cpu(), ws_a, bns_a).float(), ws_b, bns_b)       # (the fp32 rounding of the intermediate is far below the bar)
    assert_close('two fused blocks vs fp64', y2.float().double().cpu(), ref, rtol=0, atol=4e-5 * float(ref.abs().max()))


def test_fused_bottleneck_nonfinite_input_stays_local(ia):
    """an Inf in the input makes the bound non-finite: fixed scale, saturating split -- finite outputs away from the Inf's receptive field"""
    from imvoxelnet_amd import ops
    blk, ws, bns = _block(64, 3)
    x = torch.relu(torch.randn(1, 1, 16, 32, 256, generator=torch.Generator().manual_seed(6))).cuda()
    xp = make_pair(x)
    xp.slots[:ops.AMAX_SLOTS].view(torch.float32)[3] = float('inf')      # as a producer that met an Inf would have recorded
    y = ops.bottleneck_fwd_pio(xp, *blk).float()
    assert bool(torch.isfinite(y).all())
    ref = _ref64(xp.float().cpu(), ws, bns)
    assert_close('fixed-scale path', y.double().cpu(), ref, rtol=0, atol=2e-3 * float(ref.abs().max()))


def test_fused_bottleneck_rejects(ia):
    from imvoxelnet_amd import ops
    assert ops.bottleneck_supported(4, 96, 320, 64) and ops.bottleneck_supported(50, 60, 80, 128)
    assert not ops.bottleneck_supported(4, 24, 80, 256) and not ops.bottleneck_supported(4, 96, 320, 32)
    blk, _, _ = _block(64, 4)
    x = make_pair(torch.randn(1, 1, 8, 16, 128).cuda())
    with pytest.raises(ValueError):
        ops.bottleneck_fwd_pio(x, *blk)


@pytest.mark.parametrize('name,P,B,H,W', [('kitti s1', 64, 4, 96, 320), ('kitti s2', 128, 4, 48, 160), ('scannet x20 s1', 64, 20, 120, 160)])
def test_fused_bottleneck_full_size_vs_chain(ia, name, P, B, H, W):
    from imvoxelnet_amd import ops
    (f1, f2, f3), ws, bns = _block(P, 7)
    x = torch.relu(torch.randn(B, 1, H, W, 4 * P, generator=torch.Generator().manual_seed(8))).cuda()
    xp = make_pair(x)
    y = ops.bottleneck_fwd_pio(xp, f1, f2, f3).float()
    yc = f3(f2(f1(xp, out_pair=True), out_pair=True), res=xp, out_pair=True).float()
    assert_close(f'{name}: fused vs chain', y, yc, rtol=0, atol=2e-5 * float(yc.abs().max()))


# ---------------------------------------------------------------------------------------------- the first block of stage 1 (shortcut conv), one launch
def _proj_block(seed, P=64, cin=64, device='cuda'):
    """layer1.0 of ResNet-50: conv1 (cin -> P), conv2 3x3, conv3 (P -> 4P) + the 1x1 shortcut conv (cin -> 4P), each with its BatchNorm"""
    from imvoxelnet_amd import ops
    from imvoxelnet_amd.conv import FusedConv
    g = torch.Generator().manual_seed(seed)
    C4 = 4 * P

    def bn(c):
        return (0.5 + torch.rand(c, generator=g), 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g), 0.5 + torch.rand(c, generator=g))
    w1 = torch.randn(P, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    w2 = torch.randn(P, P, 3, 3, generator=g) * (2.0 / (9 * P)) ** 0.5
    w3 = torch.randn(C4, P, 1, 1, generator=g) * (2.0 / P) ** 0.5
    wd = torch.randn(C4, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    bns = [bn(P), bn(P), bn(C4), bn(C4)]
    f1 = FusedConv(w1, bn=bns[0], relu=True, dims=2, chain=True).to(device)
    f2 = FusedConv(w2, bn=bns[1], padding=1, relu=True, dims=2, chain=True).to(device)
    f3 = FusedConv(w3, bn=bns[2], relu=True, dims=2, chain=True).to(device)
    fd = FusedConv(wd, bn=bns[3], relu=False, dims=2, chain=True).to(device)
    bank = ops.ProjBank(f3, fd).to(device)
    return (f1, f2, f3, fd, bank), (w1, w2, w3, wd), bns


def _proj_ref64(x_cl, ws, bns):
    import torch.nn.functional as F
    x = x_cl[:, 0].permute(0, 3, 1, 2).double()

    def bn(y, t):
        gmm, beta, mean, var = (v.double().to(y.device) for v in t)
        return (y - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5) * gmm[None, :, None, None] + beta[None, :, None, None]
    y = F.relu(bn(F.conv2d(x, ws[0].double()), bns[0]))
    y = F.relu(bn(F.conv2d(y, ws[1].double(), padding=1), bns[1]))
    y = F.relu(bn(F.conv2d(y, ws[2].double()), bns[2]) + bn(F.conv2d(x, ws[3].double()), bns[3]))
    return y.permute(0, 2, 3, 1)[:, None]


def _proj_chain(xp, f1, f2, f3, fd):
    """the four-launch form of the pair chain (backbones._Bottleneck.forward_cl): the shortcut conv writes fp32, conv3 adds it in its epilogue"""
    return f3(f2(f1(xp, out_pair=True), out_pair=True), res=fd(xp), out_pair=True)


@pytest.mark.parametrize('B,H,W', [(1, 8, 16), (2, 13, 37), (1, 5, 9), (2, 24, 80), (3, 17, 48)])
def test_fused_proj_bottleneck_vs_chain_and_fp64(ia, B, H, W):
    from imvoxelnet_amd import ops
    (f1, f2, f3, fd, bank), ws, bns = _proj_block(300 + H)
    g = torch.Generator().manual_seed(H * W + 1)
    x = torch.relu(torch.randn(B, 1, H, W, 64, generator=g) * torch.logspace(-1.5, 0.5, 64)).cuda()
    xp = make_pair(x)
    xv = xp.float()
    got = ops.bottleneck_proj_fwd_pio(xp, f1, f2, bank)
    torch.cuda.synchronize()
    assert isinstance(got, ops.PairTensor) and got.shape == (B, 1, H, W, 256)
    y = got.float()
    yc = _proj_chain(xp, f1, f2, f3, fd).float()
    ref = _proj_ref64(xv.cpu(), ws, bns)
    rng = float(ref.abs().max())
    assert_close(f'fused projection block vs four-launch chain {B}x{H}x{W}', y, yc, rtol=0, atol=2e-5 * rng)
    assert_close(f'fused projection block vs fp64 {B}x{H}x{W}', y.double().cpu(), ref, rtol=0, atol=2e-5 * rng)
    e_f = float((y.double().cpu() - ref).abs().max())
    e_c = float((yc.double().cpu() - ref).abs().max())
    print(f'max error vs fp64: fused {e_f:.3e}  chain {e_c:.3e}  (range {rng:.3e})')
    assert e_f <= 2.0 * e_c + 1e-6 * rng, 'the fused block must be as accurate as the layer-wise chain'
    amax = got.amax()
    assert abs(amax - float(y.abs().max())) <= 1e-6 * rng + 2.0 ** -20 * amax
    s = got.scale()
    assert s > 0 and (s * amax) < 32768.0 and float(torch.tensor(s).log2()) == round(float(torch.tensor(s).log2()))


def test_fused_proj_then_identity_blocks(ia):
    """layer1.0 -> layer1.1 as two launches: the identity block reads the projection block's output, scale and maximum from the device"""
    from imvoxelnet_amd import ops
    (f1, f2, f3, fd, bank), ws_a, bns_a = _proj_block(11)
    blk_b, ws_b, bns_b = _block(64, 12)
    x = torch.relu(torch.randn(2, 1, 19, 33, 64, generator=torch.Generator().manual_seed(13))).cuda()
    xp = make_pair(x)
    y2 = ops.bottleneck_fwd_pio(ops.bottleneck_proj_fwd_pio(xp, f1, f2, bank), *blk_b)
    ref = _ref64(_proj_ref64(xp.float().cpu(), ws_a, bns_a).float(), ws_b, bns_b)
    assert_close('projection + identity block vs fp64', y2.float().double().cpu(), ref, rtol=0, atol=4e-5 * float(ref.abs().max()))


def test_fused_proj_bottleneck_nonfinite_and_rejects(ia):
    from imvoxelnet_amd import ops
    (f1, f2, f3, fd, bank), ws, bns = _proj_block(14)
    x = torch.relu(torch.randn(1, 1, 16, 32, 64, generator=torch.Generator().manual_seed(15))).cuda()
    xp = make_pair(x)
    xp.slots[:ops.AMAX_SLOTS].view(torch.float32)[3] = float('inf')
    y = ops.bottleneck_proj_fwd_pio(xp, f1, f2, bank).float()
    assert bool(torch.isfinite(y).all())
    ref = _proj_ref64(xp.float().cpu(), ws, bns)
    assert_close('fixed-scale path', y.double().cpu(), ref, rtol=0, atol=2e-3 * float(ref.abs().max()))
    assert ops.bottleneck_proj_supported(4, 96, 320, 64, 64) and not ops.bottleneck_proj_supported(4, 48, 160, 128, 256)
    with pytest.raises(ValueError):
        ops.bottleneck_proj_fwd_pio(make_pair(torch.randn(1, 1, 8, 16, 128).cuda()), f1, f2, bank)


@pytest.mark.parametrize('name,B,H,W', [('kitti', 4, 96, 320), ('scannet x20', 20, 120, 160)])
def test_fused_proj_bottleneck_full_size_vs_chain(ia, name, B, H, W):
    from imvoxelnet_amd import ops
    (f1, f2, f3, fd, bank), ws, bns = _proj_block(16)
    x = torch.relu(torch.randn(B, 1, H, W, 64, generator=torch.Generator().manual_seed(17))).cuda()
    xp = make_pair(x)
    y = ops.bottleneck_proj_fwd_pio(xp, f1, f2, bank).float()
    yc = _proj_chain(xp, f1, f2, f3, fd).float()
    assert_close(f'{name}: fused projection block vs chain', y, yc, rtol=0, atol=2e-5 * float(yc.abs().max()))
