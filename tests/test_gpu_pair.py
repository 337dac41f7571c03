"""-m gpu: split-operand MFMA (include/imvoxel.h IVX_BF16_PAIR / IVX_F16_PAIR, ivx_conv_desc.wino_operands): the pair conversions bit
for bit, the three-product kernel against the validation kernel and fp64, and the Winograd form on fp16 pairs against the fp32 form's
own error.  Runs on the MI355X box."""
import numpy as np
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


def _unpair(t, c):
    """pair tensor [..., 2c] (bf16 / fp16 bits) -> (hi, lo) as fp32 [..., c]"""
    v = t.float().reshape(*t.shape[:-1], c // 16, 2, 16)
    return v[..., 0, :].reshape(*t.shape[:-1], c), v[..., 1, :].reshape(*t.shape[:-1], c)


def test_pair_split_kernels_bit_exact(ia):
    """ivx_bf16_pair_split / ivx_f16_pair_split against the same conversions done by torch on the host (round to nearest even):
    hi = half(s x), lo = half(s x - hi), 16-channel groups [hi | lo]; fp16 saturates at +-65504."""
    import ctypes as C
    from imvoxelnet_amd import _lib, ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 5, 7, 2, 64, generator=g) * torch.logspace(-6, 4, 64)       # 10 decades across the channels
    x[0, 0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1e-30, -3e38, 65504.0, 70000.0, -1e6, 2.0 ** -20])
    xd = x.cuda()
    p = ops.bf16_pair_split(xd)
    assert p.dtype == torch.bfloat16 and tuple(p.shape) == (3, 5, 7, 2, 128)
    hi, lo = _unpair(p.cpu(), 64)
    rh = x.to(torch.bfloat16).float()
    rl = (x - rh).to(torch.bfloat16).float()
    rl[~torch.isfinite(rh)] = lo[~torch.isfinite(rh)]                              # beyond bf16's range hi is inf and lo is whatever inf - inf gives
    assert torch.equal(hi, rh) and torch.equal(lo[torch.isfinite(rh)], rl[torch.isfinite(rh)])
    fin = torch.isfinite(rh)
    assert float(((hi + lo)[fin] - x[fin]).abs().max() / x[fin].abs().max()) < 2.0 ** -16
    for scale in (1.0, 2.0 ** -4, 2.0 ** 10):
        out = torch.empty(3, 5, 7, 2, 128, device='cuda', dtype=torch.float16)
        _lib.check(_lib.lib().ivx_f16_pair_split(C.c_void_p(xd.data_ptr()), xd.numel(), C.c_float(scale), C.c_void_p(out.data_ptr()),
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'ivx_f16_pair_split')
        hi, lo = _unpair(out.cpu(), 64)
        xs = (x * scale).clamp(-65504.0, 65504.0)
        rh = xs.to(torch.float16).float()
        rl = (xs - rh).to(torch.float16).float()
        assert torch.equal(hi, rh) and torch.equal(lo, rl), scale
    with pytest.raises(ValueError):
        ops.bf16_pair_split(torch.zeros(2, 24, device='cuda'))                     # channel count not a multiple of 16


@pytest.mark.parametrize('case', [
    # B, (D,H,W), Cin, Cout, kernel, stride, pad, residual, relu, layout
    (2, (9, 11, 6), 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, True, 1),
    (1, (8, 10, 12), 64, 128, (3, 3, 3), (1, 1, 2), (1, 1, 1), False, True, 1),
    (2, (1, 24, 40), 256, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), False, False, 1),
    (1, (1, 30, 44), 48, 40, (1, 3, 3), (1, 2, 2), (0, 1, 1), True, True, 0),       # Cin % 16 == 0 only: tap-major filters
    (1, (6, 6, 3), 256, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, True, 1),
])
def test_conv_bf16_pair_operands(ia, case):
    """ivx_conv_fwd_ws with in_dtype IVX_BF16_PAIR (hi*hi + hi*lo + lo*hi on the bf16 matrix cores, fp32 accumulate and epilogue) against
    the validation kernel on the SAME pair operands (plain fp32 products of hi + lo: differs by the dropped lo*lo term and the
    summation order) and against torch fp64 on the unsplit values (adds the 2^-17 operand rounding)."""
    from imvoxelnet_amd import ops
    from imvoxelnet_amd.conv import pack_pair_weights
    B, (D, H, W), ci, co, k, st, pad, use_res, relu, layout = case
    g = torch.Generator().manual_seed(ci * 7 + co)
    x = torch.randn(B, D, H, W, ci, generator=g).abs_()
    w = torch.randn(co, ci, *k, generator=g) * (2.0 / (ci * k[0] * k[1] * k[2])) ** 0.5
    scale, shift = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1
    tref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.double(), stride=st, padding=pad).permute(0, 2, 3, 4, 1)
    tref = tref * scale.double() + shift.double()
    res = torch.randn(tref.shape, generator=g) if use_res else None
    if use_res:
        tref = tref + res.double()
    if relu:
        tref = tref.clamp_min(0)
    assert ops.conv_pair_supported(tuple(x.shape), co, k, st, pad, layout)
    wp = pack_pair_weights(w.permute(0, 2, 3, 4, 1).contiguous(), layout).cuda()
    xp = ops.bf16_pair_split(x.cuda())
    rd = res.cuda() if use_res else None
    args = (scale.cuda(), shift.cuda(), k, st, pad, relu, rd)
    got = ops.conv_fwd(xp, wp, *args, wgt_layout=layout, pair=True)
    naive = ops.conv_fwd(xp, wp, *args, wgt_layout=layout, pair=True, naive=True)
    rng = float(tref.abs().max())
    assert got.dtype == torch.float32 and got.shape == tref.shape
    assert_close('pair MFMA vs validation kernel on the same operands', got, naive, 0, 1e-5 * rng)
    assert_close('pair MFMA vs torch fp64', got, tref.float(), 0, 2e-5 * rng)


@pytest.mark.parametrize('case', [
    # B, (D,H,W), Cin, Cout, stride, residual: the layers the default rule sends to the split-operand form
    (1, (24, 26, 12), 64, 128, (2, 2, 2), False),      # NuScenesImVoxelNeck's first _get_conv (necks/imvoxelnet.py:133), reduced
    (1, (20, 20, 8), 256, 512, (2, 2, 2), False),      # FastIndoorImVoxelNeck BasicBlock3dV2.conv1 of a down layer (necks/imvoxelnet.py:237)
    (1, (10, 10, 4), 512, 512, (1, 1, 1), True),       # Atlas encoder, coarsest level: 400 positions, below the Winograd form's minimum
])
def test_fusedconv_split_operand_rule(ia, case, monkeypatch):
    """FusedConv's default rule (conv.py pair_mode = -1; csrc/model.cpp plan_conv): 3x3x3 layers the Winograd form does not take run as split pass +
    three-product bf16 MFMA kernel when the neck's GEMMs use 16-bit operands, and on fp32 MFMA otherwise; both against torch fp64."""
    from imvoxelnet_amd.conv import FusedConv
    B, (D, H, W), ci, co, st, use_res = case
    g = torch.Generator().manual_seed(ci + co)
    x = torch.randn(B, D, H, W, ci, generator=g).abs_()
    w = torch.randn(co, ci, 3, 3, 3, generator=g) * (2.0 / (ci * 27)) ** 0.5
    bn = (torch.rand(co, generator=g) + .5, torch.randn(co, generator=g) * .1, torch.randn(co, generator=g) * .1, torch.rand(co, generator=g) + .5)
    monkeypatch.setattr(FusedConv, 'pair_mode', -1)
    f = FusedConv(w, bn=bn, stride=st, padding=1, relu=True, dims=3).to('cuda')
    sc = bn[0] / torch.sqrt(bn[3] + 1e-5)
    tref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.double(), stride=st, padding=1).permute(0, 2, 3, 4, 1)
    tref = tref * sc.double() + (bn[1] - bn[2] * sc).double()
    res = torch.randn(tref.shape, generator=g) if use_res else None
    if use_res:
        tref = tref + res.double()
    tref = tref.clamp_min(0)
    rng = float(tref.abs().max())
    out = {}
    for opnd in (4, 0):
        monkeypatch.setattr(FusedConv, 'wino_operands', opnd)
        assert f.wino_tile(tuple(x.shape))[0] == 0
        assert f.takes_pair_form(tuple(x.shape)) == (opnd == 4)
        FusedConv.trace = []
        y = f(x.cuda(), res.cuda() if use_res else None)
        torch.cuda.synchronize()
        kinds, FusedConv.trace = [t[0] for t in FusedConv.trace], None
        assert kinds == (['pair_split', 'pair_gemm'] if opnd == 4 else ['direct'])
        assert_close('FusedConv (%s) vs torch fp64' % ('split-operand form' if opnd else 'fp32 MFMA'), y, tref.float(), 0, (2e-5 if opnd else 2e-6) * rng)
        out[opnd] = y
    assert not torch.equal(out[4], out[0])
    # what the rule leaves alone: 1x1x1 layers, narrow outputs (the Cout = 25 head convs), 2-D layers, tensors under SPLIT_MIN_POS positions
    monkeypatch.setattr(FusedConv, 'wino_operands', 4)
    assert not FusedConv(torch.zeros(128, 64, 1, 1, 1), dims=3).to('cuda').takes_pair_form((1, 40, 40, 16, 64))
    assert not FusedConv(torch.zeros(25, 64, 3, 3, 3), padding=1, dims=3).to('cuda').takes_pair_form((1, 40, 40, 16, 64))
    assert not FusedConv(torch.zeros(64, 64, 3, 3), padding=1, dims=2).to('cuda').takes_pair_form((1, 1, 40, 40, 64))
    assert not f.takes_pair_form((1, 5, 5, 4, ci))


def test_conv_pair_refusals(ia):
    from imvoxelnet_amd import ops
    assert not ops.conv_pair_supported((1, 4, 4, 4, 24), 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), 0)      # Cin % 16
    assert not ops.conv_pair_supported((1, 4, 4, 4, 48), 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1)      # chunk-major filters need Cin % 32
    assert not ops.conv_pair_supported((64, 216, 248, 12, 64), 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1)   # operand beyond 31-bit offsets
    with pytest.raises(TypeError):
        ops.conv_fwd(torch.zeros(1, 4, 4, 4, 32, device='cuda'), torch.zeros(8, 3, 3, 3, 32, device='cuda'), kernel=(3, 3, 3), pair=True)


WINO_CASES = [
    # B, (X,Y,Z), Cin, Cout, kw, stride_z, pad, residual, relu, layout
    (2, (9, 14, 5), 16, 12, 3, 1, (1, 1, 1), True, True, 0),       # odd X, Cin 16 (tap-major only)
    (1, (12, 10, 6), 32, 64, 3, 2, (1, 1, 1), False, True, 1),     # z stride 2, chunk-major K
    (2, (10, 12, 3), 64, 32, 3, 1, (0, 0, 0), False, False, 1),    # padding 0: z 3 -> 1
    (3, (31, 17, 2), 128, 128, 3, 1, (1, 1, 1), True, True, 1),    # odd X and Y, 128 channels
    (1, (13, 13, 1), 256, 256, 1, 1, (1, 1, 0), True, True, 1),    # the 2-D 3x3 layers' view: z kernel 1
]


@pytest.mark.parametrize('case', WINO_CASES)
@pytest.mark.parametrize('tile', [4, 6])
def test_conv_winograd_f16_pair_operands(ia, case, tile):
    """ivx_conv_winograd_fwd with wino_operands = IVX_F16_PAIR (V and U as fp16 (hi, lo) pairs with device-side power-of-two scales, three
    fp16 MFMA products per pair) against torch fp64, next to the fp32-operand form on the same inputs: same tolerance as the fp32 form's
    own test, and an error no larger than the fp32 form's (plus rounding noise)."""
    from imvoxelnet_amd import ops
    B, (X, Y, Z), ci, co, kw, sz, pad, use_res, relu, layout = case
    g = torch.Generator().manual_seed(X * 131 + ci + tile)
    x = torch.randn(B, X, Y, Z, ci, generator=g).cuda()
    w = (torch.randn(co, 3, 3, kw, ci, generator=g) * (2.0 / (9 * kw * ci)) ** 0.5).cuda()
    scale = (torch.rand(co, generator=g) + 0.5).cuda()
    shift = (torch.randn(co, generator=g) * 0.1).cuda()
    tref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).cpu().double(), w.permute(0, 4, 1, 2, 3).cpu().double(),
                                      stride=(1, 1, sz), padding=pad).permute(0, 2, 3, 4, 1)
    tref = tref * scale.cpu().double() + shift.cpu().double()
    res = torch.randn(tref.shape, generator=g).cuda() if use_res else None
    if use_res:
        tref = tref + res.cpu().double()
    if relu:
        tref = tref.clamp_min(0)
    P = ops.IVX_F16_PAIR
    assert ops.conv_winograd_supported(tuple(x.shape), co, (3, 3, kw), (1, 1, sz), pad, tile, operands=P)
    up = ops.conv_winograd_weights(w, layout, tile, operands=P)
    assert up.shape[0] == (tile + 2) ** 2 + 1                     # one more plane: the filter scale travels with the filters
    got = ops.conv_winograd_fwd(x, up, scale, shift, kw, sz, pad, relu, res, wgt_layout=layout, operands=P)
    u32 = ops.conv_winograd_weights(w, layout, tile)
    g32 = ops.conv_winograd_fwd(x, u32, scale, shift, kw, sz, pad, relu, res, wgt_layout=layout)
    rng = float(tref.abs().max())
    assert_close('f16-pair winograd vs torch fp64', got, tref.float(), 1e-4, 1e-4 * rng)
    e_pair = float((got.cpu().double() - tref).pow(2).mean().sqrt())
    e_f32 = float((g32.cpu().double() - tref).pow(2).mean().sqrt())
    print(f'rms error / range: pair {e_pair / rng:.2e}  fp32 operands {e_f32 / rng:.2e}')
    assert e_pair <= 1.5 * e_f32 + 2e-7 * rng
    # the scales come from the data: any power-of-two rescaling of the input gives the rescaled output bit for bit, down to
    # activation scales where a fixed fp16 scale would lose the lo halves (measured: 1.8e-4 rms at 1e-3 with a fixed scale)
    a = ops.conv_winograd_fwd(x, up, None, None, kw, sz, pad, False, wgt_layout=layout, operands=P)
    for k in (-12, 9):
        b = ops.conv_winograd_fwd(x * 2.0 ** k, up, None, None, kw, sz, pad, False, wgt_layout=layout, operands=P)
        assert torch.equal(b, a * 2.0 ** k), k
    z = ops.conv_winograd_fwd(torch.zeros_like(x), up, None, shift, kw, sz, pad, False, wgt_layout=layout, operands=P)
    assert torch.equal(z, shift.expand_as(z).contiguous())       # max |input| = 0: scale 1, no NaN
    # staged entry points = the one-shot call
    ops.winograd_trace = []
    try:
        got2 = ops.conv_winograd_fwd(x, up, scale, shift, kw, sz, pad, relu, res, wgt_layout=layout, operands=P)
    finally:
        ops.winograd_trace = None
    assert torch.equal(got, got2)


def test_conv_winograd_f16_pair_range(ia):
    """fp16's range: an outlier 1e4 times the typical activation stays finite and accurate to the outlier's own rounding; tiny filters
    (rms 1e-5) and large ones (rms 30) are carried by the filter scale; operands the form cannot take are refused."""
    from imvoxelnet_amd import ops
    P = ops.IVX_F16_PAIR
    g = torch.Generator().manual_seed(12)
    x = torch.randn(1, 24, 24, 4, 64, generator=g).abs_()
    x[0, 11, 13, 2, 5] = 2.0e4
    for wscale in (1.0, 1e-5 / 0.034, 30.0 / 0.034):
        w = torch.randn(64, 3, 3, 3, 64, generator=g) * (2.0 / (27 * 64)) ** 0.5 * wscale
        tref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.permute(0, 4, 1, 2, 3).double(), padding=1).permute(0, 2, 3, 4, 1)
        up = ops.conv_winograd_weights(w.cuda(), 1, 6, operands=P)
        got = ops.conv_winograd_fwd(x.cuda(), up, None, None, 3, 1, (1, 1, 1), False, wgt_layout=1, operands=P)
        u32 = ops.conv_winograd_weights(w.cuda(), 1, 6)
        g32 = ops.conv_winograd_fwd(x.cuda(), u32, None, None, 3, 1, (1, 1, 1), False, wgt_layout=1)
        assert torch.isfinite(got).all()
        rng = float(tref.abs().max())
        e_pair, e_f32 = float((got.cpu().double() - tref).abs().max()), float((g32.cpu().double() - tref).abs().max())
        print(f'filter scale {wscale:.1e}: max err / range pair {e_pair / rng:.2e} fp32 {e_f32 / rng:.2e}')
        assert e_pair <= 3e-5 * rng and e_pair <= 3.0 * e_f32 + 1e-6 * rng
    assert not ops.conv_winograd_supported((1, 12, 12, 4, 24), 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), 6, operands=P)    # Cin % 16
    assert not ops.conv_winograd_supported((1, 12, 12, 4, 64), 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), 2, operands=P)    # tile 2


@pytest.mark.parametrize('shape', [((216, 248, 12), 64, 64, (1, 1, 1), (1, 1, 1)), ((216, 248, 6), 128, 256, (1, 1, 2), (1, 1, 1)),
                                   ((216, 248, 3), 256, 256, (1, 1, 1), (0, 0, 0))])
def test_conv_winograd_f16_pair_fullsize_kitti_layers(ia, shape):
    """BASELINE config 2 sizes, batch 4: F(6x6,3x3) on fp16-pair operands against the direct fp32 MFMA kernel (1e-4 of the output range, the
    fp32 form's own bound) and exact linearity of the three-stage pipeline under power-of-two rescaling."""
    from imvoxelnet_amd import ops
    P = ops.IVX_F16_PAIR
    (X, Y, Z), ci, co, st, pad = shape
    g = torch.Generator(device='cuda').manual_seed(ci + Z)
    x = torch.randn(4, X, Y, Z, ci, device='cuda', generator=g).clamp_min_(0)
    w = torch.randn(co, 3, 3, 3, ci, device='cuda', generator=g) * (2.0 / (27 * ci)) ** 0.5
    sc = torch.rand(co, device='cuda', generator=g) + 0.5
    sh = torch.randn(co, device='cuda', generator=g) * 0.1
    ref = ops.conv_fwd(x, w, sc, sh, (3, 3, 3), st, pad, relu=True)
    u = ops.conv_winograd_weights(w, 0, 6, operands=P)
    got = ops.conv_winograd_fwd(x, u, sc, sh, 3, st[2], pad, True, operands=P)
    rng, err = float(ref.abs().max()), float((got - ref).abs().max())
    print(f'{ci}->{co} z{Z}: max|diff| {err:.3e} of range {rng:.3e}')
    assert got.shape == ref.shape and err <= 1e-4 * rng
    a = ops.conv_winograd_fwd(x, u, None, None, 3, st[2], pad, False, operands=P)
    x.mul_(2.0)
    b = ops.conv_winograd_fwd(x, u, None, None, 3, st[2], pad, False, operands=P)
    assert torch.equal(b, a * 2.0)


def test_native_handle_operand_modes_give_the_same_detections(ia):
    """The model handle with wino_operands = IVX_F16_PAIR (the default of the Python host) against the same handle on fp32 MFMA: identical
    kept labels and counts, scores within 1e-4, boxes within 1e-3 (north-star: identical indices after NMS)."""
    from imvoxelnet_amd import engine
    from imvoxelnet_amd.conv import FusedConv
    from imvoxelnet_amd.workloads import kitti_model_cfg, KITTI_TEST_CFG, kitti_meta
    model = ia.build_detector(kitti_model_cfg(n_voxels=(104, 120, 12)), test_cfg=KITTI_TEST_CFG)
    ia.randomize_(model, 21)
    with torch.no_grad():
        model.bbox_head.conv_cls.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(5))
        model.bbox_head.conv_cls.bias.fill_(-1.0)
        model.bbox_head.conv_reg.weight.normal_(0, 0.002, generator=torch.Generator().manual_seed(6))
    old = FusedConv.wino_operands
    outs = {}
    try:
        for mode in (4, 0):
            FusedConv.wino_operands = mode
            model.prepare(torch.device('cuda'))
            assert model._native is not None
            img = torch.randn(2, 1, 3, 192, 640, generator=torch.Generator().manual_seed(3)).cuda()
            metas = [kitti_meta(img_hw=(192, 640), t=(0.0, 0.01 * b, 0.0), box_type=ia.LiDARInstance3DBoxes) for b in range(2)]
            outs[mode] = model.simple_test(img, metas)
    finally:
        FusedConv.wino_operands = old
    total = 0
    for a, b in zip(outs[4], outs[0]):
        assert torch.equal(a['labels_3d'], b['labels_3d']) and len(a['scores_3d']) == len(b['scores_3d'])
        assert_close('scores', a['scores_3d'], b['scores_3d'], 0, 1e-4)
        assert_close('boxes', a['boxes_3d'].tensor, b['boxes_3d'].tensor, 0, 1e-3)
        total += len(a['scores_3d'])
    assert total > 10


@pytest.mark.parametrize('tile', [4, 6])
def test_winograd_chained_amax_partials(ia, tile):
    """ivx_conv_winograd_output_amax leaves one maximum per workgroup of the STORED tensor (after scale / shift / residual / ReLU, border
    positions excluded): their maximum is max |out| exactly, and a consumer fed with them (ivx_conv_winograd_input_amax) produces the same
    bits as one that reduces the tensor itself -- for the whole-tile and the buffer-addressed output transform, fp32 and pair operands."""
    from imvoxelnet_amd import ops
    P = ops.IVX_F16_PAIR
    g = torch.Generator().manual_seed(50 + tile)
    x = torch.randn(2, 27, 20, 4, 64, generator=g).cuda()                  # 27 and 20: border tiles for both tile sizes
    w1 = (torch.randn(64, 3, 3, 3, 64, generator=g) * 0.03).cuda()
    w2 = (torch.randn(32, 3, 3, 3, 64, generator=g) * 0.03).cuda()
    sc, sh = (torch.rand(64, generator=g) + 0.5).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()
    for operands in (0, P):
        u1, u2 = ops.conv_winograd_weights(w1, 1, tile, operands), ops.conv_winograd_weights(w2, 1, tile, operands)
        for res in (None, torch.randn(2, 27, 20, 4, 64, generator=g).cuda()):
            y, part = ops.conv_winograd_fwd(x, u1, sc, sh, 3, 1, (1, 1, 1), True, res, wgt_layout=1, operands=operands, want_amax=True)
            y0 = ops.conv_winograd_fwd(x, u1, sc, sh, 3, 1, (1, 1, 1), True, res, wgt_layout=1, operands=operands)
            assert torch.equal(y, y0)
            assert float(part.max()) == float(y.abs().max()) and float(part.min()) >= 0.0
            z = ops.conv_winograd_fwd(y, u2, None, None, 3, 1, (1, 1, 1), False, wgt_layout=1, operands=operands, amax_in=part)
            z0 = ops.conv_winograd_fwd(y, u2, None, None, 3, 1, (1, 1, 1), False, wgt_layout=1, operands=operands)
            assert torch.equal(z, z0), (operands, res is not None)


@pytest.mark.parametrize('shape', [(2, (19, 23, 12), 64, 64), (1, (14, 9, 6), 128, 128), (3, (10, 11, 3), 256, 256), (1, (13, 12, 5), 64, 96)])
def test_winograd_halo_kernel_matches_generic(ia, shape):
    """conv_wino_halo_kernel (stride-1 pad-1 layers on fp16 pairs: one staged tile serves the three z-taps, neighbours outside the column
    zeroed in registers) against the generic LDS-DMA kernel on the same operands: same products, another summation order in the
    transformed domain, which the output transform amplifies (<= 2e-5 of the range; both within 1e-4 of fp64), for the default rule and for every tile config, on volumes whose z extent is not a divisor of the tile rows."""
    from imvoxelnet_amd import _lib, ops
    L = _lib.lib()
    P = ops.IVX_F16_PAIR
    B, (X, Y, Z), ci, co = shape
    g = torch.Generator().manual_seed(ci + Z)
    x = torch.randn(B, X, Y, Z, ci, generator=g).cuda()
    w = (torch.randn(co, 3, 3, 3, ci, generator=g) * (2.0 / (27 * ci)) ** 0.5).cuda()
    u = ops.conv_winograd_weights(w, 1, 6, operands=P)
    try:
        L.ivx_conv_set_halo_mode(0)
        ref = ops.conv_winograd_fwd(x, u, None, None, 3, 1, (1, 1, 1), False, wgt_layout=1, operands=P)
        rng = float(ref.abs().max())
        for mode in (-1, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 13, 14):
            L.ivx_conv_set_halo_mode(mode)
            got = ops.conv_winograd_fwd(x, u, None, None, 3, 1, (1, 1, 1), False, wgt_layout=1, operands=P)
            assert_close(f'halo mode {mode} vs generic', got, ref, 0, 2e-5 * rng)      # measured: max 1e-5, mean 4e-7 of the range
        # round 4: the zero-row forms (30 .. 34) and the z-blocked tiles (50 / 51 on 3-slice columns, 60 on 6-slice ones) accumulate every
        # output element's products in the halo kernel's order: bit-identical to it; a z-blocked config on another column height is refused
        L.ivx_conv_set_halo_mode(13)
        base = ops.conv_winograd_fwd(x, u, None, None, 3, 1, (1, 1, 1), False, wgt_layout=1, operands=P)
        for mode in (30, 31, 33, 34, 50, 51, 60):
            L.ivx_conv_set_halo_mode(mode)
            if (mode in (50, 51) and Z != 3) or (mode == 60 and Z != 6):
                with pytest.raises(ValueError):
                    ops.conv_winograd_fwd(x, u, None, None, 3, 1, (1, 1, 1), False, wgt_layout=1, operands=P)
                continue
            got = ops.conv_winograd_fwd(x, u, None, None, 3, 1, (1, 1, 1), False, wgt_layout=1, operands=P)
            assert torch.equal(got, base), f'halo mode {mode} differs from mode 13'
    finally:
        L.ivx_conv_set_halo_mode(-1)
    tref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).cpu().double(), w.permute(0, 4, 1, 2, 3).cpu().double(), padding=1).permute(0, 2, 3, 4, 1)
    assert_close('halo kernel vs torch fp64', got.cpu(), tref.float(), 1e-4, 1e-4 * float(tref.abs().max()))


@pytest.mark.parametrize('shape', [(2, (19, 23, 12), 64, 128), (1, (14, 9, 6), 128, 256), (1, (9, 7, 4), 32, 64), (2, (7, 9, 2), 64, 64)])
def test_winograd_halo_kernel_stride2_matches_generic(ia, shape):
    """The z-stride-2 form of conv_wino_halo_kernel (even Z: input row 2 r' - 1 + kz for output row r', 2 BM staged rows, only tap 0 can
    leave the column) against the generic kernel, default rule and every config, and against torch fp64."""
    from imvoxelnet_amd import _lib, ops
    L = _lib.lib()
    P = ops.IVX_F16_PAIR
    B, (X, Y, Z), ci, co = shape
    g = torch.Generator().manual_seed(ci + 3 * Z)
    x = torch.randn(B, X, Y, Z, ci, generator=g).cuda()
    w = (torch.randn(co, 3, 3, 3, ci, generator=g) * (2.0 / (27 * ci)) ** 0.5).cuda()
    u = ops.conv_winograd_weights(w, 1, 6, operands=P)
    try:
        L.ivx_conv_set_halo_mode(0)
        ref = ops.conv_winograd_fwd(x, u, None, None, 3, 2, (1, 1, 1), False, wgt_layout=1, operands=P)
        assert ref.shape[3] == Z // 2
        rng = float(ref.abs().max())
        for mode in (-1, 21, 22, 23, 24):
            L.ivx_conv_set_halo_mode(mode)
            got = ops.conv_winograd_fwd(x, u, None, None, 3, 2, (1, 1, 1), False, wgt_layout=1, operands=P)
            assert_close(f'halo mode {mode} vs generic', got, ref, 0, 2e-5 * rng)
        # round 4: zero-row (41, 42), de-interleaved staging (43 .. 45) and the z-blocked tile for 6 -> 3 slices (71): bit-identical to mode 22
        L.ivx_conv_set_halo_mode(22)
        base = ops.conv_winograd_fwd(x, u, None, None, 3, 2, (1, 1, 1), False, wgt_layout=1, operands=P)
        for mode in (41, 42, 43, 44, 45, 71):
            L.ivx_conv_set_halo_mode(mode)
            if mode == 71 and Z != 6:
                with pytest.raises(ValueError):
                    ops.conv_winograd_fwd(x, u, None, None, 3, 2, (1, 1, 1), False, wgt_layout=1, operands=P)
                continue
            got = ops.conv_winograd_fwd(x, u, None, None, 3, 2, (1, 1, 1), False, wgt_layout=1, operands=P)
            assert torch.equal(got, base), f'halo mode {mode} differs from mode 22'
    finally:
        L.ivx_conv_set_halo_mode(-1)
    tref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).cpu().double(), w.permute(0, 4, 1, 2, 3).cpu().double(), stride=(1, 1, 2),
                                      padding=1).permute(0, 2, 3, 4, 1)
    assert_close('stride-2 halo kernel vs torch fp64', got.cpu(), tref.float(), 1e-4, 1e-4 * float(tref.abs().max()))


FUSED_CASES = [
    # B, (X,Y,Z), Cin, Cout, residual, relu   (F(4x4,3x3), stride-1 pad-1 z kernel, chunk-major filters)
    (2, (19, 23, 12), 64, 64, False, True),      # odd X and Y: border tiles in both directions; 12 slices as KITTI level 0
    (1, (16, 12, 6), 64, 64, True, True),        # residual
    (1, (9, 14, 5), 32, 128, True, False),       # Cin 32 (two pair groups), two N tiles, no ReLU
    (3, (12, 8, 3), 128, 96, False, True),       # Cout not a multiple of 64
    (1, (40, 40, 16), 128, 128, True, True),     # an indoor-neck shape (tile 4 is that family's default)
    (1, (5, 3, 2), 64, 64, False, False),        # fewer rows than one row tile
]


@pytest.mark.parametrize('case', FUSED_CASES)
def test_conv_winograd_fused_gemm_output(ia, case):
    """Round 5: ivx_conv_winograd_gemm_output_amax (conv_wino_fold4_kernel: the 36 Winograd-domain GEMMs of an F(4x4,3x3) layer with the
    output transform and the epilogue fused, M kept on chip) against the three-stage form on the same V and U -- the same products summed
    in another order: 2e-5 of the output range -- and against torch fp64 with an error no larger than the three-stage form's; the
    per-workgroup maxima are those of the stored tensor, and a consumer fed with them gives the same bits as one that reduces the tensor."""
    from imvoxelnet_amd import ops
    P = ops.IVX_F16_PAIR
    B, (X, Y, Z), ci, co, use_res, relu = case
    g = torch.Generator().manual_seed(X * 17 + ci + co)
    x = torch.randn(B, X, Y, Z, ci, generator=g).cuda()
    w = (torch.randn(co, 3, 3, 3, ci, generator=g) * (2.0 / (27 * ci)) ** 0.5).cuda()
    scale, shift = (torch.rand(co, generator=g) + 0.5).cuda(), (torch.randn(co, generator=g) * 0.1).cuda()
    res = torch.randn(B, X, Y, Z, co, generator=g).cuda() if use_res else None
    tref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).cpu().double(), w.permute(0, 4, 1, 2, 3).cpu().double(), padding=1).permute(0, 2, 3, 4, 1)
    tref = tref * scale.cpu().double() + shift.cpu().double()
    if use_res:
        tref = tref + res.cpu().double()
    if relu:
        tref = tref.clamp_min(0)
    u = ops.conv_winograd_weights(w, 1, 4, operands=P)
    want = ops.conv_winograd_fwd(x, u, scale, shift, 3, 1, (1, 1, 1), relu, res, wgt_layout=1, operands=P)
    got, part = ops.conv_winograd_fwd(x, u, scale, shift, 3, 1, (1, 1, 1), relu, res, wgt_layout=1, operands=P, fused=True, want_amax=True)
    rng = float(tref.abs().max())
    assert got.shape == want.shape
    assert_close('fused GEMM + output vs three stages', got, want, 0, 2e-5 * rng)
    e_f, e_3 = float((got.cpu().double() - tref).pow(2).mean().sqrt()), float((want.cpu().double() - tref).pow(2).mean().sqrt())
    print(f'rms error / range: fused {e_f / rng:.2e}  three-stage {e_3 / rng:.2e}')
    assert e_f <= 1.5 * e_3 + 2e-7 * rng
    assert float(part.max()) == float(got.abs().max()) and float(part.min()) >= 0.0
    # power-of-two rescaling of the input: exact (the operand scale comes from max |input|, the fold is linear)
    a = ops.conv_winograd_fwd(x, u, None, None, 3, 1, (1, 1, 1), False, wgt_layout=1, operands=P, fused=True)
    b = ops.conv_winograd_fwd(x * 2.0 ** -7, u, None, None, 3, 1, (1, 1, 1), False, wgt_layout=1, operands=P, fused=True)
    assert torch.equal(b, a * 2.0 ** -7)
    # chained: the consumer's operand scale from this layer's partial maxima == from the tensor
    w2 = (torch.randn(32, 3, 3, 3, co, generator=g) * 0.03).cuda() if co % 32 == 0 else None
    if w2 is not None:
        u2 = ops.conv_winograd_weights(w2, 1, 4, operands=P)
        z = ops.conv_winograd_fwd(got, u2, None, None, 3, 1, (1, 1, 1), False, wgt_layout=1, operands=P, amax_in=part)
        z0 = ops.conv_winograd_fwd(got, u2, None, None, 3, 1, (1, 1, 1), False, wgt_layout=1, operands=P)
        assert torch.equal(z, z0)


def test_conv_winograd_fused_refusals(ia):
    from imvoxelnet_amd import ops, _lib
    import ctypes as C
    L = _lib.lib()
    P = ops.IVX_F16_PAIR
    d = ops._wino_desc(1, 12, 12, 4, 64, 64, 3, 1, (1, 1, 1), False, 1, operands=P)
    assert L.ivx_conv_winograd_fused_supported(C.byref(d), 4) == 1
    assert L.ivx_conv_winograd_fused_supported(C.byref(d), 6) == 0                                                             # F(4x4) only
    assert L.ivx_conv_winograd_fused_supported(C.byref(ops._wino_desc(1, 12, 12, 4, 64, 64, 3, 2, (1, 1, 1), False, 1, operands=P)), 4) == 0   # z stride 2
    assert L.ivx_conv_winograd_fused_supported(C.byref(ops._wino_desc(1, 12, 12, 3, 64, 64, 3, 1, (1, 1, 0), False, 1, operands=P)), 4) == 0   # z padding 0
    assert L.ivx_conv_winograd_fused_supported(C.byref(ops._wino_desc(1, 12, 12, 4, 64, 64, 3, 1, (1, 1, 1), False, 0, operands=P)), 4) == 0   # tap-major filters
    assert L.ivx_conv_winograd_fused_supported(C.byref(ops._wino_desc(1, 12, 12, 4, 64, 64, 3, 1, (1, 1, 1), False, 1, operands=0)), 4) == 0   # fp32 operands
    assert L.ivx_conv_winograd_fused_supported(C.byref(ops._wino_desc(8, 216, 248, 12, 64, 64, 3, 1, (1, 1, 1), False, 1, operands=P)), 4) == 0   # 36 planes of V >= 2 GiB
