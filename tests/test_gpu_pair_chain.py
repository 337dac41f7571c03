"""-m gpu: chained fp16-pair activations (include/imvoxel.h ivx_pair_io / ivx_conv_fwd_pio, ivx_model_cfg.trunk_operands): the 2-D trunk on
the 16-bit matrix cores without conversion passes.  Op level: the pair-IO epilogue (fp32 / pair output, fp32 / pair / nearest-upsampled
residual, split-K reduction) against the validation kernel and torch fp64, the device-side scale rule and the recorded maxima.  Model
level: ResNet-50 + FPN in the chained form against the fp32-MFMA form and the oracle, both hosts bit-identical, and the full-size KITTI
A/B of the two operand modes with identical kept anchor indices.  Runs on the MI355X box."""
import ctypes as C
import math

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


def _pow2_scale(amax):
    if not (amax > 0 and amax < 3e38):
        return 1.0
    _, e = math.frexp(float(np.float32(amax)))
    return 2.0 ** (15 - e)


def make_pair(x):
    """fp32 device tensor [.., C] -> ops.PairTensor with the scale the device rule gives its exact maximum and the slots filled as a
    producer would have (test helper: the product never converts an fp32 tensor -- its producers write pairs)."""
    from imvoxelnet_amd import _lib, ops
    amax = float(x.abs().max())
    s = _pow2_scale(amax)
    data = torch.empty(x.shape[:-1] + (2 * x.shape[-1],), device=x.device, dtype=torch.float16)
    _lib.check(_lib.lib().ivx_f16_pair_split(C.c_void_p(x.data_ptr()), x.numel(), C.c_float(s), C.c_void_p(data.data_ptr()),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'ivx_f16_pair_split')
    slots = ops.new_slots(x.device)
    slots[:ops.AMAX_SLOTS].view(torch.float32)[7] = amax          # any slot: the readers take the maximum over all of them
    slots[ops.AMAX_SLOTS:ops.AMAX_SLOTS + 1].view(torch.float32)[0] = s
    return ops.PairTensor(data, slots)


def test_pair_tensor_roundtrip(ia):
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(2, 1, 9, 7, 64, generator=g) * torch.logspace(-3, 2, 64)).cuda()
    p = make_pair(x)
    assert p.shape == (2, 1, 9, 7, 64) and p.data.dtype == torch.float16
    back = p.float()
    # 22 significant bits for values within 2^-18 of the maximum, an absolute floor of 2^-25 / s below
    err = (back - x).abs()
    assert float((err - (x.abs() * 2.0 ** -21 + 2.0 ** -24 / p.scale())).max()) <= 0
    assert p.amax() == float(x.abs().max())


PIO_CASES = [
    # B, (H,W), Cin, Cout, k, stride, pad, residual ('', 'f32', 'pair', 'up_f32', 'up_pair'), relu, out_pair
    (2, (24, 40), 64, 256, 1, 1, 0, 'pair', True, True),        # bottleneck conv3 + identity
    (2, (24, 40), 256, 64, 1, 1, 0, '', True, True),            # conv1
    (1, (30, 44), 128, 128, 3, 2, 1, '', True, True),           # strided conv2
    (2, (17, 23), 64, 64, 3, 1, 1, '', True, False),            # FPN output conv: fp32 out
    (1, (12, 40), 2048, 512, 1, 1, 0, '', True, True),          # few rows, long K: split-K + reduction kernel
    (4, (12, 40), 512, 2048, 1, 1, 0, 'f32', True, True),       # conv3 + shortcut conv output (fp32 residual)
    (2, (24, 40), 512, 64, 1, 1, 0, 'up_f32', False, True),     # FPN lateral + nearest-upsampled coarser level
    (2, (24, 40), 256, 64, 1, 1, 0, 'up_pair', False, False),
    (3, (48, 160), 64, 64, 3, 1, 1, '', True, True),            # many tiles: main launch + K-split tail
]


@pytest.mark.parametrize('case', PIO_CASES)
def test_conv_pio_vs_fp64(ia, case):
    """ivx_conv_fwd_pio: the MFMA kernel against the validation kernel on the same pair operands and against torch fp64 on the values
    the pairs stand for; the chosen scale is the power of two the bound rule gives, never overflows, and the slots hold max |out|."""
    from imvoxelnet_amd import ops
    B, (H, W), ci, co, k, st, pad, res_kind, relu, out_pair = case
    g = torch.Generator().manual_seed(ci * 3 + co + k)
    x = (torch.randn(B, 1, H, W, ci, generator=g).abs_() * 3.0).cuda()
    w = torch.randn(co, ci, 1, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
    scale, shift = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1
    xp = make_pair(x)
    xv = xp.float()                                               # the values the kernel sees
    wt = w.permute(0, 2, 3, 4, 1).reshape(co, k * k, ci).contiguous()
    packed, sp, wb, sb = ops.pair_pack_filters(wt, scale, shift)
    # the filters as the pair form holds them
    pw = packed.float().reshape(co, ci // 32, k * k, 2, 2, 16)
    w_eff = (pw[..., 0, :] + pw[..., 1, :]).reshape(co, ci // 32, k * k, 32).permute(0, 2, 1, 3).reshape(co, k * k, ci)
    s_w = float(scale[0] / sp[0])
    assert math.log2(s_w) == int(math.log2(s_w)) and torch.equal(sp * s_w, scale)
    assert float((w_eff / s_w - wt).abs().max()) <= float(wt.abs().max()) * 2.0 ** -21
    ref = torch.nn.functional.conv2d(xv[:, 0].permute(0, 3, 1, 2).double().cpu(),
                                     (w_eff / s_w).reshape(co, k, k, ci).permute(0, 3, 1, 2).double(), stride=st, padding=pad)
    ref = ref.permute(0, 2, 3, 1).unsqueeze(1) * scale.double() + shift.double()
    res, res_mode = None, 0
    if res_kind:
        shp = tuple(ref.shape) if not res_kind.startswith('up') else (B, 1, ref.shape[2] // 2, ref.shape[3] // 2, co)
        rf = (torch.randn(shp, generator=g) * 2.0).cuda()
        res = make_pair(rf) if res_kind.endswith('pair') else rf
        if not isinstance(res, ops.PairTensor):
            res.ivx_slots = ops.new_slots('cuda')
            res.ivx_slots[:ops.AMAX_SLOTS].view(torch.float32)[3] = float(rf.abs().max())
        rv = (res.float() if isinstance(res, ops.PairTensor) else rf).double().cpu()
        if res_kind.startswith('up'):
            res_mode = 2
            rv = rv.repeat_interleave(2, 2).repeat_interleave(2, 3)
        ref = ref + rv
    if relu:
        ref = ref.clamp_min(0)
    args = (xp, packed.cuda(), sp.cuda(), shift.cuda(), (1, k, k), (1, st, st), (0, pad, pad), relu, wb, sb)
    got = ops.conv_fwd_pio(*args, res=res, res_mode=res_mode, out_pair=out_pair)
    naive = ops.conv_fwd_pio(*args, res=res, res_mode=res_mode, out_pair=out_pair, naive=True)
    rng = float(ref.abs().max())
    if out_pair:
        assert isinstance(got, ops.PairTensor) and isinstance(naive, ops.PairTensor)
        s = got.scale()
        a_in = float(xv.abs().max())
        a_res = float(res.float().abs().max() if isinstance(res, ops.PairTensor) else res.abs().max()) if res is not None else 0.0
        assert s == _pow2_scale(float(np.float32(np.float32(a_in) * np.float32(wb) + np.float32(sb) + np.float32(a_res)) * np.float32(1.001))) or \
            abs(math.log2(s) - math.log2(_pow2_scale((a_in * wb + sb + a_res) * 1.001))) <= 1          # (fp32 vs double at a power-of-two boundary)
        assert s == naive.scale()
        assert bool(torch.isfinite(got.data.float()).all()) and float(got.data.float().abs().max()) < 2.0 ** 15 * 1.01
        gv, nv = got.float(), naive.float()
        slots = got.slots
    else:
        assert isinstance(got, torch.Tensor) and got.dtype == torch.float32
        gv, nv = got, naive
        slots = got.ivx_slots
    assert gv.shape == ref.shape
    assert_close('pair-IO MFMA kernel vs validation kernel', gv, nv, 0, 1e-5 * rng)
    assert_close('pair-IO MFMA kernel vs torch fp64', gv, ref.float().cuda(), 0, 2e-5 * rng)
    # the recorded maximum is that of the values before the output split (22-bit rounding apart for a pair output)
    rec = float(slots[:ops.AMAX_SLOTS].view(torch.float32).max())
    assert abs(rec - float(gv.abs().max())) <= rec * 2.0 ** -20


DEEP_CASES = [
    # B, (H,W), Cin, Cout, k, stride, residual, out_pair -- the /8 .. /32 layer shapes of the KITTI trunk (240 - 480 tiles) and short / odd K loops
    (4, (24, 80), 1024, 256, 1, 1, '', True), (4, (24, 80), 256, 256, 3, 1, '', True), (4, (24, 80), 256, 1024, 1, 1, 'pair', True),
    (4, (48, 160), 256, 256, 3, 2, '', True), (4, (12, 40), 2048, 512, 1, 1, '', True), (2, (24, 40), 512, 64, 1, 1, 'up_f32', False),
    (1, (9, 13), 64, 64, 1, 1, '', True), (1, (9, 13), 96, 64, 3, 1, '', False), (2, (17, 23), 160, 128, 1, 1, 'f32', True),
]


@pytest.mark.parametrize('case', DEEP_CASES)
def test_deep_ring_tiles_are_bit_identical(ia, case):
    """Round 5: the deep-ring forms of the pair tiles (conv_igemm_v4_kernel<.., NB = 4>: 166 = 64 x 64, 174 = 128 x 128; A/B configs) issue
    the same products in the same order as the two-buffer tiles 66 / 74 -- only the DMA of slab s + NB is issued earlier -- so outputs,
    scales and recorded maxima are equal bit for bit, for K loops shorter than the ring, odd slab counts and every residual form.  Likewise the
    128 x 128 tile on 8 / 16 waves (177 / 179: the rule's choice up to two / one tile per CU) and with the residual requested in front of the K
    loop (475)."""
    from imvoxelnet_amd import _lib, ops
    L = _lib.lib()
    B, (H, W), ci, co, k, st, res_kind, out_pair = case
    g = torch.Generator().manual_seed(ci + 7 * co + k)
    x = (torch.randn(B, 1, H, W, ci, generator=g).abs_() * 2.0).cuda()
    w = torch.randn(co, ci, 1, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
    scale, shift = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1
    xp = make_pair(x)
    packed, sp, wb, sb = ops.pair_pack_filters(w.permute(0, 2, 3, 4, 1).reshape(co, k * k, ci).contiguous(), scale, shift)
    Ho, Wo = (H + 2 * (k // 2) - k) // st + 1, (W + 2 * (k // 2) - k) // st + 1
    res, res_mode = None, 0
    if res_kind:
        shp = (B, 1, Ho, Wo, co) if not res_kind.startswith('up') else (B, 1, Ho // 2, Wo // 2, co)
        rf = (torch.randn(shp, generator=g) * 2.0).cuda()
        res = make_pair(rf) if res_kind.endswith('pair') else rf
        if not isinstance(res, ops.PairTensor):
            res.ivx_slots = ops.new_slots('cuda')
            res.ivx_slots[:ops.AMAX_SLOTS].view(torch.float32)[3] = float(rf.abs().max())
        res_mode = 2 if res_kind.startswith('up') else 0
    args = (xp, packed.cuda(), sp.cuda(), shift.cuda(), (1, k, k), (1, st, st), (0, k // 2, k // 2), True, wb, sb)

    def run(cfg):
        L.ivx_conv_set_tile_override(cfg)
        try:
            y = ops.conv_fwd_pio(*args, res=res, res_mode=res_mode, out_pair=out_pair)
            torch.cuda.synchronize()
            return y
        finally:
            L.ivx_conv_set_tile_override(0)

    for base, deep in ((66, (166,)), (74, (174, 177, 179, 475)), (82, (183,))):      # 183: the last neck layer's GEMM tile on a three-buffer ring
        want = run(base)
        for cfg in deep:
            got = run(cfg)
            if out_pair:
                assert got.scale() == want.scale() and torch.equal(got.data, want.data), (base, cfg)
                if cfg in (177, 179):      # other workgroup shapes: the maxima land in other slots, the recorded maximum is the same
                    assert int(got.slots.max()) == int(want.slots.max()), (base, cfg)
                else:
                    assert torch.equal(got.slots, want.slots), (base, cfg)
            else:
                assert torch.equal(got, want) and torch.equal(got.ivx_slots, want.ivx_slots), (base, cfg)
    want = run(74)
    auto = run(0)                      # the library's own choice: same bits as tile 66 / 74 ...
    gv = auto.float() if out_pair else auto
    wv = want.float() if out_pair else want
    assert_close('default plan vs tile 74', gv, wv, 0, 1e-5 * float(wv.abs().max()))      # ... or to accumulation order when it splits K


def test_maxpool_pair_and_image_amax(ia):
    """ivx_nchw_to_nhwc_amax records max |image|; ivx_maxpool2d_fwd_pair == the fp32 pool, written as pairs with the scale of the bound."""
    from imvoxelnet_amd import ops
    g = torch.Generator().manual_seed(5)
    img = (torch.randn(3, 3, 64, 96, generator=g) * 2.5).cuda()
    cl = ops.to_channels_last_amax(img, pad_to=4)
    assert torch.equal(cl, ops.to_channels_last(img, pad_to=4))
    assert float(cl.ivx_slots[:ops.AMAX_SLOTS].view(torch.float32).max()) == float(img.abs().max())
    x = torch.randn(3, 1, 32, 48, 64, generator=g).cuda().relu_()
    wb, sb = 7.5, 0.25
    p = ops.maxpool2d_pair(x, cl.ivx_slots, wb, sb)
    ref = ops.maxpool2d(x)
    assert p.shape == tuple(ref.shape)
    assert p.scale() == _pow2_scale(float(np.float32(np.float32(float(img.abs().max())) * np.float32(wb) + np.float32(sb)) * np.float32(1.001)))
    assert float((p.float() - ref).abs().max()) <= float(ref.abs().max()) * 2.0 ** -21 + 2.0 ** -24 / p.scale()
    assert p.amax() == float(ref.abs().max())


def _kitti_like_model(ia, seed=3, dcn=False):
    from imvoxelnet_amd.workloads import kitti_model_cfg, KITTI_TEST_CFG
    nv = (24, 28, 12)
    cfg = kitti_model_cfg(n_voxels=nv, in_ch=64, out_ch=64)
    ox = 0.5 + nv[0] * .32 / 2
    cfg['bbox_head']['anchor_generator']['ranges'] = [[ox - nv[0] * .16, -nv[1] * .16, -1.78, ox + nv[0] * .16 - .32, nv[1] * .16 - .32, -1.78]]
    model = ia.build_detector(cfg, test_cfg=dict(KITTI_TEST_CFG, score_thr=0.05))
    ia.randomize_(model, seed)
    return model, ox


def test_trunk_pair_chain_vs_fp32_mfma(ia, monkeypatch):
    """ResNet-50 + FPN level 0 through the layer-by-layer host: chained fp16-pair activations against fp32 MFMA (the two settings of
    FusedConv.trunk_operands).  The pair form keeps 22 bits per operand: the maps agree to 2e-5 of their range."""
    from imvoxelnet_amd.conv import FusedConv
    from imvoxelnet_amd import ops
    model, _ = _kitti_like_model(ia)
    img = torch.randn(2, 1, 3, 128, 224, generator=torch.Generator().manual_seed(11)).cuda()
    outs = {}
    for mode in (0, 4):
        monkeypatch.setattr(FusedConv, 'trunk_operands', mode)
        model.prepare('cuda', native=False)
        assert bool(getattr(model.backbone, 'chain', False)) == (mode == 4)
        FusedConv.trace = []
        outs[mode] = model.features_2d_cl(img)
        kinds = [r[-1] for r in FusedConv.trace]
        FusedConv.trace = None
        n_pair = sum(k.endswith(' pair') for k in kinds)
        n_fused = sum(k.startswith('bottleneck ') and k.endswith(' pair, one launch') for k in kinds)
        n_stem = sum(k.startswith('stem ') and k.endswith(' pair, one launch') for k in kinds)
        # every bottleneck conv, shortcut conv, FPN lateral and the FPN output conv run on pairs.  The five identity blocks of stages 1 and 2
        # are one launch each (ops.bottleneck_fwd_pio) instead of three, the first block of stage 1 one (ops.bottleneck_proj_fwd_pio: its shortcut
        # conv included) instead of four, and so is the head (layout change + stem + max-pool: ops.stem_pool_pair)
        assert (n_fused, n_stem) == ((0, 0) if mode == 0 else (6, 1)), (mode, n_fused, n_stem)
        assert n_pair == (0 if mode == 0 else 3 * (16 - n_fused) + 3 + 4 + 1), (mode, n_pair, len(kinds))
    torch.cuda.synchronize()
    rng = float(outs[0].abs().max())
    assert_close('FPN level 0: pair chain vs fp32 MFMA', outs[4], outs[0], 0, 2e-5 * rng)


@pytest.mark.parametrize('family', ['kitti', 'nuscenes_dcn'])
def test_trunk_pair_chain_native_equals_layerwise(ia, family):
    """Both hosts of the chain -- csrc/model.cpp (ivx_backbone_fpn_fwd) and the Python composition -- decide the tensor formats by the
    same rule and call the same kernels: bit-identical FPN maps, DCNv2 stages (fp32 islands inside the chain) included."""
    from imvoxelnet_amd import engine
    from imvoxelnet_amd.conv import FusedConv
    assert FusedConv.trunk_operands == 4, 'the default is the chained pair form'
    if family == 'kitti':
        model, _ = _kitti_like_model(ia)
    else:
        from imvoxelnet_amd.workloads import nuscenes_model_cfg, NUSCENES_TEST_CFG
        cfg = nuscenes_model_cfg(n_voxels=(24, 24, 12), dcn=True)
        model = ia.build_detector(cfg, test_cfg=NUSCENES_TEST_CFG)
        ia.randomize_(model, 5)
    model.prepare('cuda')
    assert model._native is not None
    img = torch.randn(1, 2, 3, 96, 160, generator=torch.Generator().manual_seed(2)).cuda()
    lay = model.features_2d_cl(img)
    nat = model._native.backbone_fpn(img.reshape(2, 3, 96, 160))
    torch.cuda.synchronize()
    assert torch.equal(lay, nat), float((lay - nat).abs().max())


def test_kitti_fullsize_operand_modes_identical_kept_anchor_indices(ia, monkeypatch):
    """BASELINE config 2 at full size (4 x 3x384x1280 -> 216x248x12): the default operand modes (fp16-pair chain in the 2-D trunk,
    fp16-pair operands in the Winograd-domain neck GEMMs) against fp32 MFMA everywhere -- the anchor indices of the kept boxes of
    every sample are identical and in identical order (north star: identical indices after NMS); scores within 1e-4."""
    from gpu_util import match_rows, assert_same_kept
    from imvoxelnet_amd.conv import FusedConv
    from imvoxelnet_amd import workloads as kc
    model = ia.build_detector(kc.kitti_model_cfg(), test_cfg=kc.KITTI_TEST_CFG)
    ia.randomize_(model, 123)
    with torch.no_grad():
        model.bbox_head.conv_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(5))
        model.bbox_head.conv_cls.bias.fill_(-2.0)
        model.bbox_head.conv_reg.weight.normal_(0, 0.002, generator=torch.Generator().manual_seed(6))
        model.bbox_head.conv_dir_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(7))
    B = 4
    metas = [kc.kitti_meta(t=(0.02 * b, 0.01 * b, 0.0), box_type=ia.LiDARInstance3DBoxes) for b in range(B)]
    img = torch.randn(B, 1, 3, 384, 1280, generator=torch.Generator().manual_seed(21)).cuda()
    res = {}
    for name, (trunk, wino) in {'default': (4, 4), 'fp32': (0, 0)}.items():
        monkeypatch.setattr(FusedConv, 'trunk_operands', trunk)
        monkeypatch.setattr(FusedConv, 'wino_operands', wino)
        model.prepare(torch.device('cuda'))
        vol, valid = model.lift_cl(model.features_2d_cl(img), metas)
        boxes, scores, labels, count, (ci, cb, cs) = model.detect_cl(vol, metas, want_candidates=True)
        out = model.simple_test(img, metas)                 # the native handle in the same mode
        kept = []
        for b in range(B):
            n = int(count[b])
            ids = ci[b].cpu()[match_rows(torch.cat([boxes[b, :n, :6], scores[b, :n, None]], 1), torch.cat([cb[b, :, :6], cs[b, :, None]], 1))]
            kept.append((ids.numpy(), scores[b, :n].cpu().numpy()))
            assert torch.equal(out[b]['scores_3d'], scores[b, :n].cpu()) and torch.equal(out[b]['boxes_3d'].tensor, boxes[b, :n].cpu())
        res[name] = (kept, vol, valid)
        del vol, boxes
    assert torch.equal(res['default'][2], res['fp32'][2])
    rng = float(res['fp32'][1].abs().max())
    assert_close('volume: default operand modes vs fp32 MFMA', res['default'][1], res['fp32'][1], 0, 2e-5 * rng)
    total = 0
    for b in range(B):
        (gi, gs), (ri, rs) = res['default'][0][b], res['fp32'][0][b]
        assert_same_kept(f'kitti full size, sample {b}: default vs fp32 operands', gi, gs, ri, rs)
        assert np.allclose(gs, rs, rtol=1e-4, atol=1e-6)
        total += len(gi)
    assert total > 20


def test_nonfinite_input_stays_local_in_pair_operands(ia, monkeypatch):
    """The advisor's round-3 finding: a pair tensor's scale comes from a maximum (measured, or a bound built from measured maxima), and ONE
    Inf / NaN element makes that maximum useless.  Rule (winograd.hip wino_pow2_scale / conv_igemm.hip conv_pair_io): a non-finite maximum
    selects the fixed scale 2^-8 and a saturating split, so the damage stays where fp32 arithmetic keeps it -- the outputs whose receptive
    field holds the element -- instead of every value of the tensor overflowing.  Checked on (a) a Winograd F(6x6,3x3) layer with fp16
    pair operands and (b) the chained stem + max-pool of the 2-D trunk: far from the element the results equal the clean run's."""
    from imvoxelnet_amd.conv import FusedConv
    from imvoxelnet_amd import ops
    g = torch.Generator().manual_seed(17)
    # (a) 3-D layer, Winograd domain, pair operands
    monkeypatch.setattr(FusedConv, 'winograd', True)
    monkeypatch.setattr(FusedConv, 'wino_operands', 4)
    monkeypatch.setattr(FusedConv, 'winograd_min_pos', 0)
    w = torch.randn(64, 64, 3, 3, 3, generator=g) * (2.0 / (64 * 27)) ** 0.5
    fc = FusedConv(w, padding=1, relu=False, dims=3).to('cuda')      # (no ReLU: the kernels' `v > 0 ? v : 0` turns a NaN into 0)
    x = torch.randn(1, 48, 54, 6, 64, generator=g).cuda().relu_()
    clean = fc(x)
    for bad in (float('inf'), float('nan')):
        xb = x.clone()
        xb[0, 20, 25, 3, 7] = bad
        y = fc(xb)
        torch.cuda.synchronize()
        far = torch.ones(48, 54, dtype=torch.bool, device='cuda')
        far[20 - 9:20 + 10, 25 - 9:25 + 10] = False           # every 8x8 input tile that can hold the element, plus the 3x3 reach
        yf, cf = y[0][far], clean[0][far]
        assert bool(torch.isfinite(yf).all()), 'a non-finite input element spread beyond its tiles'
        assert float((yf - cf).abs().max()) <= 1e-4 * float(clean.abs().max())
        assert not bool(torch.isfinite(y[0, 20, 25]).all())    # ... and it is still visible where fp32 arithmetic shows it
    # (b) the chained trunk: image -> layout change (records max |image| = Inf) -> stem conv (bound = Inf: fixed scale, saturating) -> pool
    monkeypatch.setattr(FusedConv, 'trunk_operands', 4)
    model, _ = _kitti_like_model(ia)
    model.prepare('cuda', native=False)
    bb = model.backbone
    assert bb.chain
    img = torch.randn(1, 3, 128, 224, generator=g).cuda()
    ib = img.clone()
    ib[0, 1, 60, 100] = float('inf')

    def stem_pool(im):      # backbones.ResNet._stages: fp32 stem -> pair max-pool scaled by the bound max |image| * wbound + sbound
        x0 = ops.to_channels_last_amax(im.contiguous(), pad_to=4)
        return ops.maxpool2d_pair(bb.stem(x0), ops.slots_of(x0), bb.stem.wbound, bb.stem.sbound, 3, 2, 1)
    a, b = stem_pool(img), stem_pool(ib)
    assert isinstance(a, ops.PairTensor) and isinstance(b, ops.PairTensor) and b.scale() == 2.0 ** -8
    av, bv = a.float(), b.float()
    far = torch.ones(av.shape[2], av.shape[3], dtype=torch.bool, device='cuda')
    far[15 - 3:15 + 4, 25 - 3:25 + 4] = False          # stem output (30, 50) +- 2 sees pixel (60, 100); pooled (15, 25) +- 2 sees those
    assert bool(torch.isfinite(bv[0, 0][far]).all())
    assert float((bv[0, 0][far] - av[0, 0][far]).abs().max()) <= 1e-4 * float(av.abs().max())


def test_dcn_columns_in_the_pair_chain(ia):
    """ivx_dcn_im2col_fwd_pair: the DCNv2 columns of a pair map are the fp32 columns of the decoded map rounded to the pair format (the
    columns keep the map's scale: a convex combination times a mask in (0, 1) cannot exceed max |x|), their recorded maximum is exact, and
    the contraction over them (FusedConv on the 9 C columns, pair filters) agrees with the fp32 path to 22 bits."""
    from imvoxelnet_amd import ops
    from imvoxelnet_amd.conv import FusedConv
    g = torch.Generator().manual_seed(23)
    B, H, W, Cn = 2, 29, 50, 64
    x = (torch.randn(B, 1, H, W, Cn, generator=g) * 1.7).cuda().relu_()
    for stride in (1, 2):
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        om = torch.randn(B, 1, Ho, Wo, 28, generator=g).cuda()
        om[..., :18] *= 2.5                                     # offsets of a few pixels, some beyond the border
        xp = make_pair(x)
        xd = xp.float()                                         # what the pair map holds
        ref = ops.dcn_im2col(xd, om, 3, stride, 1, 1)
        got = ops.dcn_im2col_pair(xp, om, 3, stride, 1, 1)
        assert isinstance(got, ops.PairTensor) and got.shape == tuple(ref.shape) and got.scale() == xp.scale()
        gv = got.float()
        assert float((gv - ref).abs().max()) <= float(ref.abs().max()) * 2.0 ** -21 + 2.0 ** -24 / got.scale()
        assert got.amax() == float(ref.abs().max())
        # contraction: 1x1 over K = 9 C, pair filters against fp32 MFMA on the fp32 columns
        w = torch.randn(64, 9 * Cn, 1, 1, generator=g) * (2.0 / (9 * Cn)) ** 0.5
        bn = (torch.rand(64, generator=g) + .5, torch.randn(64, generator=g) * .1, torch.randn(64, generator=g) * .1, torch.rand(64, generator=g) + .5)
        f_pair = FusedConv(w, bn=bn, relu=True, dims=2, chain=True).to('cuda')
        f_f32 = FusedConv(w, bn=bn, relu=True, dims=2).to('cuda')
        assert f_pair.pair_ok
        y_pair = f_pair(got)
        y_f32 = f_f32(ref)
        assert_close('DCNv2 contraction on pair columns vs fp32 MFMA on fp32 columns', y_pair, y_f32, 0, 2e-5 * float(y_f32.abs().max()))
