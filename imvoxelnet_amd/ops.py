"""Tensor-level wrappers over the C-ABI.  torch tensors are containers only: every wrapper
checks device / dtype / contiguity (as CHECK_INPUT does for the reference's native op,
mmdet3d/ops/iou3d/src/iou3d.cpp:17-23), allocates the output with torch.empty and passes raw
pointers plus the current HIP stream to libimvoxel_hip.so.

Internal activation layout is channels-last: 5-D [B, D, H, W, C] (a 2-D map has D == 1).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, AnchorHeadDesc, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


FP8 = torch.float8_e4m3fn      # OCP e4m3 bytes with a per-tensor scale kept by the caller (conv.QTensor)
_DT = {torch.float32: 0, torch.bfloat16: 1, FP8: 2}   # IVX_F32 / IVX_BF16 / IVX_FP8


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name} must be a torch.Tensor')
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a device (HIP) tensor; the MI355X path has no CPU fallback')
    if t.dtype != dtype:
        raise TypeError(f'{name} must be {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name} must be contiguous')
    return t


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# ------------------------------------------------------------------ layout
def image_s2d_bf16(img):
    """bf16 mode: image [N,3,H,W] fp32 -> 2x2 space-to-depth blocks [N,1,H/2+1,W/2+1,16] bf16 (ivx_image_s2d_bf16): the input of
    the stem in its 4x4 stride-1 form (backbones.ResNet)."""
    _chk(img, 'img')
    N, Cn, H, W = img.shape
    if Cn != 3:
        raise ValueError('image_s2d_bf16 takes a 3-channel image')
    out = torch.empty((N, 1, H // 2 + 1, W // 2 + 1, 16), device=img.device, dtype=torch.bfloat16)
    check(_lib.lib().ivx_image_s2d_bf16(_ptr(img), N, H, W, _ptr(out), _stream()), 'ivx_image_s2d_bf16')
    return out


def to_channels_last(x, pad_to=None):
    """[B,C,*spatial] (reference layout) -> [B,D,H,W,Cpad] channels-last (D=1 for 2-D input)."""
    _chk(x, 'x')
    B, Cn = x.shape[0], x.shape[1]
    sp = list(x.shape[2:])
    if len(sp) == 2:
        sp = [1] + sp
    S = sp[0] * sp[1] * sp[2]
    Cp = Cn if pad_to is None else ((Cn + pad_to - 1) // pad_to) * pad_to
    out = torch.empty([B] + sp + [Cp], device=x.device, dtype=torch.float32)
    check(_lib.lib().ivx_nchw_to_nhwc(_ptr(x), B, Cn, S, Cp, _ptr(out), _stream()), 'ivx_nchw_to_nhwc')
    return out


def from_channels_last(x, ndim_spatial=3):
    """[B,D,H,W,C] -> [B,C,D,H,W] (or [B,C,H,W] when ndim_spatial == 2); the reference layout is always fp32."""
    if x.dtype == torch.bfloat16:
        x = x.float()
    _chk(x, 'x')
    B, D, H, W, Cn = x.shape
    S = D * H * W
    shape = [B, Cn, D, H, W] if ndim_spatial == 3 else [B, Cn, H, W]
    if ndim_spatial == 2 and D != 1:
        raise ValueError('2-D output requested but D != 1')
    out = torch.empty(shape, device=x.device, dtype=torch.float32)
    check(_lib.lib().ivx_nhwc_to_nchw(_ptr(x), B, S, Cn, _ptr(out), _stream()), 'ivx_nhwc_to_nchw')
    return out


# ------------------------------------------------------------------ conv
def conv_fwd(x, wgt, scale=None, shift=None, kernel=(1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0), relu=False,
             res=None, res_mode=0, naive=False, out=None, wgt_layout=0, out_mode=0, res_after_act=False, post_scale=1.0,
             out_dtype=None, res_scale=1.0, pair=False):
    """x [B,D,H,W,Cin], wgt [Cout,KD,KH,KW,Cin] (packed), scale/shift [Cout] -> [B,Do,Ho,Wo,Cout].
    x and wgt are fp32 (the reference's precision), both bf16, or both e4m3 bytes (torch.float8_e4m3fn; the caller folds the
    tensors' scales into scale / shift / res_scale, see ivx_conv_desc); out_dtype (default: x.dtype) is the storage type of
    the output and of the residual.  Accumulation and the epilogue are fp32 in every case.
    pair=True: x and wgt are IVX_BF16_PAIR operands (bf16_pair_split / conv.pack_pair_weights: bf16 tensors with 2*Cin stored
    elements per voxel / tap); the output defaults to fp32."""
    if pair:
        if x.dtype != torch.bfloat16 or x.shape[4] % 32:
            raise TypeError('pair=True takes the bf16 tensor made by bf16_pair_split (2 * Cin elements per voxel, Cin % 16 == 0)')
        out_dtype = torch.float32 if out_dtype is None else out_dtype
    if x.dtype not in _DT:
        raise TypeError(f'x must be float32, bfloat16 or float8_e4m3fn, got {x.dtype}')
    out_dtype = x.dtype if out_dtype is None else out_dtype
    if out_dtype not in _DT:
        raise TypeError(f'out_dtype must be float32, bfloat16 or float8_e4m3fn, got {out_dtype}')
    _chk(x, 'x', x.dtype)
    _chk(wgt, 'wgt', x.dtype)
    B, D, H, W, Cin = x.shape
    Cout = wgt.shape[0]
    ck = {torch.float32: 32, torch.bfloat16: 64, FP8: 128}[x.dtype]
    want = (kernel[0], kernel[1], kernel[2], Cin) if wgt_layout == 0 else (Cin // ck, kernel[0], kernel[1], kernel[2], ck)
    if tuple(wgt.shape[1:]) != want:
        raise ValueError(f'weight shape {tuple(wgt.shape)} does not match kernel {kernel} / Cin {Cin} / layout {wgt_layout}')
    d = ConvDesc(B, D, H, W, Cin // 2 if pair else Cin, Cout, kernel[0], kernel[1], kernel[2], stride[0], stride[1], stride[2],
                 padding[0], padding[1], padding[2], int(bool(relu)), 0, 0, 0, int(wgt_layout), int(out_mode), int(bool(res_after_act)), float(post_scale),
                 IVX_BF16_PAIR if pair else _DT[x.dtype], _DT[out_dtype], float(res_scale))
    do, ho, wo = C.c_int32(), C.c_int32(), C.c_int32()
    L = _lib.lib()
    check(L.ivx_conv_out_dims(C.byref(d), C.byref(do), C.byref(ho), C.byref(wo)), 'ivx_conv_out_dims')
    oshape = (B, do.value, ho.value, wo.value, Cout)
    if out_mode == 1:
        oshape = (B, 2 * D, 2 * H, 2 * W, Cout // 8)
    if res is not None:
        _chk(res, 'res', out_dtype)
        if res_mode == 0:
            res_mode = 1
        if res_mode == 1 and tuple(res.shape) != oshape:
            raise ValueError(f'residual shape {tuple(res.shape)} != output shape')
        if res_mode == 2:
            if res.shape[0] != B or res.shape[1] != 1 or res.shape[4] != Cout:
                raise ValueError('res_mode 2 residual must be [B,1,h,w,Cout]')
            d.res_h, d.res_w = res.shape[2], res.shape[3]
        d.res_mode = res_mode
    for t, n in ((scale, 'scale'), (shift, 'shift')):
        if t is not None:
            _chk(t, n)
            if t.numel() != oshape[-1]:
                raise ValueError(f'{n} must have {oshape[-1]} elements')
    if out is None:
        out = torch.empty(oshape, device=x.device, dtype=out_dtype)
    else:
        _chk(out, 'out', out_dtype)
    if naive:
        check(L.ivx_conv_fwd_naive(C.byref(d), _ptr(x), _ptr(wgt), _ptr(scale), _ptr(shift), _ptr(res), _ptr(out), _stream()),
              'ivx_conv_fwd_naive')
        return out
    wsb = L.ivx_conv_workspace_bytes(C.byref(d))
    ws = torch.empty((wsb,), device=x.device, dtype=torch.uint8) if wsb > 0 else None
    check(L.ivx_conv_fwd_ws(C.byref(d), _ptr(x), _ptr(wgt), _ptr(scale), _ptr(shift), _ptr(res), _ptr(out), _ptr(ws), max(wsb, 0),
                            _stream()), 'ivx_conv_fwd_ws')
    return out


IVX_BF16_PAIR = 3


def bf16_pair_split(x, out=None):
    """fp32 [..., C] (C % 16 == 0, contiguous) -> the IVX_BF16_PAIR operand: bf16 [..., 2C], per 16 channels [hi x16 | lo x16] with
    hi = bf16(x), lo = bf16(x - hi)  (ivx_bf16_pair_split)."""
    _chk(x, 'x')
    if x.shape[-1] % 16:
        raise ValueError('the channel count must be a multiple of 16')
    if out is None:
        out = torch.empty(x.shape[:-1] + (2 * x.shape[-1],), device=x.device, dtype=torch.bfloat16)
    else:
        _chk(out, 'out', torch.bfloat16)
    check(_lib.lib().ivx_bf16_pair_split(_ptr(x), x.numel(), _ptr(out), _stream()), 'ivx_bf16_pair_split')
    return out


def conv_pair_supported(x_shape, Cout, kernel, stride, padding, wgt_layout=1):
    """True when the fp32 convolution can run in the split-operand (bf16 pair) form: ivx_conv_pair_supported."""
    B, D, H, W, Cin = x_shape
    d = ConvDesc(B, D, H, W, Cin, Cout, kernel[0], kernel[1], kernel[2], stride[0], stride[1], stride[2], padding[0], padding[1], padding[2],
                 0, 0, 0, 0, int(wgt_layout), 0, 0, 1.0, 0, 0, 1.0)
    return bool(_lib.lib().ivx_conv_pair_supported(C.byref(d)))


IVX_F16_PAIR = 4


def _wino_desc(B, D, H, W, Cin, Cout, kw, stride_w, padding, relu, wgt_layout, res_mode=0, res_after_act=False, post_scale=1.0, operands=0):
    return ConvDesc(B, D, H, W, Cin, Cout, 3, 3, kw, 1, 1, stride_w, padding[0], padding[1], padding[2], int(bool(relu)),
                    int(res_mode), 0, 0, int(wgt_layout), 0, int(bool(res_after_act)), float(post_scale), 0, 0, 1.0, int(operands))


def conv_winograd_supported(x_shape, Cout, kernel, stride, padding, tile=2, operands=0):
    """True when ivx_conv_winograd_fwd can run this fp32 convolution (3x3xKW, stride 1 on the first two axes, planes < 2 GiB)."""
    if kernel[0] != 3 or kernel[1] != 3 or stride[0] != 1 or stride[1] != 1 or x_shape[4] % 4 or Cout % 4 or tile not in (2, 4, 6):
        return False
    B, D, H, W, Cin = x_shape
    d = _wino_desc(B, D, H, W, Cin, Cout, kernel[2], stride[2], padding, False, 0, operands=operands)
    return bool(_lib.lib().ivx_conv_winograd_supported(C.byref(d), tile))


def conv_winograd_weights(wgt, wgt_layout, tile=2, operands=0):
    """wgt [Cout,3,3,KW,Cin] fp32 (layout 0, on the device) -> transformed filters u [(tile+2)^2, Cout, KW*Cin] whose K order
    is `wgt_layout` (0: tap-major, 1: 32-channel chunks).  operands = IVX_F16_PAIR: the same bytes hold fp16 (hi, lo) pairs
    (ivx_conv_desc.wino_operands); pass the same value to conv_winograd_fwd."""
    _chk(wgt, 'wgt')
    Cout, kd, kh, kw, Cin = wgt.shape
    if kd != 3 or kh != 3:
        raise ValueError('Winograd F(m x m, 3x3) needs a 3x3 kernel on the first two axes')
    d = _wino_desc(1, 4, 4, max(kw, 1), Cin, Cout, kw, 1, (1, 1, kw // 2), False, wgt_layout, operands=operands)
    L = _lib.lib()
    n = L.ivx_conv_winograd_weight_elems(C.byref(d), tile)
    if n < 0:
        check(-1, 'ivx_conv_winograd_weight_elems')
    # pair operands: one more plane (it carries the filter scale); the same bytes then hold fp16 pairs
    u = torch.empty(((tile + 2) ** 2 + (1 if operands else 0), Cout, kw * Cin), device=wgt.device, dtype=torch.float32)
    assert u.numel() == n
    check(L.ivx_conv_winograd_weights(C.byref(d), tile, _ptr(wgt), _ptr(u), _stream()), 'ivx_conv_winograd_weights')
    return u


# optional stage timing of the Winograd path (bench.py): list of (stage, start_event, end_event, executed_flops)
winograd_trace = None


def conv_winograd_fwd(x, u, scale=None, shift=None, kw=3, stride_w=1, padding=(1, 1, 1), relu=False, res=None, out=None,
                      wgt_layout=0, res_after_act=False, post_scale=1.0, operands=0, amax_in=None, want_amax=False, fused=False):
    """Same result as conv_fwd for a 3x3xkw kernel with stride (1,1,stride_w), computed in the F(m x m, 3x3) minimal-filtering
    form (fp32).  x [B,D,H,W,Cin]; u from conv_winograd_weights (its first dimension, 16 / 36 / 64, selects m = 2 / 4 / 6)."""
    _chk(x, 'x')
    _chk(u, 'u')
    B, D, H, W, Cin = x.shape
    Cout = u.shape[1]
    tile = {16: 2, 36: 4, 64: 6}.get(u.shape[0] - (1 if operands else 0))
    if tile is None or tuple(u.shape[1:]) != (Cout, kw * Cin):
        raise ValueError(f'transformed filters {tuple(u.shape)} do not match kw {kw} / Cin {Cin}')
    d = _wino_desc(B, D, H, W, Cin, Cout, kw, stride_w, padding, relu, wgt_layout, 1 if res is not None else 0, res_after_act,
                   post_scale, operands)
    do, ho, wo = C.c_int32(), C.c_int32(), C.c_int32()
    L = _lib.lib()
    check(L.ivx_conv_out_dims(C.byref(d), C.byref(do), C.byref(ho), C.byref(wo)), 'ivx_conv_out_dims')
    oshape = (B, do.value, ho.value, wo.value, Cout)
    if res is not None:
        _chk(res, 'res')
        if tuple(res.shape) != oshape:
            raise ValueError(f'residual shape {tuple(res.shape)} != output shape')
    for t, n in ((scale, 'scale'), (shift, 'shift')):
        if t is not None:
            _chk(t, n)
            if t.numel() != Cout:
                raise ValueError(f'{n} must have {Cout} elements')
    if out is None:
        out = torch.empty(oshape, device=x.device, dtype=torch.float32)
    else:
        _chk(out, 'out')
    wsb = L.ivx_conv_winograd_workspace_bytes(C.byref(d), tile)
    if wsb < 0:
        check(-1, 'ivx_conv_winograd_workspace_bytes')
    ws = torch.empty((wsb,), device=x.device, dtype=torch.uint8)
    if fused:
        # F(4x4,3x3) pair operands: GEMM + output transform in one launch, M on chip (ivx_conv_winograd_gemm_output_amax)
        if not L.ivx_conv_winograd_fused_supported(C.byref(d), tile):
            raise ValueError('this layer does not take the fused GEMM + output form (ivx_conv_winograd_fused_supported)')
        part = None
        if want_amax:
            part = torch.empty((L.ivx_conv_winograd_fused_blocks(C.byref(d), tile),), device=x.device, dtype=torch.float32)
        if amax_in is not None:
            _chk(amax_in, 'amax_in')
        check(L.ivx_conv_winograd_input_amax(C.byref(d), tile, _ptr(x), _ptr(ws), wsb, _ptr(amax_in), 0 if amax_in is None else amax_in.numel(),
                                             _stream()), 'ivx_conv_winograd_input_amax')
        check(L.ivx_conv_winograd_gemm_output_amax(C.byref(d), tile, _ptr(u), _ptr(scale), _ptr(shift), _ptr(res), _ptr(out), _ptr(ws), wsb, _ptr(part),
                                                   _stream()), 'ivx_conv_winograd_gemm_output_amax')
        return (out, part) if want_amax else out
    if amax_in is not None or want_amax:
        # chained layers (ivx_conv_winograd_output_amax / _input_amax): amax_in = the producer's per-workgroup maxima of x;
        # want_amax: also return this layer's own -> (out, partials)
        part = None
        if want_amax:
            nb = L.ivx_conv_winograd_output_blocks(C.byref(d), tile)
            if nb < 0:
                check(-1, 'ivx_conv_winograd_output_blocks')
            part = torch.empty((nb,), device=x.device, dtype=torch.float32)
        if amax_in is not None:
            _chk(amax_in, 'amax_in')
        check(L.ivx_conv_winograd_input_amax(C.byref(d), tile, _ptr(x), _ptr(ws), wsb, _ptr(amax_in), 0 if amax_in is None else amax_in.numel(),
                                             _stream()), 'ivx_conv_winograd_input_amax')
        check(L.ivx_conv_winograd_gemm(C.byref(d), tile, _ptr(u), _ptr(ws), wsb, _stream()), 'ivx_conv_winograd_gemm')
        check(L.ivx_conv_winograd_output_amax(C.byref(d), tile, _ptr(scale), _ptr(shift), _ptr(res), _ptr(out), _ptr(ws), wsb, _ptr(part),
                                              _stream()), 'ivx_conv_winograd_output_amax')
        return (out, part) if want_amax else out
    if winograd_trace is None:
        check(L.ivx_conv_winograd_fwd(C.byref(d), tile, _ptr(x), _ptr(u), _ptr(scale), _ptr(shift), _ptr(res), _ptr(out), _ptr(ws),
                                      wsb, _stream()), 'ivx_conv_winograd_fwd')
        return out
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    check(L.ivx_conv_winograd_input(C.byref(d), tile, _ptr(x), _ptr(ws), wsb, _stream()), 'ivx_conv_winograd_input')
    ev[1].record()
    check(L.ivx_conv_winograd_gemm(C.byref(d), tile, _ptr(u), _ptr(ws), wsb, _stream()), 'ivx_conv_winograd_gemm')
    ev[2].record()
    check(L.ivx_conv_winograd_output(C.byref(d), tile, _ptr(scale), _ptr(shift), _ptr(res), _ptr(out), _ptr(ws), wsb, _stream()),
          'ivx_conv_winograd_output')
    ev[3].record()
    tiles = B * ((oshape[1] + tile - 1) // tile) * ((oshape[2] + tile - 1) // tile)
    winograd_trace.append(('input', ev[0], ev[1], 0.0))
    frac = float(L.ivx_conv_winograd_issued_fraction(C.byref(d)))      # the z-blocked tile skips the taps outside a 3-slice column
    winograd_trace.append(('gemm', ev[1], ev[2], (3.0 if operands else 1.0) * frac * 2.0 * (tile + 2) ** 2 * tiles * oshape[3] * Cout * kw * Cin))
    winograd_trace.append(('output', ev[2], ev[3], 0.0))
    return out


class WinogradLayerPlan:
    """One fp32 3x3xkw layer in the F(m x m, 3x3) form for a fixed input shape: descriptor, output shape and workspace
    size, with the three stages as separate calls on caller-owned buffers (tools/gemm_ab.py times them one by one)."""

    def __init__(self, x_shape, Cout, kw, stride_w, padding, relu, wgt_layout, tile, has_res=False, res_after_act=False,
                 post_scale=1.0, operands=0):
        B, D, H, W, Cin = x_shape
        self.tile = int(tile)
        self.d = _wino_desc(B, D, H, W, Cin, Cout, kw, stride_w, padding, relu, wgt_layout, 1 if has_res else 0, res_after_act,
                            post_scale, operands)
        do, ho, wo = C.c_int32(), C.c_int32(), C.c_int32()
        L = _lib.lib()
        check(L.ivx_conv_out_dims(C.byref(self.d), C.byref(do), C.byref(ho), C.byref(wo)), 'ivx_conv_out_dims')
        self.x_shape = tuple(x_shape)
        self.oshape = (B, do.value, ho.value, wo.value, Cout)
        self.ws_bytes = L.ivx_conv_winograd_workspace_bytes(C.byref(self.d), self.tile)
        if self.ws_bytes < 0:
            check(-1, 'ivx_conv_winograd_workspace_bytes')
        n2 = (self.tile + 2) ** 2
        tiles = B * ((self.oshape[1] + tile - 1) // tile) * ((self.oshape[2] + tile - 1) // tile)
        self.gemm_flops = 2.0 * n2 * tiles * self.oshape[3] * Cout * kw * Cin
        self.v_bytes = 4.0 * n2 * tiles * W * Cin
        self.m_bytes = 4.0 * n2 * tiles * self.oshape[3] * Cout

    def input(self, x, ws):
        check(_lib.lib().ivx_conv_winograd_input(C.byref(self.d), self.tile, _ptr(x), _ptr(ws), ws.numel(), _stream()),
              'ivx_conv_winograd_input')

    def gemm(self, u, ws):
        check(_lib.lib().ivx_conv_winograd_gemm(C.byref(self.d), self.tile, _ptr(u), _ptr(ws), ws.numel(), _stream()),
              'ivx_conv_winograd_gemm')

    def output(self, scale, shift, res, out, ws):
        check(_lib.lib().ivx_conv_winograd_output(C.byref(self.d), self.tile, _ptr(scale), _ptr(shift), _ptr(res), _ptr(out), _ptr(ws),
                                                  ws.numel(), _stream()), 'ivx_conv_winograd_output')


def maxpool2d(x, k=3, s=2, p=1):
    if x.dtype not in _DT:
        raise TypeError(f'x must be float32, bfloat16 or float8_e4m3fn, got {x.dtype}')
    _chk(x, 'x', x.dtype)
    B, D, H, W, Cn = x.shape
    if D != 1:
        raise ValueError('maxpool2d expects a 2-D map (D == 1)')
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    out = torch.empty((B, 1, Ho, Wo, Cn), device=x.device, dtype=x.dtype)
    L = _lib.lib()
    fn = {torch.float32: L.ivx_maxpool2d_fwd, torch.bfloat16: L.ivx_maxpool2d_fwd_bf16, FP8: L.ivx_maxpool2d_fwd_fp8}[x.dtype]
    check(fn(_ptr(x), B, H, W, Cn, k, s, p, _ptr(out), _stream()), 'ivx_maxpool2d_fwd')
    return out


# ------------------------------------------------------------------ chained fp16-pair activations (include/imvoxel.h, ivx_pair_io)
AMAX_SLOTS = 64


def new_slots(device):
    """Scalar block of one tensor of a pair chain: AMAX_SLOTS words of max |tensor| (float bits, accumulated by the producing kernel
    with atomic max: they start at zero) followed by the tensor's power-of-two scale."""
    return torch.zeros(AMAX_SLOTS + 16, device=device, dtype=torch.int32)


def _scale_ptr(slots):
    return C.c_void_p(slots.data_ptr() + 4 * AMAX_SLOTS)


class PairTensor:
    """An IVX_F16_PAIR activation: data = float16 [B,D,H,W,2C] (per 16 channels [hi x16 | lo x16]) of s * x, with the scalar block
    (amax slots, scale s) its producer filled on the device.  `shape` is the logical fp32 shape [B,D,H,W,C]."""
    __slots__ = ('data', 'slots')

    def __init__(self, data, slots):
        self.data, self.slots = data, slots

    shape = property(lambda self: tuple(self.data.shape[:4]) + (self.data.shape[4] // 2,))
    device = property(lambda self: self.data.device)
    dtype = property(lambda self: torch.float32)       # what the values are; the storage is fp16 pairs

    def numel(self):
        return self.data.numel() // 2

    def element_size(self):
        return 4

    def float(self):
        """fp32 values (hi + lo) / s (ivx_f16_pair_merge)."""
        out = torch.empty(self.shape, device=self.data.device, dtype=torch.float32)
        check(_lib.lib().ivx_f16_pair_merge(_ptr(self.data), out.numel(), _scale_ptr(self.slots), _ptr(out), _stream()), 'ivx_f16_pair_merge')
        return out

    def amax(self):
        """max |x| as the producer recorded it (host float; synchronises)."""
        return float(self.slots[:AMAX_SLOTS].view(torch.float32).max())

    def scale(self):
        return float(self.slots[AMAX_SLOTS:AMAX_SLOTS + 1].view(torch.float32)[0])


def pair_from_float(x):
    """fp32 device tensor [.., C] (C % 16 == 0) -> PairTensor, scaled by its exact maximum (one pass over the tensor + a host sync:
    for tools and tests; inside the model the producing kernels write pairs)."""
    import math
    _chk(x, 'x')
    amax = float(x.abs().max())
    s = 1.0
    if 0.0 < amax < 3e38:
        s = 2.0 ** (15 - math.frexp(amax)[1])
    data = torch.empty(tuple(x.shape[:-1]) + (2 * x.shape[-1],), device=x.device, dtype=torch.float16)
    check(_lib.lib().ivx_f16_pair_split(_ptr(x), x.numel(), C.c_float(s), _ptr(data), _stream()), 'ivx_f16_pair_split')
    slots = new_slots(x.device)
    slots[:1].view(torch.float32)[0] = amax
    slots[AMAX_SLOTS:AMAX_SLOTS + 1].view(torch.float32)[0] = s
    return PairTensor(data, slots)


def slots_of(t):
    """The scalar block a tensor of the chain carries (a PairTensor's, or the one attached to an fp32 tensor by its producer), or None."""
    return t.slots if isinstance(t, PairTensor) else getattr(t, 'ivx_slots', None)


def to_channels_last_amax(x, pad_to=None):
    """to_channels_last that also records max |x| in a scalar block attached to the result (out.ivx_slots): ivx_nchw_to_nhwc_amax."""
    _chk(x, 'x')
    B, Cn = x.shape[0], x.shape[1]
    sp = list(x.shape[2:])
    if len(sp) == 2:
        sp = [1] + sp
    S = sp[0] * sp[1] * sp[2]
    Cp = Cn if pad_to is None else ((Cn + pad_to - 1) // pad_to) * pad_to
    out = torch.empty([B] + sp + [Cp], device=x.device, dtype=torch.float32)
    slots = new_slots(x.device)
    check(_lib.lib().ivx_nchw_to_nhwc_amax(_ptr(x), B, Cn, S, Cp, _ptr(out), _ptr(slots), _stream()), 'ivx_nchw_to_nhwc_amax')
    out.ivx_slots = slots
    return out


def maxpool2d_pair(x, amax_in, wbound, sbound, k=3, s=2, p=1):
    """nn.MaxPool2d on an fp32 map -> PairTensor scaled by the bound amax_in * wbound + sbound of the input map (ivx_maxpool2d_fwd_pair);
    amax_in: the scalar block of the tensor the bound refers to (the image)."""
    _chk(x, 'x')
    B, D, H, W, Cn = x.shape
    if D != 1 or Cn % 16:
        raise ValueError('maxpool2d_pair expects a 2-D map with C % 16 == 0')
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    out = torch.empty((B, 1, Ho, Wo, 2 * Cn), device=x.device, dtype=torch.float16)
    slots = new_slots(x.device)
    check(_lib.lib().ivx_maxpool2d_fwd_pair(_ptr(x), B, H, W, Cn, k, s, p, _ptr(out), _ptr(amax_in), float(wbound), float(sbound),
                                            _scale_ptr(slots), _ptr(slots), _stream()), 'ivx_maxpool2d_fwd_pair')
    return PairTensor(out, slots)


def pair_pack_filters(w_tap, scale, shift, pack=True):
    """Host: fp32 filters [Cout, taps, Cin] (tap-major, CPU) + the epilogue vectors -> (pair filters float16 [Cout, Cin/32, taps, 64] or None,
    scale / s_w [Cout] or None, wbound, sbound) through ivx_pair_pack_filters (the native handle packs with the same function)."""
    w_tap = w_tap.detach().to(torch.float32).cpu().contiguous()
    co, taps, ci = w_tap.shape
    sc = None if scale is None else scale.detach().to(torch.float32).cpu().contiguous()
    sf = None if shift is None else shift.detach().to(torch.float32).cpu().contiguous()
    packed = torch.empty((co, ci // 32, taps, 64), dtype=torch.float16) if pack else None
    sp = torch.empty(co, dtype=torch.float32) if pack else None
    wb, sb = C.c_float(), C.c_float()
    check(_lib.lib().ivx_pair_pack_filters(_ptr_any(w_tap), co, taps, ci, _ptr_any(sc), _ptr_any(sf), _ptr_any(packed), _ptr_any(sp), C.byref(wb),
                                           C.byref(sb)), 'ivx_pair_pack_filters')
    return packed, sp, wb.value, sb.value


def _ptr_any(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def conv_fwd_pio(x, wgt, scale, shift, kernel, stride, padding, relu, wbound, sbound, res=None, res_mode=0, out_pair=False,
                 res_after_act=False, post_scale=1.0, naive=False):
    """Convolution on a PairTensor (ivx_conv_fwd_pio): wgt = pair filters of pair_pack_filters (device), scale = scale / s_w.
    res: None, an fp32 tensor or a PairTensor.  Returns a PairTensor (out_pair) or an fp32 tensor; either carries the scalar block with
    max |out| (result.slots / result.ivx_slots)."""
    if not isinstance(x, PairTensor):
        raise TypeError('conv_fwd_pio takes a PairTensor')
    _chk(x.data, 'x', torch.float16)
    _chk(wgt, 'wgt', torch.float16)
    B, D, H, W, Cin = x.shape
    Cout = wgt.shape[0]
    if tuple(wgt.shape[1:]) != (Cin // 32, kernel[0] * kernel[1] * kernel[2], 64):
        raise ValueError(f'pair filters {tuple(wgt.shape)} do not match kernel {kernel} / Cin {Cin}')
    d = ConvDesc(B, D, H, W, Cin, Cout, kernel[0], kernel[1], kernel[2], stride[0], stride[1], stride[2], padding[0], padding[1], padding[2],
                 int(bool(relu)), 0, 0, 0, 1, 0, int(bool(res_after_act)), float(post_scale), IVX_F16_PAIR, IVX_F16_PAIR if out_pair else 0, 1.0)
    do, ho, wo = C.c_int32(), C.c_int32(), C.c_int32()
    L = _lib.lib()
    check(L.ivx_conv_out_dims(C.byref(d), C.byref(do), C.byref(ho), C.byref(wo)), 'ivx_conv_out_dims')
    oshape = (B, do.value, ho.value, wo.value, Cout)
    out_pair = bool(out_pair) and Cout % 16 == 0 and B * do.value * ho.value * wo.value * Cout * 4 < 2 ** 31     # (as csrc/model.cpp wants_pair)
    io = _lib.PairIO()
    io.in_scale = _scale_ptr(x.slots)
    io.amax_in = _ptr(x.slots)
    rd = None
    if res is not None:
        res_mode = res_mode or 1
        rshape = tuple(res.shape)
        if res_mode == 1 and rshape != oshape:
            raise ValueError(f'residual shape {rshape} != output shape')
        if res_mode == 2:
            if rshape[0] != B or rshape[1] != 1 or rshape[4] != Cout:
                raise ValueError('res_mode 2 residual must be [B,1,h,w,Cout]')
            d.res_h, d.res_w = rshape[2], rshape[3]
        d.res_mode = res_mode
        if isinstance(res, PairTensor):
            rd = _chk(res.data, 'res', torch.float16)
            io.res_dtype, io.res_scale = IVX_F16_PAIR, _scale_ptr(res.slots)
        else:
            rd = _chk(res, 'res')
        rs = slots_of(res)
        io.amax_res = _ptr(rs)
        if out_pair and rs is None:
            raise ValueError('a pair output needs the scalar block (max |res|) of its residual')
    for t, n in ((scale, 'scale'), (shift, 'shift')):
        if t is not None:
            _chk(t, n)
            if t.numel() != Cout:
                raise ValueError(f'{n} must have {Cout} elements')
    slots = new_slots(x.device)
    io.amax_out = _ptr(slots)
    io.out_scale = _scale_ptr(slots)
    io.wbound, io.sbound = float(wbound), float(sbound)
    if out_pair:
        out = torch.empty(oshape[:4] + (2 * Cout,), device=x.device, dtype=torch.float16)
    else:
        out = torch.empty(oshape, device=x.device, dtype=torch.float32)
    if naive:
        check(L.ivx_conv_fwd_pio_naive(C.byref(d), C.byref(io), _ptr(x.data), _ptr(wgt), _ptr(scale), _ptr(shift), _ptr(rd), _ptr(out), _stream()),
              'ivx_conv_fwd_pio_naive')
    else:
        wsb = L.ivx_conv_pio_workspace_bytes(C.byref(d), C.byref(io))
        if wsb < 0:
            check(-1, 'ivx_conv_pio_workspace_bytes')
        ws = torch.empty((wsb,), device=x.device, dtype=torch.uint8) if wsb > 0 else None
        check(L.ivx_conv_fwd_pio(C.byref(d), C.byref(io), _ptr(x.data), _ptr(wgt), _ptr(scale), _ptr(shift), _ptr(rd), _ptr(out), _ptr(ws),
                                 max(wsb, 0), _stream()), 'ivx_conv_fwd_pio')
    if out_pair:
        return PairTensor(out, slots)
    out.ivx_slots = slots
    return out


def stem_pool_pack_filters(w, scale):
    """Host: [64,3,7,7] stem filters + BN scale [64] (CPU tensors) -> (fragment-ordered pair filters as a uint8 tensor, scale / s_w [64]) through
    ivx_stem_pool_pack_filters (the native handle packs with the same function)."""
    w = w.detach().to(torch.float32).cpu().contiguous()
    if tuple(w.shape) != (64, 3, 7, 7):
        raise ValueError('the one-launch stem is built for [64, 3, 7, 7] filters')
    sc = scale.detach().to(torch.float32).cpu().contiguous()
    L = _lib.lib()
    packed = torch.empty(int(L.ivx_stem_pool_filter_bytes()), dtype=torch.uint8)
    sp = torch.empty(64, dtype=torch.float32)
    check(L.ivx_stem_pool_pack_filters(_ptr_any(w), _ptr_any(sc), _ptr_any(packed), _ptr_any(sp)), 'ivx_stem_pool_pack_filters')
    return packed, sp


def stem_pool_pair(img, wfrag, scale_p, shift, wbound, sbound):
    """fp32 NCHW image [N,3,H,W] -> PairTensor [N,1,Hp,Wp,64]: conv 7x7 s2 p3 + BN + ReLU + MaxPool2d(3,2,1) in one launch (ivx_amax_f32 +
    ivx_stem_pool_fwd_pair, csrc/stem.hip).  wfrag / scale_p: device copies of stem_pool_pack_filters' outputs."""
    _chk(img, 'img')
    N, Cn, H, W = img.shape
    if Cn != 3:
        raise ValueError('stem_pool_pair takes a 3-channel image')
    L = _lib.lib()
    hp, wp = C.c_int32(), C.c_int32()
    check(L.ivx_stem_pool_out_dims(H, W, C.byref(hp), C.byref(wp)), 'ivx_stem_pool_out_dims')
    islots, slots = new_slots(img.device), new_slots(img.device)
    check(L.ivx_amax_f32(_ptr(img), img.numel(), _ptr(islots), _stream()), 'ivx_amax_f32')
    out = torch.empty((N, 1, hp.value, wp.value, 128), device=img.device, dtype=torch.float16)
    check(L.ivx_stem_pool_fwd_pair(_ptr(img), N, H, W, _ptr(wfrag), _ptr(scale_p), _ptr(shift), float(wbound), float(sbound), _ptr(islots), _ptr(out),
                                   _scale_ptr(slots), _ptr(slots), _stream()), 'ivx_stem_pool_fwd_pair')
    return PairTensor(out, slots)


def bottleneck_supported(B, H, W, planes):
    """ivx_bottleneck_supported: the one-launch form of an identity bottleneck exists for this map (csrc/bottleneck.hip)."""
    d = _lib.BottleneckDesc(int(B), int(H), int(W), int(planes))
    return bool(_lib.lib().ivx_bottleneck_supported(C.byref(d)))


def bottleneck_fwd_pio(x, f1, f2, f3):
    """One identity bottleneck (1x1 -> 3x3 -> 1x1 + shortcut, every BN / ReLU) in one launch on a PairTensor [B,1,H,W,4P] (ivx_bottleneck_fwd_pio).
    f1 / f2 / f3: the block's three layers, objects with wpair, scale_p, shift, wbound, sbound (conv.FusedConv(chain=True)).  -> PairTensor."""
    if not isinstance(x, PairTensor):
        raise TypeError('bottleneck_fwd_pio takes a PairTensor')
    _chk(x.data, 'x', torch.float16)
    B, D, H, W, Cn = x.shape
    P = Cn // 4
    if D != 1 or not bottleneck_supported(B, H, W, P) or Cn != 4 * P:
        raise ValueError(f'no fused bottleneck for an input of shape {x.shape}')
    want = ((P, Cn // 32, 1, 64), (P, P // 32, 9, 64), (Cn, P // 32, 1, 64))
    for f, shp in zip((f1, f2, f3), want):
        _chk(f.wpair, 'wpair', torch.float16)
        if tuple(f.wpair.shape) != shp:
            raise ValueError(f'pair filters {tuple(f.wpair.shape)} do not match the bottleneck ({shp})')
        _chk(f.scale_p, 'scale_p')
        _chk(f.shift, 'shift')
    d = _lib.BottleneckDesc(B, H, W, P)
    io = _lib.BottleneckIO()
    slots = new_slots(x.device)
    io.in_scale, io.amax_in = _scale_ptr(x.slots), _ptr(x.slots)
    io.out_scale, io.amax_out = _scale_ptr(slots), _ptr(slots)
    for i, f in enumerate((f1, f2, f3)):
        io.wbound[i], io.sbound[i] = float(f.wbound), float(f.sbound)
    out = torch.empty_like(x.data)
    check(_lib.lib().ivx_bottleneck_fwd_pio(C.byref(d), C.byref(io), _ptr(x.data), _ptr(f1.wpair), _ptr(f1.scale_p), _ptr(f1.shift), _ptr(f2.wpair),
                                            _ptr(f2.scale_p), _ptr(f2.shift), _ptr(f3.wpair), _ptr(f3.scale_p), _ptr(f3.shift), _ptr(out), _stream()),
          'ivx_bottleneck_fwd_pio')
    return PairTensor(out, slots)


def bottleneck_proj_supported(B, H, W, planes, cin):
    """ivx_bottleneck_proj_supported: the one-launch form of a stage's first block (shortcut conv, stride 1) exists for this map."""
    d = _lib.BottleneckDesc(int(B), int(H), int(W), int(planes))
    return bool(_lib.lib().ivx_bottleneck_proj_supported(C.byref(d), int(cin)))


class ProjBank:
    """the joint pair filter bank of conv3 and the shortcut conv (ivx_bottleneck_proj_pack) + its epilogue vectors and bound terms"""

    def __init__(self, f3, fd):
        P, cin = f3.cin, fd.cin
        if f3._w_tap_host is None or fd._w_tap_host is None or f3.cout != 4 * P or fd.cout != 4 * P:
            raise ValueError('bottleneck_proj_pack takes the 1x1 conv3 (P -> 4P) and the 1x1 shortcut conv (Cin -> 4P) of one block, built with chain=True')
        self.planes, self.cin = P, cin
        packed = torch.empty((4 * P, (cin + P) // 32, 1, 64), dtype=torch.float16)
        sc, sf = torch.empty(4 * P, dtype=torch.float32), torch.empty(4 * P, dtype=torch.float32)
        wb3, sb, wbd = C.c_float(), C.c_float(), C.c_float()
        check(_lib.lib().ivx_bottleneck_proj_pack(_ptr_any(f3._w_tap_host), _ptr_any(f3._scale_host), _ptr_any(f3._shift_host), _ptr_any(fd._w_tap_host),
                                                  _ptr_any(fd._scale_host), _ptr_any(fd._shift_host), P, cin, _ptr_any(packed), _ptr_any(sc), _ptr_any(sf),
                                                  C.byref(wb3), C.byref(sb), C.byref(wbd)), 'ivx_bottleneck_proj_pack')
        self._host = (packed, sc, sf)
        self.wbound3, self.sbound, self.wboundd = wb3.value, sb.value, wbd.value
        self.wpair = self.scale_p = self.shift = None

    def to(self, device):
        self.wpair, self.scale_p, self.shift = (t.to(device) for t in self._host)
        return self


def bottleneck_proj_fwd_pio(x, f1, f2, bank):
    """The first block of ResNet stage 1 (1x1 -> 3x3 -> 1x1 + the 1x1 shortcut conv of the input, every BN / ReLU) in one launch on a PairTensor
    [B,1,H,W,Cin] (ivx_bottleneck_proj_fwd_pio).  f1 / f2: conv.FusedConv(chain=True) of conv1 / conv2, bank: ProjBank(conv3, shortcut conv).to(device)
    -> PairTensor [B,1,H,W,4P]."""
    if not isinstance(x, PairTensor):
        raise TypeError('bottleneck_proj_fwd_pio takes a PairTensor')
    _chk(x.data, 'x', torch.float16)
    B, D, H, W, Cn = x.shape
    P = bank.planes
    if D != 1 or Cn != bank.cin or not bottleneck_proj_supported(B, H, W, P, Cn):
        raise ValueError(f'no fused projection bottleneck for an input of shape {x.shape} (planes {P})')
    want = ((P, Cn // 32, 1, 64), (P, P // 32, 9, 64), (4 * P, (Cn + P) // 32, 1, 64))
    for f, shp in zip((f1, f2, bank), want):
        _chk(f.wpair, 'wpair', torch.float16)
        if tuple(f.wpair.shape) != shp:
            raise ValueError(f'pair filters {tuple(f.wpair.shape)} do not match the bottleneck ({shp})')
        _chk(f.scale_p, 'scale_p')
        _chk(f.shift, 'shift')
    d = _lib.BottleneckDesc(B, H, W, P)
    io = _lib.BottleneckIO()
    slots = new_slots(x.device)
    io.in_scale, io.amax_in = _scale_ptr(x.slots), _ptr(x.slots)
    io.out_scale, io.amax_out = _scale_ptr(slots), _ptr(slots)
    for i, f in enumerate((f1, f2)):
        io.wbound[i], io.sbound[i] = float(f.wbound), float(f.sbound)
    io.wbound[2], io.sbound[2] = float(bank.wbound3), float(bank.sbound)
    out = torch.empty((B, 1, H, W, 8 * P), device=x.data.device, dtype=torch.float16)
    check(_lib.lib().ivx_bottleneck_proj_fwd_pio(C.byref(d), Cn, C.byref(io), float(bank.wboundd), _ptr(x.data), _ptr(f1.wpair), _ptr(f1.scale_p),
                                                 _ptr(f1.shift), _ptr(f2.wpair), _ptr(f2.scale_p), _ptr(f2.shift), _ptr(bank.wpair), _ptr(bank.scale_p),
                                                 _ptr(bank.shift), _ptr(out), _stream()), 'ivx_bottleneck_proj_fwd_pio')
    return PairTensor(out, slots)


def global_avgpool(x):
    """[B,D,H,W,C] channels-last -> [B,1,1,1,C]: mean over every spatial position."""
    _chk(x, 'x')
    B, Cn = x.shape[0], x.shape[-1]
    S = x.numel() // (B * Cn)
    out = torch.empty((B, 1, 1, 1, Cn), device=x.device, dtype=torch.float32)
    check(_lib.lib().ivx_global_avgpool_fwd(_ptr(x), B, S, Cn, _ptr(out), _stream()), 'ivx_global_avgpool_fwd')
    return out


def dcn_im2col(x, offset_mask, kernel=3, stride=1, pad=1, dil=1):
    """x [B,1,H,W,C], offset_mask [B,1,Ho,Wo,>=3*k*k] -> modulated deformable columns [B,1,Ho,Wo,k*k*C]."""
    _chk(x, 'x')
    _chk(offset_mask, 'offset_mask')
    B, D, H, W, Cn = x.shape
    Ho = (H + 2 * pad - (dil * (kernel - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kernel - 1) + 1)) // stride + 1
    if D != 1 or tuple(offset_mask.shape[:4]) != (B, 1, Ho, Wo):
        raise ValueError('offset/mask map does not match the output size')
    col = torch.empty((B, 1, Ho, Wo, kernel * kernel * Cn), device=x.device, dtype=torch.float32)
    check(_lib.lib().ivx_dcn_im2col_fwd(_ptr(x), _ptr(offset_mask), B, H, W, Cn, kernel, kernel, stride, pad, dil,
                                        offset_mask.shape[4], _ptr(col), _stream()), 'ivx_dcn_im2col_fwd')
    return col


def dcn_im2col_pair(x, offset_mask, kernel=3, stride=1, pad=1, dil=1):
    """The columns inside the pair chain (ivx_dcn_im2col_fwd_pair): x a PairTensor [B,1,H,W,2C], offset_mask fp32 [B,1,Ho,Wo,>=3*k*k] ->
    PairTensor [B,1,Ho,Wo,2*k*k*C] with x's scale and its own recorded maximum."""
    if not isinstance(x, PairTensor):
        raise TypeError('dcn_im2col_pair takes a PairTensor')
    _chk(offset_mask, 'offset_mask')
    B, D, H, W, C2 = x.data.shape
    Cn = C2 // 2
    Ho = (H + 2 * pad - (dil * (kernel - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kernel - 1) + 1)) // stride + 1
    if D != 1 or Cn % 16 or tuple(offset_mask.shape[:4]) != (B, 1, Ho, Wo):
        raise ValueError('offset/mask map does not match the output size (or C % 16 != 0)')
    col = torch.empty((B, 1, Ho, Wo, 2 * kernel * kernel * Cn), device=x.data.device, dtype=torch.float16)
    slots = new_slots(x.data.device)
    check(_lib.lib().ivx_dcn_im2col_fwd_pair(_ptr(x.data), _scale_ptr(x.slots), _ptr(offset_mask), B, H, W, Cn, kernel, kernel, stride, pad, dil,
                                             offset_mask.shape[4], _ptr(col), _scale_ptr(slots), _ptr(slots), _stream()), 'ivx_dcn_im2col_fwd_pair')
    return PairTensor(col, slots)


def upsample_trilinear2x(x):
    if x.dtype not in _DT:
        raise TypeError(f'x must be float32 or bfloat16, got {x.dtype}')
    _chk(x, 'x', x.dtype)
    B, D, H, W, Cn = x.shape
    out = torch.empty((B, 2 * D, 2 * H, 2 * W, Cn), device=x.device, dtype=x.dtype)
    fn = _lib.lib().ivx_upsample_trilinear2x_fwd if x.dtype == torch.float32 else _lib.lib().ivx_upsample_trilinear2x_fwd_bf16
    check(fn(_ptr(x), B, D, H, W, Cn, _ptr(out), _stream()), 'ivx_upsample_trilinear2x_fwd')
    return out


# ------------------------------------------------------------------ unprojection
# optional stage timing (bench.py): when a list, backproject_mean appends ('lift', start_event, end_event) around its launch
stage_trace = None


def backproject_mean(feat, proj, new_origin, crop_hw, voxel_size, n_voxels):
    if stage_trace is None:
        return _backproject_mean(feat, proj, new_origin, crop_hw, voxel_size, n_voxels)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = _backproject_mean(feat, proj, new_origin, crop_hw, voxel_size, n_voxels)
    e1.record()
    stage_trace.append(('lift', e0, e1))
    return out


def _backproject_mean(feat, proj, new_origin, crop_hw, voxel_size, n_voxels):
    """feat [B*V,1,FH,FW,C] channels-last, proj [B,V,3,4], new_origin [B,3], crop_hw [B,2] int32 (device)
    -> volume [B,X,Y,Z,C] (feat's dtype), valid [B,X,Y,Z] bool.
    bf16 storage (optional reduced-precision mode): with one view the lift is a pure gather-copy (no arithmetic on the
    features, imvoxelnet.py:75 divides by a count of 1), so a bf16 map with C channels is passed as C/2 32-bit words."""
    if feat.dtype == torch.bfloat16:
        if proj.shape[1] == 1 and feat.shape[-1] % 2 == 0:
            vol, valid = _backproject_mean(feat.view(torch.float32), proj, new_origin, crop_hw, voxel_size, n_voxels)
            return vol.view(torch.bfloat16), valid
        _chk(feat, 'feat', torch.bfloat16)
        _chk(proj, 'proj')
        _chk(new_origin, 'new_origin')
        _chk(crop_hw, 'crop_hw', torch.int32)
        B, V = proj.shape[0], proj.shape[1]
        BV, D, FH, FW, Cn = feat.shape
        if BV != B * V or D != 1 or tuple(proj.shape[2:]) != (3, 4):
            raise ValueError('feat / proj shapes do not agree')
        X, Y, Z = (int(v) for v in n_voxels)
        vol = torch.empty((B, X, Y, Z, Cn), device=feat.device, dtype=torch.bfloat16)
        valid = torch.empty((B, X, Y, Z), device=feat.device, dtype=torch.uint8)
        vs = (C.c_float * 3)(*[float(v) for v in voxel_size])
        check(_lib.lib().ivx_backproject_mean_fwd_bf16(_ptr(feat), B, V, FH, FW, Cn, _ptr(proj), _ptr(new_origin), _ptr(crop_hw),
                                                       vs, X, Y, Z, _ptr(vol), _ptr(valid), _stream()), 'ivx_backproject_mean_fwd_bf16')
        return vol, valid.view(torch.bool)
    _chk(feat, 'feat')
    _chk(proj, 'proj')
    _chk(new_origin, 'new_origin')
    _chk(crop_hw, 'crop_hw', torch.int32)
    B, V = proj.shape[0], proj.shape[1]
    BV, D, FH, FW, Cn = feat.shape
    if BV != B * V or D != 1 or tuple(proj.shape[2:]) != (3, 4):
        raise ValueError('feat / proj shapes do not agree')
    X, Y, Z = (int(v) for v in n_voxels)
    vol = torch.empty((B, X, Y, Z, Cn), device=feat.device, dtype=torch.float32)
    valid = torch.empty((B, X, Y, Z), device=feat.device, dtype=torch.uint8)
    vs = (C.c_float * 3)(*[float(v) for v in voxel_size])
    check(_lib.lib().ivx_backproject_mean_fwd(_ptr(feat), B, V, FH, FW, Cn, _ptr(proj), _ptr(new_origin), _ptr(crop_hw),
                                              vs, X, Y, Z, _ptr(vol), _ptr(valid), _stream()), 'ivx_backproject_mean_fwd')
    return vol, valid.view(torch.bool)


def backproject_sum(feat, proj, new_origin, crop_hw, voxel_size, n_voxels):
    """View-sharded mode: like backproject_mean but returns the raw view sum [B,X,Y,Z,C] and the int32 view count
    [B,X,Y,Z] of THIS rank's views (to be all-reduced, then volume_normalize_)."""
    _chk(feat, 'feat')
    _chk(proj, 'proj')
    _chk(new_origin, 'new_origin')
    _chk(crop_hw, 'crop_hw', torch.int32)
    B, V = proj.shape[0], proj.shape[1]
    BV, D, FH, FW, Cn = feat.shape
    if BV != B * V or D != 1 or tuple(proj.shape[2:]) != (3, 4):
        raise ValueError('feat / proj shapes do not agree')
    X, Y, Z = (int(v) for v in n_voxels)
    vol = torch.empty((B, X, Y, Z, Cn), device=feat.device, dtype=torch.float32)
    cnt = torch.empty((B, X, Y, Z), device=feat.device, dtype=torch.int32)
    vs = (C.c_float * 3)(*[float(v) for v in voxel_size])
    check(_lib.lib().ivx_backproject_sum_fwd(_ptr(feat), B, V, FH, FW, Cn, _ptr(proj), _ptr(new_origin), _ptr(crop_hw),
                                             vs, X, Y, Z, _ptr(vol), _ptr(cnt), _stream()), 'ivx_backproject_sum_fwd')
    return vol, cnt


def volume_normalize_(vol_sum, count):
    """In place: vol = count ? vol / count : 0; returns (vol, valid bool [B,X,Y,Z])."""
    _chk(vol_sum, 'vol_sum')
    _chk(count, 'count', torch.int32)
    if tuple(count.shape) != tuple(vol_sum.shape[:-1]):
        raise ValueError('count must have the spatial shape of the volume')
    valid = torch.empty(count.shape, device=vol_sum.device, dtype=torch.uint8)
    check(_lib.lib().ivx_volume_normalize_fwd(_ptr(vol_sum), _ptr(count), count.numel(), vol_sum.shape[-1], _ptr(valid), _stream()),
          'ivx_volume_normalize_fwd')
    return vol_sum, valid.view(torch.bool)


# ------------------------------------------------------------------ detection tail
def anchor_head_get_bboxes(head_out, anchors, H, W, num_anchors, num_classes, offs, cfg, dir_offset=0.0,
                           dir_limit_offset=1.0, hw_transposed=False, want_candidates=False):
    """head_out [B,*,*,CH] channels-last map of the fused head conv; anchors [H*W*A,7].
    Returns (boxes [B,max_num,7], scores [B,max_num], labels [B,max_num] int64, count [B] int32[, cands])."""
    _chk(head_out, 'head_out')
    _chk(anchors, 'anchors')
    B, CH = head_out.shape[0], head_out.shape[-1]
    if head_out.numel() != B * H * W * CH:
        raise ValueError('head_out does not hold B*H*W*CH values')
    if anchors.shape[0] != H * W * num_anchors or anchors.shape[1] != 7:
        raise ValueError('anchors must be [H*W*A, 7]')
    nms_pre, max_num = int(cfg['nms_pre']), int(cfg['max_num'])
    d = AnchorHeadDesc(B, H, W, CH, num_anchors, num_classes, offs[0], offs[1], offs[2], nms_pre, max_num,
                       int(bool(cfg['use_rotate_nms'])), int(bool(hw_transposed)), float(cfg.get('score_thr', 0)),
                       float(cfg['nms_thr']), float(dir_offset), float(dir_limit_offset))
    L = _lib.lib()
    ws_bytes = L.ivx_anchor_head_workspace_bytes(C.byref(d))
    if ws_bytes < 0:
        check(-1, 'ivx_anchor_head_workspace_bytes')
    ws = torch.empty((ws_bytes,), device=head_out.device, dtype=torch.uint8)
    boxes = torch.empty((B, max_num, 7), device=head_out.device, dtype=torch.float32)
    scores = torch.empty((B, max_num), device=head_out.device, dtype=torch.float32)
    labels = torch.empty((B, max_num), device=head_out.device, dtype=torch.int64)
    count = torch.empty((B,), device=head_out.device, dtype=torch.int32)
    ci = cb = cs = None
    if want_candidates:
        ci = torch.empty((B, nms_pre), device=head_out.device, dtype=torch.int64)
        cb = torch.empty((B, nms_pre, 7), device=head_out.device, dtype=torch.float32)
        cs = torch.empty((B, nms_pre), device=head_out.device, dtype=torch.float32)
    check(L.ivx_anchor_head_get_bboxes(C.byref(d), _ptr(head_out), _ptr(anchors), _ptr(ws), ws_bytes, _ptr(boxes),
                                       _ptr(scores), _ptr(labels), _ptr(count), _ptr(ci), _ptr(cb), _ptr(cs), _stream()),
          'ivx_anchor_head_get_bboxes')
    if want_candidates:
        return boxes, scores, labels, count, (ci, cb, cs)
    return boxes, scores, labels, count


def nms_bev_sorted(boxes_sorted, thresh, rotated=True):
    """boxes [n,5] sorted by descending score -> (keep [n] int64, num [1] int32) device tensors."""
    _chk(boxes_sorted, 'boxes')
    n = boxes_sorted.shape[0]
    L = _lib.lib()
    ws_bytes = L.ivx_nms_workspace_bytes(n)
    ws = torch.empty((max(int(ws_bytes), 256),), device=boxes_sorted.device, dtype=torch.uint8)
    keep = torch.empty((max(n, 1),), device=boxes_sorted.device, dtype=torch.int64)
    num = torch.empty((1,), device=boxes_sorted.device, dtype=torch.int32)
    check(L.ivx_nms_bev(_ptr(boxes_sorted), n, float(thresh), int(bool(rotated)), _ptr(ws), ws.numel(), _ptr(keep),
                        _ptr(num), _stream()), 'ivx_nms_bev')
    return keep, num


def multiclass_nms_bev(boxes, scores, num_classes, score_thr, nms_thr, rotated, max_num):
    """boxes [n,5] BEV (x1,y1,x2,y2,ry), scores [n, >= num_classes] -> (idx [m] int64 into the n candidates,
    labels [m] int64, count [1] int32), all on the device; m = min(max_num, n * num_classes) slots, `count` valid."""
    _chk(boxes, 'boxes')
    _chk(scores, 'scores')
    n = boxes.shape[0]
    if scores.shape[0] != n or scores.shape[1] < num_classes or boxes.shape[1] != 5:
        raise ValueError('boxes must be [n,5] and scores [n, >= num_classes]')
    L = _lib.lib()
    wsb = L.ivx_multiclass_nms_workspace_bytes(n, int(num_classes))
    if wsb < 0:
        check(-1, 'ivx_multiclass_nms_workspace_bytes')
    ws = torch.empty((max(int(wsb), 256),), device=boxes.device, dtype=torch.uint8)
    m = max(1, min(int(max_num), n * int(num_classes)))
    idx = torch.empty((m,), device=boxes.device, dtype=torch.int64)
    lab = torch.empty((m,), device=boxes.device, dtype=torch.int64)
    cnt = torch.empty((1,), device=boxes.device, dtype=torch.int32)
    check(L.ivx_multiclass_nms_bev(_ptr(boxes), _ptr(scores), n, scores.shape[1], int(num_classes), float(score_thr), float(nms_thr),
                                   int(bool(rotated)), int(max_num), _ptr(ws), ws.numel(), _ptr(idx), _ptr(lab), _ptr(cnt), _stream()),
          'ivx_multiclass_nms_bev')
    return idx, lab, cnt


def indoor_tail(cand_boxes, cand_scores, n_classes, score_thr, nms_thr, use_rotate_nms=False, max_num=None):
    """Cross-level tail of the anchor-free indoor heads (ivx_indoor_tail_get_bboxes): cand_boxes / cand_scores = per-level lists of
    [B, k_l, R] / [B, k_l, n_classes] device tensors (R = 6 ScanNet corners, 7 SUN RGB-D boxes) -> (boxes [B,M,7] rows of the box
    object's tensor (bottom-face z), scores [B,M], labels int64 [B,M], count int32 [B]); M = sum(k_l) for ScanNet, max_num (the
    reference passes test_cfg.nms_pre) capped at sum(k_l) * n_classes for SUN RGB-D."""
    from ._lib import IndoorTailDesc
    L = _lib.lib()
    B, R = cand_boxes[0].shape[0], cand_boxes[0].shape[2]
    d = IndoorTailDesc()
    d.B, d.n_levels, d.n_classes, d.n_reg = B, len(cand_boxes), int(n_classes), R
    K = 0
    for l, (cb, cs) in enumerate(zip(cand_boxes, cand_scores)):
        _chk(cb, 'cand_boxes')
        _chk(cs, 'cand_scores')
        if cb.shape[0] != B or cs.shape[:2] != cb.shape[:2] or cs.shape[2] != n_classes or cb.shape[2] != R:
            raise ValueError('candidate lists must be [B, k, R] / [B, k, n_classes] per level')
        d.k[l] = cb.shape[1]
        K += cb.shape[1]
    M = K if R == 6 else max(1, min(int(max_num if max_num is not None else K), K * int(n_classes)))
    d.use_rotate_nms, d.max_num, d.score_thr, d.nms_thr = int(bool(use_rotate_nms)), M, float(score_thr), float(nms_thr)
    wsb = L.ivx_indoor_tail_workspace_bytes(C.byref(d))
    if wsb < 0:
        check(-1, 'ivx_indoor_tail_workspace_bytes')
    dev = cand_boxes[0].device
    ws = torch.empty((max(int(wsb), 256),), device=dev, dtype=torch.uint8)
    boxes, scores = torch.empty((B, M, 7), device=dev, dtype=torch.float32), torch.empty((B, M), device=dev, dtype=torch.float32)
    labels, count = torch.empty((B, M), device=dev, dtype=torch.int64), torch.empty((B,), device=dev, dtype=torch.int32)
    pb = (C.c_void_p * 4)(*([t.data_ptr() for t in cand_boxes] + [None] * (4 - len(cand_boxes))))
    ps = (C.c_void_p * 4)(*([t.data_ptr() for t in cand_scores] + [None] * (4 - len(cand_scores))))
    check(L.ivx_indoor_tail_get_bboxes(C.byref(d), pb, ps, _ptr(ws), ws.numel(), _ptr(boxes), _ptr(scores), _ptr(labels), _ptr(count), _stream()),
          'ivx_indoor_tail_get_bboxes')
    return boxes, scores, labels, count


def boxes_overlap_bev(a, b, iou=False):
    _chk(a, 'a')
    _chk(b, 'b')
    out = torch.zeros((a.shape[0], b.shape[0]), device=a.device, dtype=torch.float32)
    check(_lib.lib().ivx_boxes_overlap_bev(_ptr(a), a.shape[0], _ptr(b), b.shape[0], int(bool(iou)), _ptr(out), _stream()),
          'ivx_boxes_overlap_bev')
    return out


def aligned_3d_nms_dev(boxes, scores, classes, thresh, single_workgroup=False):
    _chk(boxes, 'boxes')
    _chk(scores, 'scores')
    _chk(classes, 'classes', torch.int64)
    n = boxes.shape[0]
    pick = torch.empty((max(n, 1),), device=boxes.device, dtype=torch.int64)
    num = torch.empty((1,), device=boxes.device, dtype=torch.int32)
    L = _lib.lib()
    if single_workgroup:
        check(L.ivx_aligned_3d_nms(_ptr(boxes), _ptr(scores), _ptr(classes), n, float(thresh), _ptr(pick), _ptr(num), _stream()),
              'ivx_aligned_3d_nms')
        return pick, num
    wsb = L.ivx_aligned_3d_nms_workspace_bytes(n)
    if wsb < 0:
        raise ValueError(f'ivx_aligned_3d_nms_ws: at most 65536 boxes (got {n})')
    ws = torch.empty((wsb,), device=boxes.device, dtype=torch.uint8)
    check(L.ivx_aligned_3d_nms_ws(_ptr(boxes), _ptr(scores), _ptr(classes), n, float(thresh), _ptr(ws), wsb, _ptr(pick), _ptr(num),
                                  _stream()), 'ivx_aligned_3d_nms_ws')
    return pick, num
