"""Host side of the fused convolution: packs reference-format parameters for ivx_conv_fwd.

A `FusedConv` is the deploy form of  Conv{2,3}d [+bias] -> [eval BatchNorm] -> [+residual] -> [ReLU]:
  weights  [Cout,Cin,(kd,)kh,kw] (torch layout)  ->  [Cout,kd,kh,kw,Cin_pad4]  (K contiguous per filter)
  epilogue y = acc*scale + shift  with  scale = gamma/sqrt(var+eps),  shift = beta + (bias - mean)*scale
The reference never fuses Conv3d+BN (tools/fuse_conv_bn.py handles Conv2d only); applying the BN affine
in the epilogue instead of folding it into the weights keeps the accumulate identical to conv-then-BN.
"""
import os

import torch

from . import ops


def _spatial3(v, dims, fill):
    """int / 2-tuple / 3-tuple -> (d, h, w); a 2-D op gets `fill` on the depth axis."""
    if isinstance(v, int):
        return (fill, v, v) if dims == 2 else (v, v, v)
    v = tuple(int(i) for i in v)
    if len(v) == 2:
        return (fill,) + v
    if len(v) != 3:
        raise ValueError(f'expected an int, 2-tuple or 3-tuple, got {v}')
    return v


def fold_batchnorm(bn, bias, eps=1e-5):
    """(gamma, beta, running_mean, running_var) [+ conv bias] -> (scale, shift) of the conv epilogue, through the library's
    host function ivx_fold_batchnorm (IEEE fp32, fixed operation order), so every host of the kernels -- this module and
    the native model handle (csrc/model.cpp) -- hands them identical bits."""
    import ctypes as C
    from . import _lib
    g, b, m, v = (t.detach().to(torch.float32).cpu().contiguous() for t in bn)
    n = g.numel()
    bias_t = None if bias is None else bias.detach().to(torch.float32).cpu().contiguous()
    scale, shift = torch.empty(n), torch.empty(n)
    _lib.check(_lib.lib().ivx_fold_batchnorm(C.c_void_p(g.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()),
                                             None if bias_t is None else C.c_void_p(bias_t.data_ptr()), float(eps), n,
                                             C.c_void_p(scale.data_ptr()), C.c_void_p(shift.data_ptr())), 'ivx_fold_batchnorm')
    return scale, shift


_storage = [torch.float32]


class storage_dtype:
    """with storage_dtype(torch.bfloat16): FusedConvs built inside store activations and weights as bf16 (optional
    reduced-precision mode; the default and the reference's precision is float32).  Accumulation stays fp32."""

    def __init__(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16, torch.float8_e4m3fn):
            raise TypeError('storage dtype must be torch.float32, torch.bfloat16 or torch.float8_e4m3fn')
        self.dtype = dtype

    def __enter__(self):
        _storage.append(self.dtype)
        return self

    def __exit__(self, *exc):
        _storage.pop()
        return False


def current_storage_dtype():
    return _storage[-1]


FP8 = torch.float8_e4m3fn
FP8_MAX = 448.0


def pack_pair_weights(w5, layout=1):
    """fp32 filters [Cout,kd,kh,kw,Cin] (Cin % 16 == 0) -> the IVX_BF16_PAIR operand (include/imvoxel.h): bf16, every value as
    hi = bf16(w), lo = bf16(w - hi), 16-channel groups [hi x16 | lo x16]; layout 0: [Cout,kd,kh,kw,2Cin], layout 1 (Cin % 32 == 0):
    chunk-major [Cout, 2Cin/64, kd,kh,kw, 64]."""
    w5 = w5.detach().to(torch.float32)
    co, kd, kh, kw, ci = w5.shape
    if ci % 16 or (layout == 1 and ci % 32):
        raise ValueError('pair filters need Cin % 16 == 0 (layout 1: % 32)')
    hi = w5.to(torch.bfloat16)
    lo = (w5 - hi.to(torch.float32)).to(torch.bfloat16)
    wp = torch.cat([hi.reshape(co, kd, kh, kw, ci // 16, 16), lo.reshape(co, kd, kh, kw, ci // 16, 16)], dim=-1).reshape(co, kd, kh, kw, 2 * ci)
    if layout == 1:
        wp = wp.reshape(co, kd, kh, kw, 2 * ci // 64, 64).permute(0, 4, 1, 2, 3, 5)
    return wp.contiguous()


class QTensor:
    """An e4m3 activation with its per-tensor scale: value = data.float() * scale (optional fp8 storage of the 2-D trunk in
    the bf16 mode).  Produced and consumed by FusedConv / ops.maxpool2d; `scale` is a Python float fixed at calibration."""
    __slots__ = ('data', 'scale')

    def __init__(self, data, scale):
        self.data, self.scale = data, float(scale)

    shape = property(lambda self: self.data.shape)
    device = property(lambda self: self.data.device)
    dtype = property(lambda self: self.data.dtype)

    def numel(self):
        return self.data.numel()

    def element_size(self):
        return 1

    def float(self):
        return self.data.float() * self.scale


class FusedConv:
    # optional algorithmic-FLOP accounting (bench.py): 2 * output positions * Cout * Cin * taps per call;
    # exec_flops counts what the MFMA kernel actually executes (fewer for layers run in the Winograd form)
    count_flops = False
    flops = 0.0
    exec_flops = 0.0
    # fp32 3x3xk layers with stride 1 on the first two axes, >= winograd_min_ch input or output channels and
    # >= winograd_min_pos input positions run as F(m x m, 3x3) (ivx_conv_winograd_fwd).  Measured on the KITTI neck
    # (batch 4, tools/conv_bench.py --winograd, profiles/r01_conv_layers.log), direct -> m = 2 -> 4 -> 6 in ms:
    # 256->256 16.1 -> 9.0 -> 5.3 -> 4.3, 128->128 8.2 -> 5.7 -> 3.3 -> 2.7, 64->128 (z stride 2) 4.6 -> 3.9 -> 2.3 -> 1.9,
    # 64->64 4.8 -> 4.3 -> 2.6 -> 2.2.  Also a gain on the indoor necks down to a few thousand positions (SUN RGB-D fast
    # 123 -> 165 scenes/s); only the coarsest levels (< winograd_min_pos positions) stay direct.
    winograd = os.environ.get('IVX_WINOGRAD', '1') != '0'
    # m of F(m x m, 3x3): 0 = automatic (6 when a sample's output plane has >= winograd_tile6_min_plane positions on the
    # transformed axes -- the KITTI / nuScenes necks, full-resolution 2-D maps -- else 4: on the 40 x 40 and 80 x 80 indoor
    # volumes the 6 x 6 tiles waste up to 10 % at the border and leave too few tiles per plane; measured SUN RGB-D fast
    # 166 scenes/s with m = 4 vs 157 with m = 6, KITTI 118.6 vs 135.9 images/s); 2, 4 or 6 force one tile everywhere
    winograd_tile = int(os.environ.get('IVX_WINOGRAD_TILE', '0'))
    winograd_tile6_min_plane = 16384
    winograd_min_ch = 64
    winograd_2d_min_ch = int(os.environ.get('IVX_WINOGRAD_2D_MIN_CH', '128'))   # 2-D 3x3 layers (ResNet conv2, FPN outputs)
    winograd_min_pos = int(os.environ.get('IVX_WINOGRAD_MIN_POS', '2000'))
    # Split-operand direct form (include/imvoxel.h IVX_BF16_PAIR) for the layers the Winograd form does not take (1x1, strided, ...):
    # fp32 activations and filters as (hi, lo) bf16 pairs, three bf16 MFMA products per pair with fp32 accumulation -- 16x the fp32 MFMA
    # rate at 2^-17 operand precision, one split pass over the input.  Measured on the KITTI neck it loses to the Winograd form
    # (1.9 / 2.9 / 5.2 ms vs 1.8 / 2.7 / 4.5 for the 64 / 128 / 256-channel layers: the minimal-filtering form needs 5x fewer products).
    # -1 (default): the rule -- 3x3x3 layers with Cin % 32 == 0 and Cout >= 64 that the Winograd form does not take (the strided convolutions of
    #     NuScenesImVoxelNeck / FastIndoorImVoxelNeck / the Atlas encoder, and the layers of the coarsest levels: fewer than winograd_min_pos
    #     positions under a K loop of 13824 .. 27648), from SPLIT_MIN_POS input positions on, when the Winograd-domain GEMMs run on 16-bit operands
    #     too (wino_operands = 4; with 0 every product of the neck stays on fp32 MFMA).  Measured (tools/neck_layers.py, profiles/r06_split_form.md):
    #     64 -> 128 stride 2 at 312 x 312 x 12 0.616 -> 0.369 ms, 256 -> 512 stride 2 at 40 x 40 x 16 0.239 -> 0.126, 512 -> 512 at 10 x 10 x 4
    #     0.084 -> 0.065; 1x1x1 layers and the Cout = 25 head convs gain nothing (HBM / latency-bound) and stay fp32.  csrc/model.cpp plan_conv mirrors it.
    # 0: off;  1: every 3-D layer with Cin % 32 == 0 from pair_min_pos positions on;  2: 2-D layers as well
    pair_mode = int(os.environ.get('IVX_CONV_PAIR', '-1'))
    pair_min_pos = 2000
    SPLIT_MIN_POS = 256
    # Operands of the Winograd-domain GEMMs (ivx_conv_desc.wino_operands): 4 = fp16 (hi, lo) pairs, three fp16 MFMA products per pair
    # (~3.3x the fp32 MFMA rate at 22-bit operands: the error stays at the level of the fp32 form's own rounding);
    # 0 = fp32 MFMA (exact fp32 products)
    wino_operands = int(os.environ.get('IVX_WINO_OPERANDS', '4'))
    # 2-D trunk (ResNet + FPN; ivx_model_cfg.trunk_operands): 4 = activations chained as fp16 (hi, lo) pair tensors with device-side
    # scales (ops.PairTensor, ivx_conv_fwd_pio): every convolution whose input is such a tensor issues three fp16 MFMA products per
    # multiply-add with fp32 accumulation and its epilogue writes the operand of the next layer -- no conversion pass; 0 = fp32 MFMA
    trunk_operands = int(os.environ.get('IVX_TRUNK_OPERANDS', '4'))
    # optional per-call timing (bench.py): when a list, every call appends
    # (kind, start_event, end_event, executed_flops, bytes, is_3d) with kind 'direct' | 'wino_input' | 'wino_gemm' |
    # 'wino_output'; the events bracket exactly the launches of that stage on the stream they run on; is_3d tells the 3-D
    # neck layers from the 2-D trunk
    trace = None
    # fp8 calibration (optional fp8 storage of the 2-D trunk): while `calib` is a dict, every call records the running
    # max |output| under the layer's key (id of its weight parameter); layers built later with out_dtype FP8 read their output
    # scale from it: scale = amax * calib_margin / 448
    calib = None
    calib_margin = 1.0

    def __init__(self, weight, bias=None, bn=None, stride=1, padding=0, relu=False, dims=3, eps=1e-5, layout=None,
                 dtype=None, out_dtype=None, key=None, chain=False):
        """weight: [Cout,Cin,kh,kw] (dims=2) or [Cout,Cin,kd,kh,kw] (dims=3) tensor (any device).
        bn: None or (gamma, beta, running_mean, running_var).
        dtype: storage type of the input and the packed weights (float32 = the reference's precision; bfloat16 is the
        optional reduced-precision mode); out_dtype: storage type of the output / residual (default: dtype)."""
        dtype = _storage[-1] if dtype is None else dtype
        out_dtype = dtype if out_dtype is None else out_dtype
        self.key = id(weight) if key is None else key
        w = weight.detach().to(torch.float32)
        if dtype == torch.bfloat16 and w.shape[1] % 8 != 0:
            dtype = torch.float32      # e.g. the 3-channel stem: fp32 image in, reduced-precision map out
        if dtype == FP8 and w.shape[1] % 16 != 0:
            raise ValueError('fp8 input needs a multiple of 16 input channels')
        self.w_scale = None            # fp8 weights: per-output-channel scale, folded into the epilogue scale
        if dtype == FP8:
            self.w_scale = (w.reshape(w.shape[0], -1).abs().amax(1).clamp_min(1e-12) / FP8_MAX).cpu()
            w = w / self.w_scale.view(-1, *([1] * (w.dim() - 1))).to(w.device)
        self.out_scale = 1.0
        if out_dtype == FP8:
            if FusedConv.calib is None or self.key not in FusedConv.calib:
                raise RuntimeError('fp8 output needs a calibration record for this layer (FusedConv.calib; see ImVoxelNet.calibrate_fp8)')
            self.out_scale = max(float(FusedConv.calib[self.key]), 1e-12) * FusedConv.calib_margin / FP8_MAX
        self._qvec = {}                # (in_scale, out_scale) -> device (scale, shift) with the tensor scales folded in
        self.dtype, self.out_dtype = dtype, out_dtype
        if dims == 2:
            w = w.unsqueeze(2)
        self.stride = _spatial3(stride, dims, 1)
        self.padding = _spatial3(padding, dims, 0)
        self.cout, self.cin = w.shape[0], w.shape[1]
        self.kernel = tuple(w.shape[2:])
        epc = {torch.float32: 4, torch.bfloat16: 8, FP8: 16}[dtype]       # elements per 16-byte chunk
        ck = {torch.float32: 32, torch.bfloat16: 64, FP8: 128}[dtype]     # channels per 128-byte chunk (layout 1)
        self.cin_pad = (self.cin + epc - 1) // epc * epc
        wp = w.permute(0, 2, 3, 4, 1).contiguous()
        if self.cin_pad != self.cin:
            wp = torch.nn.functional.pad(wp, (0, self.cin_pad - self.cin))
        wp_tap = wp                                  # [Cout,kd,kh,kw,Cin_pad], tap-major: what the pair form packs (chain=True)
        # layout 1 (chunk-major K) whenever the channel count allows it: see include/imvoxel.h
        self.layout = 1 if (self.cin_pad % ck == 0 and layout != 0) else 0
        if self.layout == 1:
            co, kd, kh, kw, ci = wp.shape
            wp = wp.reshape(co, kd, kh, kw, ci // ck, ck).permute(0, 4, 1, 2, 3, 5)
        self._w_host = wp.contiguous().to(dtype)
        # candidate for the minimal-filtering form: keep the tap-major fp32 filters for ivx_conv_winograd_weights
        # 3-D layers transform their first two axes (the z axis stays direct); a 2-D 3x3 layer [B,1,H,W,C] is the same thing
        # on the view [B,H,W,1,C] with a 3x3x1 kernel (identical memory for the activations and the tap-major filters)
        self._w0_host = None
        self._wino2d = self.kernel == (1, 3, 3) and self.stride == (1, 1, 1)
        wino3d = self.kernel[0] == 3 and self.kernel[1] == 3 and self.stride[0] == 1 and self.stride[1] == 1
        min_ch = FusedConv.winograd_2d_min_ch if self._wino2d else FusedConv.winograd_min_ch
        if ((wino3d or self._wino2d) and dtype == torch.float32 and out_dtype == torch.float32 and self.cin_pad == self.cin
                and self.cout % 4 == 0 and max(self.cin, self.cout) >= min_ch and self.cin % 4 == 0 and type(self) is FusedConv):
            w0 = w.permute(0, 2, 3, 4, 1).contiguous()                       # [Cout,kd,kh,kw,Cin]
            self._w0_host = w0.reshape(self.cout, 3, 3, 1, self.cin) if self._wino2d else w0
        # candidate for the split-operand form: pair-packed filters (made on the host once)
        self._wp_host = None
        self._dims = dims
        self._split_cand = (dims == 3 and self.kernel == (3, 3, 3) and self.cout >= 64 and dtype == torch.float32 and out_dtype == torch.float32
                            and self.cin_pad == self.cin and self.cin % 32 == 0 and type(self) is FusedConv)
        if ((self._split_cand and FusedConv.pair_mode < 0) or
                (dtype == torch.float32 and out_dtype == torch.float32 and self.cin_pad == self.cin and self.cin % 32 == 0 and type(self) is FusedConv
                 and FusedConv.pair_mode >= (1 if dims == 3 else 2))):
            self._wp_host = pack_pair_weights(w.permute(0, 2, 3, 4, 1).contiguous(), 1)
        self.wp = None
        self.u = None          # {tile: transformed filters}, filled by to() / on first use of a tile
        self._w0 = None
        scale = torch.ones(self.cout)
        shift = torch.zeros(self.cout)
        if bias is not None:
            shift = bias.detach().to(torch.float32).cpu().clone()
        if bn is not None:
            scale, shift = fold_batchnorm(bn, shift, eps)
        self._identity_epilogue = bias is None and bn is None
        self._scale_host, self._shift_host = scale.contiguous(), shift.contiguous()
        self.relu = relu
        self.out_mode = 0
        self.w = self.scale = self.shift = None
        # chain=True (the layers of the 2-D trunk when FusedConv.trunk_operands == 4): pair filters + scale / s_w and the terms of the
        # output bound (ivx_pair_pack_filters, as csrc/model.cpp pack_layer); layers the pair form does not take (the stem) get the bound only
        self.pair_ok = False
        self.wbound = self.sbound = None
        self._wpair_host = self._scale_p_host = self.wpair = self.scale_p = None
        self._w_tap_host = None
        if chain and dtype == torch.float32 and out_dtype == torch.float32 and dims == 2 and type(self) is FusedConv:
            self.pair_ok = self.cin_pad == self.cin and self.cin % 32 == 0 and self.cout % 4 == 0
            taps = self.kernel[0] * self.kernel[1] * self.kernel[2]
            self._wpair_host, self._scale_p_host, self.wbound, self.sbound = ops.pair_pack_filters(
                wp_tap.reshape(self.cout, taps, self.cin_pad), self._scale_host, self._shift_host, pack=self.pair_ok)
            # 1x1 layers keep their fp32 filters on the host: conv3 + shortcut conv of a stage's first block are packed jointly (ops.bottleneck_proj_pack)
            self._w_tap_host = wp_tap.reshape(self.cout, taps, self.cin_pad).contiguous() if taps == 1 else None

    def to(self, device):
        self.w = self._w_host.to(device)
        if self._wp_host is not None:
            self.wp = self._wp_host.to(device)
        if self._w0_host is not None and FusedConv.winograd:
            self._w0 = self._w0_host.to(device)        # tap-major filters stay resident: a tile's filters are made on first use
            self.u = {}
            if FusedConv.winograd_tile:
                self._filters(FusedConv.winograd_tile)
        if not self._identity_epilogue:
            self.scale = self._scale_host.to(device)
            self.shift = self._shift_host.to(device)
        if self._wpair_host is not None:
            self.wpair = self._wpair_host.to(device)
            self.scale_p = self._scale_p_host.to(device)
        return self

    def __call__(self, x, res=None, res_mode=0, relu=None, naive=False, res_after_act=False, post_scale=1.0, out_pair=False):
        """out_pair (pair chain only): write the output as a PairTensor -- the caller knows whether every consumer takes one."""
        if self.w is None:
            raise RuntimeError('FusedConv.to(device) must be called before use')
        if isinstance(x, ops.PairTensor):
            return self._call_pio(x, res, res_mode, relu, naive, res_after_act, post_scale, out_pair)
        if isinstance(res, ops.PairTensor):
            raise TypeError('a pair residual needs a pair input (the fp32 kernels read fp32 residuals)')
        if self.dtype == FP8 or self.out_dtype == FP8 or isinstance(res, QTensor):
            return self._call_quantized(x, res, res_mode, relu, naive, res_after_act, post_scale)
        y = self._call(x, res, res_mode, relu, naive, res_after_act, post_scale)
        if FusedConv.calib is not None:
            FusedConv.calib[self.key] = max(FusedConv.calib.get(self.key, 0.0), float(y.float().abs().max()))
        return y

    def _call_quantized(self, x, res, res_mode, relu, naive, res_after_act, post_scale):
        """fp8 storage: x / res may be QTensors (e4m3 bytes + scale); the tensors' scales are folded into the epilogue vectors
        (scale = bn_scale * s_w[co] * s_in / s_out, shift = bn_shift / s_out, residual multiplier s_res / s_out)."""
        s_in = x.scale if isinstance(x, QTensor) else 1.0
        xd = x.data if isinstance(x, QTensor) else x
        if (self.dtype == FP8) != isinstance(x, QTensor):
            raise TypeError('an fp8 layer takes a QTensor (and only an fp8 layer does)')
        s_out = self.out_scale
        vec = self._qvec.get((s_in, s_out))
        if vec is None:
            sc = self._scale_host * (self.w_scale if self.w_scale is not None else 1.0) * (s_in / s_out)
            vec = self._qvec[(s_in, s_out)] = (sc.float().contiguous().to(self.w.device), (self._shift_host / s_out).float().contiguous().to(self.w.device))
        rd = res.data if isinstance(res, QTensor) else res
        res_scale = ((res.scale if isinstance(res, QTensor) else 1.0) / s_out) if res is not None else 1.0
        y = self._call(xd, rd, res_mode, relu, naive, res_after_act, post_scale, epi=(vec[0], vec[1], res_scale))
        return QTensor(y, s_out) if self.out_dtype == FP8 else y

    def _call(self, x, res, res_mode, relu, naive, res_after_act, post_scale, epi=None):
        """epi: (scale, shift, res_scale) of this call when they differ from the layer's own (the quantised modes: the tensors'
        scales folded in); the direct kernel only -- the Winograd form is fp32."""
        B = x.shape[0]
        m, xs, wk, wst, wpad = self.wino_tile(tuple(x.shape), x.dtype, res_mode, naive)
        wino = m > 0
        if not wino and epi is None and self.takes_pair_form(tuple(x.shape), x.dtype, naive):   # layers the Winograd form does not take
            return self._pair(x, res, res_mode, relu, res_after_act, post_scale)
        if not wino:
            m = FusedConv.winograd_tile or 6      # only used by the executed-FLOP accounting below (not reached: wino False)
        if FusedConv.count_flops:
            od, oh, ow = ((x.shape[1 + a] + 2 * self.padding[a] - self.kernel[a]) // self.stride[a] + 1 for a in range(3))
            direct = 2.0 * x.shape[0] * od * oh * ow * self.cout * self.cin * self.kernel[0] * self.kernel[1] * self.kernel[2]
            FusedConv.flops += direct
            t1, t2, t3 = (oh, ow, 1) if self._wino2d else (od, oh, ow)
            FusedConv.exec_flops += (2.0 * (m + 2) ** 2 * x.shape[0] * ((t1 + m - 1) // m) * ((t2 + m - 1) // m) * t3 * self.cout *
                                     self.cin * wk[2]) if wino else direct
        if wino:
            if FusedConv.trace is not None:
                ops.winograd_trace = []
            xv = x.view(xs)
            rv = None if res is None else res.view(B, res.shape[2], res.shape[3], 1, self.cout) if self._wino2d else res
            opnd = self._wino_operands(m)
            y = ops.conv_winograd_fwd(xv, self._filters(m, opnd), self.scale, self.shift, wk[2], wst[2], wpad, self.relu if relu is None else relu,
                                      rv, wgt_layout=self.layout, res_after_act=res_after_act, post_scale=post_scale, operands=opnd)
            if FusedConv.trace is not None:
                tiles = y.shape[0] * ((y.shape[1] + m - 1) // m) * ((y.shape[2] + m - 1) // m)
                v_bytes = 4.0 * (m + 2) ** 2 * tiles * xv.shape[3] * self.cin      # transformed input: (m+2)^2 planes [tiles, Z, Cin]
                m_bytes = 4.0 * (m + 2) ** 2 * tiles * y.shape[3] * self.cout      # (m+2)^2 partial outputs [tiles, Zo, Cout]
                by = {'input': 4.0 * x.numel() + v_bytes, 'gemm': v_bytes + m_bytes,
                      'output': m_bytes + 4.0 * y.numel() * (2 if res is not None else 1)}
                FusedConv.trace += [('wino_' + n, e0, e1, fl, by[n], x.shape[1] > 1 and x.shape[3] > 1, self._describe(x, m)) for n, e0, e1, fl in ops.winograd_trace]
                ops.winograd_trace = None
            return y.view(B, 1, y.shape[1], y.shape[2], self.cout) if self._wino2d else y
        if FusedConv.trace is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = self._direct(x, res, res_mode, relu, naive, res_after_act, post_scale, epi)
            e1.record()
            FusedConv.trace.append(('direct', e0, e1, 2.0 * y.numel() * self.cin * self.kernel[0] * self.kernel[1] * self.kernel[2]
                                    if self.out_mode == 0 else 2.0 * x.numel() * self.cout,
                                    float(x.numel() * x.element_size() + y.numel() * y.element_size() + self.w.numel() * self.w.element_size()
                                          + (res.numel() * res.element_size() if res is not None else 0)),     # algorithmic bytes
                                    x.shape[1] > 1 and x.shape[3] > 1, self._describe(x, 0)))
            return y
        return self._direct(x, res, res_mode, relu, naive, res_after_act, post_scale, epi)

    def _call_pio(self, x, res, res_mode, relu, naive, res_after_act, post_scale, out_pair):
        """the fp16-pair form on a PairTensor input (ops.conv_fwd_pio); out_pair needs Cout % 16 == 0"""
        if self.wpair is None:
            raise TypeError('this layer has no pair filters (FusedConv(chain=True) with Cin % 32 == 0)')
        tr = FusedConv.trace is not None
        fl = 2.0 * x.shape[0] * self.cout * self.cin * self.kernel[0] * self.kernel[1] * self.kernel[2]
        if tr:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        y = ops.conv_fwd_pio(x, self.wpair, self.scale_p, self.shift, self.kernel, self.stride, self.padding, self.relu if relu is None else relu,
                             self.wbound, self.sbound, res, res_mode, out_pair=bool(out_pair) and self.cout % 16 == 0, res_after_act=res_after_act,
                             post_scale=post_scale, naive=naive)
        npos = y.shape[1] * y.shape[2] * y.shape[3]
        if FusedConv.count_flops:
            FusedConv.flops += fl * npos
            FusedConv.exec_flops += 3.0 * fl * npos
        if tr:
            e1.record()
            FusedConv.trace.append(('direct', e0, e1, 3.0 * fl * npos,
                                    float(4 * x.numel() + 4 * y.numel() + 2 * self.wpair.numel() + (4 * res.numel() if res is not None else 0)),
                                    False, self._describe(x, 0) + ' pair'))
        return y

    def takes_pair_form(self, x_shape, dtype=torch.float32, naive=False):
        if self.wp is None or naive or dtype != torch.float32:
            return False
        npos = x_shape[0] * x_shape[1] * x_shape[2] * x_shape[3]
        if FusedConv.pair_mode < 0:       # the default rule (class comment; csrc/model.cpp plan_conv)
            ok = self._split_cand and FusedConv.wino_operands == ops.IVX_F16_PAIR and npos >= FusedConv.SPLIT_MIN_POS
        else:
            ok = FusedConv.pair_mode >= (1 if self._dims == 3 else 2) and npos >= FusedConv.pair_min_pos
        return ok and ops.conv_pair_supported(x_shape, self.cout, self.kernel, self.stride, self.padding, 1)

    def _pair(self, x, res, res_mode, relu, res_after_act, post_scale):
        """split pass (fp32 -> bf16 pairs) + the three-product bf16 MFMA kernel with the usual fused fp32 epilogue"""
        tr = FusedConv.trace is not None
        if FusedConv.count_flops:
            od, oh, ow = ((x.shape[1 + a] + 2 * self.padding[a] - self.kernel[a]) // self.stride[a] + 1 for a in range(3))
            direct = 2.0 * x.shape[0] * od * oh * ow * self.cout * self.cin * self.kernel[0] * self.kernel[1] * self.kernel[2]
            FusedConv.flops += direct
            FusedConv.exec_flops += 3.0 * direct
        if tr:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
        xp = ops.bf16_pair_split(x)
        if tr:
            e[1].record()
        y = ops.conv_fwd(xp, self.wp, self.scale, self.shift, self.kernel, self.stride, self.padding, self.relu if relu is None else relu, res,
                         res_mode, wgt_layout=1, res_after_act=res_after_act, post_scale=post_scale, pair=True)
        if tr:
            e[2].record()
            is3d = x.shape[1] > 1 and x.shape[3] > 1
            fl = 2.0 * y.numel() * self.cin * self.kernel[0] * self.kernel[1] * self.kernel[2]
            FusedConv.trace.append(('pair_split', e[0], e[1], 0.0, 8.0 * x.numel(), is3d, self._describe(x, 0) + ' pair'))
            FusedConv.trace.append(('pair_gemm', e[1], e[2], 3.0 * fl, float(4 * x.numel() + 4 * y.numel() + 2 * self.wp.numel()
                                                                            + (4 * res.numel() if res is not None else 0)), is3d,
                                    self._describe(x, 0) + ' pair'))
        return y

    def wino_tile(self, x_shape, dtype=torch.float32, res_mode=0, naive=False):
        """-> (m, xs, wk, wst, wpad): m = tile of the F(m x m, 3x3) form this layer takes for an input of shape x_shape
        (0: the direct kernel), and the convolution as the Winograd entry points see it (transformed axes first, direct
        axis last; a 2-D 3x3 layer [B,1,H,W,C] is the view [B,H,W,1,C] with a 3x3x1 kernel)."""
        B = x_shape[0]
        if self._wino2d:
            xs, wk, wst, wpad = (B, x_shape[2], x_shape[3], 1, self.cin), (3, 3, 1), (1, 1, 1), (self.padding[1], self.padding[2], 0)
        else:
            xs, wk, wst, wpad = tuple(x_shape), self.kernel, self.stride, self.padding
        ok = (self.u is not None and FusedConv.winograd and not naive and res_mode in (0, 1) and dtype == torch.float32
              and x_shape[0] * x_shape[1] * x_shape[2] * x_shape[3] >= FusedConv.winograd_min_pos)
        m = FusedConv.winograd_tile or (6 if (xs[1] + 2 * wpad[0] - 2) * (xs[2] + 2 * wpad[1] - 2) >= FusedConv.winograd_tile6_min_plane
                                        else 4)
        ok = ok and ops.conv_winograd_supported(xs, self.cout, wk, wst, wpad, m)
        return (m if ok else 0), xs, wk, wst, wpad

    # a 3x3 layer of the pair chain whose Winograd form (three launches on fp32 tensors, fp16 pair operands in the transformed domain)
    # beats its direct pair form: wide and on a large map, where the direct form is bound by its 5x as many matrix products (measured,
    # tools/pio_ab.py: 256 -> 256 at 120x160x50 views 2.25 vs 3.14 ms, at 20 views 1.00 vs 1.30; 256 -> 256 at 30x40x50 0.29 vs 0.21, 128 ->
    # 128 at 60x80x50 0.36 vs 0.24: the rule below takes only the FPN output conv of the FastIndoor configs).  Mirrored by csrc/model.cpp.
    WINO_OVER_PAIR_MIN_CH, WINO_OVER_PAIR_MIN_POS = 256, 200000

    def prefers_winograd(self, npos):
        return (self._wino2d and self.u is not None and FusedConv.winograd and FusedConv.wino_operands == ops.IVX_F16_PAIR
                and self.cin >= FusedConv.WINO_OVER_PAIR_MIN_CH and self.cout >= FusedConv.WINO_OVER_PAIR_MIN_CH
                and npos >= FusedConv.WINO_OVER_PAIR_MIN_POS and self.cin % 32 == 0)

    def _describe(self, x, tile):
        k, st = 'x'.join(map(str, self.kernel)), ''.join(map(str, self.stride))
        return f'{self.cin}->{self.cout} k{k} s{st} in {tuple(x.shape[:4])}' + (f' F{tile}' if tile else '')

    def _wino_operands(self, tile):
        ok = FusedConv.wino_operands == ops.IVX_F16_PAIR and tile >= 4 and self.cin % (32 if self.layout == 1 else 16) == 0
        return ops.IVX_F16_PAIR if ok else 0

    def _filters(self, tile, operands=None):
        operands = self._wino_operands(tile) if operands is None else operands
        if (tile, operands) not in self.u:
            self.u[(tile, operands)] = ops.conv_winograd_weights(self._w0, self.layout, tile, operands)
        return self.u[(tile, operands)]

    def _direct(self, x, res, res_mode, relu, naive, res_after_act, post_scale, epi=None):
        scale, shift, res_scale = epi if epi is not None else (self.scale, self.shift, 1.0)
        return ops.conv_fwd(x, self.w, scale, shift, self.kernel, self.stride, self.padding,
                            self.relu if relu is None else relu, res, res_mode, naive=naive, wgt_layout=self.layout,
                            out_mode=self.out_mode, res_after_act=res_after_act, post_scale=post_scale,
                            out_dtype=self.out_dtype, res_scale=res_scale)



class FusedConvTranspose2x(FusedConv):
    """nn.ConvTranspose3d(kernel 2, stride 2, bias=False) [+ eval BN] [+ ReLU] as ONE 1x1x1 GEMM with 8*Cout columns
    (column n = ((a*2+e)*2+f)*Cout + co) whose epilogue scatters to out[b, 2d+a, 2h+e, 2w+f, co]
    (necks/imvoxelnet.py:54-56: the up-blocks of FastIndoorImVoxelNeck).  weight: [Cin, Cout, 2, 2, 2]."""

    def __init__(self, weight, bn=None, relu=False, eps=1e-5, dtype=None, out_dtype=None):
        w = weight.detach().to(torch.float32)
        cin, cout = w.shape[0], w.shape[1]
        if tuple(w.shape[2:]) != (2, 2, 2):
            raise ValueError('only kernel 2 / stride 2 transposed convolutions are built')
        as_conv = w.permute(2, 3, 4, 1, 0).reshape(8 * cout, cin, 1, 1, 1)      # [(a,e,f,co), ci]
        super().__init__(as_conv, None, None, 1, 0, relu, dims=3, eps=eps, dtype=dtype, out_dtype=out_dtype)
        self.out_mode = 1
        self.cout_real = cout
        scale, shift = torch.ones(cout), torch.zeros(cout)
        if bn is not None:
            scale, shift = fold_batchnorm(bn, None, eps)
            self._identity_epilogue = False
        self._scale_host, self._shift_host = scale.contiguous(), shift.contiguous()
