"""Two-stream software pipeline for a chain of Winograd-form 3-D conv layers (the KITTI / nuScenes necks).

A layer in the F(m x m, 3x3) form is three launches: input transform (HBM-bound), grouped GEMM (MFMA-bound), output
transform (HBM-bound).  Run back to back, the transforms are ~28 % of the KITTI neck although the matrix cores idle
while they stream.  Samples of a batch are independent, so the batch is cut into `chunks` slices and the stages are
issued on two HIP streams:

    GEMM stream:       g(L,0)  g(L,1)      g(L+1,0)     g(L+1,1)   ...
    transform stream:          out(L,0)+in(L+1,0)   out(L,1)+in(L+1,1)   ...

i.e. while the matrix cores work on one slice, the transform kernels of the other slice stream through HBM.  The
transform launches use a capped grid (ivx_conv_winograd_set_transform_blocks) so they occupy few workgroup slots.
Every slice runs exactly the kernels of the sequential path on exactly its data: the result is bit-identical to
FusedConv.__call__ layer by layer (tests/test_gpu_kernels.py::test_pipelined_stack_equals_sequential).

Reference: the layer chain is necks/imvoxelnet.py:99-113 (KittiImVoxelNeck.model) / :131-145 (NuScenesImVoxelNeck).
"""
import os

import torch

from . import ops


def _env_int(name, default):
    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


# knobs.  MEASURED (profiles/r02_overlap.md): on MI355X the two kinds of kernel barely overlap -- issued together, a grouped
# GEMM and a transform pair take 80-95 % of the SUM of their stand-alone times whatever the grid cap, stream priority or GEMM
# occupancy, and the whole KITTI neck gains 0.2-1.8 % (24.84 -> 24.40 ms at best).  The pipeline therefore stays OFF by
# default (IVX_PIPE_CHUNKS=2 switches it on); it is kept because it is exact and costs nothing when off.
CHUNKS = _env_int('IVX_PIPE_CHUNKS', 0)            # 0 / 1 = sequential path
XF_BLOCKS = _env_int('IVX_PIPE_XF_BLOCKS', 2048)   # grid cap of the transform kernels while pipelined (0 = uncapped)
XF_PRIORITY = _env_int('IVX_PIPE_XF_PRIO', -1)     # stream priority of the transform stream (-1 = high, 0 = normal)

_streams = {}


def _xf_stream(device):
    key = (device.index if device.index is not None else torch.cuda.current_device(), XF_PRIORITY)
    if key not in _streams:
        _streams[key] = torch.cuda.Stream(device=device, priority=XF_PRIORITY)
    return _streams[key]


def pipelined_ok(convs, x, chunks):
    """All layers take the Winograd form at the per-chunk shape and the batch splits evenly."""
    if chunks < 2 or x.shape[0] % chunks or x.dtype != torch.float32:
        return False
    shape = (x.shape[0] // chunks,) + tuple(x.shape[1:])
    for f in convs:
        m, xs, wk, wst, wpad = f.wino_tile(shape)
        if m == 0 or f._wino2d:
            return False
        shape = ops.WinogradLayerPlan(xs, f.cout, wk[2], wst[2], wpad, f.relu, f.layout, m).oshape
    return True


def stack_forward_pipelined(convs, res_from, x, chunks=None, trace=None):
    """convs: FusedConv list (a chain); res_from[i]: None or the index k of the activation added as residual before the
    ReLU of layer i (k = 0: the input x, k = j + 1: the output of layer j).  x [B,X,Y,Z,C] channels-last fp32.
    Returns the output of the last layer.  trace (optional list): gets (kind, start, end, flops, bytes, True) tuples
    like FusedConv.trace, one per stage launch, the events recorded on the stream the stage ran on."""
    chunks = CHUNKS if chunks is None else chunks
    B = x.shape[0]
    Bc = B // chunks
    dev = x.device
    main = torch.cuda.current_stream(dev)
    side = _xf_stream(dev)
    # plans per layer (chunk shape) and full-batch activation buffers
    plans, acts = [], [x]
    shape = (Bc,) + tuple(x.shape[1:])
    for i, f in enumerate(convs):
        m, xs, wk, wst, wpad = f.wino_tile(shape)
        pl = ops.WinogradLayerPlan(xs, f.cout, wk[2], wst[2], wpad, f.relu, f.layout, m, has_res=res_from[i] is not None)
        plans.append(pl)
        shape = pl.oshape
        acts.append(torch.empty((B,) + tuple(pl.oshape[1:]), device=dev, dtype=torch.float32))
    ws_bytes = max(p.ws_bytes for p in plans)
    ws = [torch.empty((ws_bytes,), device=dev, dtype=torch.uint8) for _ in range(chunks)]
    us = [f._filters(p.tile) for f, p in zip(convs, plans)]

    def ev():
        return torch.cuda.Event(enable_timing=trace is not None)

    def traced(kind, fn, flops, nbytes):
        if trace is None:
            return fn()
        e0, e1 = ev(), ev()
        e0.record()
        fn()
        e1.record()
        trace.append((kind, e0, e1, flops, nbytes, True))

    n = len(convs)
    side.wait_stream(main)                      # x is ready on the main stream
    ops.winograd_set_transform_blocks(XF_BLOCKS)
    try:
        ev_in = [[None] * chunks for _ in range(n)]
        ev_g = [[None] * chunks for _ in range(n)]
        # prologue: input transforms of layer 0
        with torch.cuda.stream(side):
            for c in range(chunks):
                xc = acts[0][c * Bc:(c + 1) * Bc]
                traced('wino_input', lambda: plans[0].input(xc, ws[c]), 0.0, 4.0 * xc.numel() + plans[0].v_bytes)
                ev_in[0][c] = ev()
                ev_in[0][c].record()
        for i in range(n):
            for c in range(chunks):
                # GEMM of (layer i, chunk c) on the main stream
                main.wait_event(ev_in[i][c])
                traced('wino_gemm', lambda: plans[i].gemm(us[i], ws[c]), plans[i].gemm_flops, plans[i].v_bytes + plans[i].m_bytes)
                ev_g[i][c] = ev()
                ev_g[i][c].record()
                # bridge on the transform stream: output transform of (i, c), then input transform of (i + 1, c)
                with torch.cuda.stream(side):
                    side.wait_event(ev_g[i][c])
                    f = convs[i]
                    oc = acts[i + 1][c * Bc:(c + 1) * Bc]
                    rc = None if res_from[i] is None else acts[res_from[i]][c * Bc:(c + 1) * Bc]
                    traced('wino_output', lambda: plans[i].output(f.scale, f.shift, rc, oc, ws[c]), 0.0,
                           plans[i].m_bytes + 4.0 * oc.numel() * (2 if rc is not None else 1))
                    if i + 1 < n:
                        traced('wino_input', lambda: plans[i + 1].input(oc, ws[c]), 0.0, 4.0 * oc.numel() + plans[i + 1].v_bytes)
                        ev_in[i + 1][c] = ev()
                        ev_in[i + 1][c].record()
        main.wait_stream(side)                  # the last output transforms
    finally:
        ops.winograd_set_transform_blocks(0)
    return acts[-1]
