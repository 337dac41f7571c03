#!/usr/bin/env python
"""bench.py -- ImVoxelNet inference throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL; weak scaling, per-GPU batch fixed)

A step = one pass of the hot path (ImVoxelNet.simple_test: ResNet-50 + FPN -> unprojection -> KittiImVoxelNeck
-> Anchor3DHead -> decode + rotated NMS -> D2H of detections [-> all-gather of detections when N > 1]) over one
batch of synthetic KITTI-shaped input already resident in HBM.  Workload = BASELINE.json configs[1]:
1 view 3x384x1280, 216x248x12 voxels, batch 4 per GPU, fp32 (the reference's arithmetic type).

Prints ONE JSON line (rank 0) with `roofline` (the implicit-GEMM kernel over its launches on the 3-D neck -- executed
FLOPs, the neck runs in the Winograd F(6x6,3x3) form -- measured live with HIP events on the launch stream inside the timed
region), `roofline_winograd_transforms` (the HBM-bound transform kernels around those launches) and, at N == 1, `cpu_baseline` (the oracle's torch-fp32/C port of
the same path timed on this box's host cores on one image).
"""
import argparse
import json
import os

import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense peak (no sparsity)
BATCH_PER_GPU = 4


def measured_ceilings(dev, bf16=False):
    """Device ceilings measured on this box at start-up (csrc/ubench.hip): the dense issue rate of the MFMA form the conv kernel
    uses and the HBM streaming copy rate over 2 x 1 GiB buffers.  ~0.3 s; the buffers are freed before the model is built."""
    import ctypes as C
    from imvoxelnet_amd import _lib
    L = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    v = C.c_double()
    scratch = torch.empty((1 << 20,), device=dev, dtype=torch.uint8)
    for name, dt in (('mfma_f32_tflops', 0),) + ((('mfma_bf16_tflops', 1),) if bf16 else ()):
        _lib.check(L.ivx_ubench_mfma(dt, C.c_void_p(scratch.data_ptr()), scratch.numel(), C.byref(v), st), 'ivx_ubench_mfma')
        out[name] = round(v.value, 1)
    n = 1 << 30
    a, b = torch.empty((n,), device=dev, dtype=torch.uint8), torch.empty((n,), device=dev, dtype=torch.uint8)
    a.zero_()
    _lib.check(L.ivx_ubench_copy(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), n, C.byref(v), st), 'ivx_ubench_copy')
    out['hbm_copy_gbps'] = round(v.value, 1)
    del a, b, scratch
    torch.cuda.empty_cache()
    return out


def neck_flops_per_sample(nv, c_in, c_out):
    """Algorithmic FLOPs (2 x MAC) of KittiImVoxelNeck for one sample (2603.4 GFLOP at KITTI)."""
    X, Y, Z = nv
    c1, c2, c4 = c_in, c_in * 2, c_in * 4
    z2, z3 = (Z + 2 - 3) // 2 + 1, ((Z + 2 - 3) // 2 + 1 + 2 - 3) // 2 + 1
    macs = 2 * X * Y * Z * c1 * c1 * 27
    macs += X * Y * z2 * c2 * c1 * 27 + 2 * X * Y * z2 * c2 * c2 * 27
    macs += X * Y * z3 * c4 * c2 * 27 + 2 * X * Y * z3 * c4 * c4 * 27
    macs += (X - 2) * (Y - 2) * (z3 - 2) * c_out * c4 * 27
    return 2.0 * macs


def neck_algorithmic_bytes_per_sample(nv, c_in, c_out):
    """Bytes KittiImVoxelNeck's nine layers must move once per sample: every layer's fp32 input, output, residual (the second conv of a
    block) and filters -- what a direct convolution per layer would read and write (the Winograd form's V / M workspaces are NOT in it)."""
    X, Y, Z = nv
    c1, c2, c4 = c_in, c_in * 2, c_in * 4
    z2, z3 = (Z + 2 - 3) // 2 + 1, ((Z + 2 - 3) // 2 + 1 + 2 - 3) // 2 + 1
    v1, v2, v3 = X * Y * Z, X * Y * z2, X * Y * z3
    b = 0
    for v, c in ((v1, c1), (v2, c2), (v3, c4)):          # block: conv (in + out) and conv + residual (in + res + out), 2 x 27 c^2 filters
        b += 4 * (2 * v * c + 3 * v * c + 2 * 27 * c * c)
    b += 4 * (v1 * c1 + v2 * c2 + 27 * c1 * c2) + 4 * (v2 * c2 + v3 * c4 + 27 * c2 * c4)        # the two z-stride-2 convs
    b += 4 * (v3 * c4 + (X - 2) * (Y - 2) * (z3 - 2) * c_out + 27 * c4 * c_out)                 # the last, pad-0 conv
    return float(b)


def bench_lift(args, ia, kc, dev):
    """Unprojection-only stress (BASELINE configs 4 / 5: "HBM-bound gather stress"): the fused multi-view lift alone on
    synthetic FPN maps, timed with HIP events over `steps` launches.  Algorithmic bytes per scene =
    features read once + volume written once + valid mask."""
    from imvoxelnet_amd import ops
    spec = {'lift_nuscenes': (6, 64, (232, 400), (192, 192, 32), (.32, .32, .32), lambda: kc.nuscenes_meta(box_type=ia.LiDARInstance3DBoxes), 1),
            'lift_scannet': (50, 64, (120, 160), (80, 80, 32), (.08, .08, .08), lambda: kc.indoor_meta(50, box_type=ia.DepthInstance3DBoxes), 2)}[args.config]
    V, Cf, (fh, fw), nv, vs, mk, B = spec
    esz = 2 if args.storage == 'bf16' else 4
    model = type('M', (), {'_compute_projection': staticmethod(ia.ImVoxelNet._compute_projection)})()   # camera set-up only
    model.n_voxels, model.voxel_size = nv, vs
    metas = [mk() for _ in range(B)]
    proj, origin, crop = ia.ImVoxelNet._camera_setup(model, metas, 4, dev)
    feat = torch.randn(B * V, 1, fh, fw, Cf, generator=torch.Generator().manual_seed(7)).to(dev)
    if args.storage == 'bf16':
        feat = feat.to(torch.bfloat16)
    for _ in range(max(1, args.warmup)):
        vol, valid = ops.backproject_mean(feat, proj, origin, crop, vs, nv)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.steps):
        vol, valid = ops.backproject_mean(feat, proj, origin, crop, vs, nv)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    n = nv[0] * nv[1] * nv[2]
    by = B * (V * Cf * fh * fw * esz + Cf * n * esz + n)
    print(json.dumps({'metric': f'scenes/sec, unprojection only ({args.config}: {V} views {Cf}x{fh}x{fw} -> {"x".join(map(str, nv))})',
                      'value': round(B / (ms * 1e-3), 1), 'unit': 'scenes/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
                      'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.storage,
                      'data': 'synthetic', 'config': {'workload': args.config, 'views': V, 'batch_per_gpu': B,
                                                      'valid_fraction': round(float(valid.float().mean()), 4)},
                      'roofline': {'bound': 'hbm', 'kernel': 'backproject_mean_kernel', 'achieved': round(by / (ms * 1e-3) / 1e9, 1), 'peak': 8000.0,
                                   'unit': 'GB/s', 'frac': round(by / (ms * 1e-3) / 8e12, 4), 'traffic': None,
                                   'algorithmic_MB_per_scene': round(by / B / 1e6, 1)}}))


def bench_other(args, ia, kc, dev, rank, world, emit=True):
    """Single-process throughput of the other BASELINE.json workloads (parity-test configurations; not the headline
    metric).  Same timing method; the roofline entry is the conv kernel over the 3-D neck with FLOPs counted per call."""
    from imvoxelnet_amd.conv import FusedConv
    spec = {
        'nuscenes': (kc.nuscenes_model_cfg(), kc.NUSCENES_TEST_CFG, 6, (928, 1600), 1, lambda: kc.nuscenes_meta(box_type=ia.LiDARInstance3DBoxes)),
        'scannet_fast': (kc.scannet_fast_model_cfg(), kc.SCANNET_FAST_TEST_CFG, args.views or 50, (480, 640), 1,
                         lambda: kc.indoor_meta(args.views or 50, box_type=ia.DepthInstance3DBoxes)),
        'sunrgbd_fast': (kc.sunrgbd_fast_model_cfg(), kc.SUNRGBD_FAST_TEST_CFG, 1, (480, 640), 1,
                         lambda: kc.indoor_meta(1, origin=(0, 3, -1), box_type=ia.DepthInstance3DBoxes)),
        'scannet_v1': (kc.scannet_v1_model_cfg(), kc.SCANNET_V1_TEST_CFG, args.views or 50, (480, 640), 1,
                       lambda: kc.indoor_meta(args.views or 50, box_type=ia.DepthInstance3DBoxes)),
    }[args.config]
    cfg, tcfg, V, (H, W), B, mk = spec
    B = args.batch if args.batch != BATCH_PER_GPU else B
    model = ia.build_detector(cfg, test_cfg=tcfg)
    ia.randomize_(model, 0)
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        if hasattr(model.bbox_head, 'conv_cls'):
            model.bbox_head.conv_cls.weight.normal_(0, 0.02, generator=g)
            model.bbox_head.conv_cls.bias.fill_(-2.0)
            model.bbox_head.conv_reg.weight.normal_(0, 0.002, generator=g)
        else:
            model.bbox_head.cls_conv.weight.normal_(0, 0.01, generator=g)
            model.bbox_head.cls_conv.bias.fill_(-2.0)
            model.bbox_head.reg_conv.weight.normal_(0, 0.002, generator=g)
    model.prepare(dev, dtype=torch.bfloat16 if args.storage == 'bf16' else torch.float32)
    fp8_note = None
    # view sharding: every rank holds the same scene(s) and works on its slice of the views
    img = torch.randn(B, V, 3, H, W, generator=torch.Generator().manual_seed(1000 + (0 if args.shard == 'views' else rank))).to(dev)
    metas = [mk() for _ in range(B)]
    if args.trunk_fp8:
        if args.storage != 'bf16':
            raise SystemExit('--trunk-fp8 goes on top of --storage bf16 (BASELINE config 5: "bf16 with fp8 2D-conv MFMA")')
        model.calibrate_fp8(img, stages=args.fp8_stages, residual=args.fp8_residual, variant=args.fp8_variant)   # one bf16 pass over the batch: per-tensor activation scales
        fp8_note = ('ResNet-50 %s stored as e4m3 (calibrated per-tensor / per-channel scales; stages: %s; residual stream: %s), v_mfma_f32_32x32x16_fp8_fp8'
                    % ('activations and weights' if args.fp8_residual == 'fp8' else 'bottleneck interiors (conv1 / conv2 outputs, conv2 / conv3 weights)',
                       'all' if args.fp8_stages is None else args.fp8_stages, args.fp8_residual))
    n = args.steps + args.warmup
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    neck_flops, neck_exec = [0.0], [0.0]    # direct-convolution FLOPs / FLOPs executed (fewer for Winograd-form layers)

    view_sharded = args.shard == 'views'

    trunk_tr = [[] for _ in range(n)]        # per step: stage events of the 2-D trunk's conv launches

    def step(i, traced=False):
        FusedConv.trace = trunk_tr[i] if traced else None
        if view_sharded and args.exchange != 'all_reduce' and hasattr(model.neck_3d, 'strides') and isinstance(model.bbox_head, ia.Anchor3DHead):
            FusedConv.trace = None               # reduce-scatter over x-slabs: the neck runs on this rank's slab inside the call
            ev[i][0].record(); ev[i][1].record()
            res = model.simple_test_view_sharded(img, metas, exchange='reduce_scatter' if multi else 'all_reduce')
            return [(r['boxes_3d'].tensor, r['scores_3d'], r['labels_3d']) for r in res]
        if view_sharded:
            from imvoxelnet_amd.dist import view_sharded_lift
            vol, valid = view_sharded_lift(model, img, metas)       # 2-D trunk + partial lift on this rank's views, all-reduce
        else:
            p0 = model.features_2d_cl(img)
            vol, valid = model.lift_cl(p0, metas)
        FusedConv.trace = None
        ev[i][0].record()
        FusedConv.flops, FusedConv.exec_flops, FusedConv.count_flops = 0.0, 0.0, True
        y = model.neck_3d.forward_cl(vol)
        FusedConv.count_flops = False
        neck_flops[0], neck_exec[0] = FusedConv.flops, FusedConv.exec_flops
        ev[i][1].record()
        if isinstance(model.bbox_head, ia.Anchor3DHead):
            h = model.bbox_head.forward_cl(y)
            out = model.bbox_head.get_bboxes_cl(h, y.shape[2], y.shape[1], metas, hw_transposed=True)
            return model.bbox_head._wrap(*out, metas)
        res = model.bbox_head.get_bboxes_cl(model.bbox_head.forward_cl(y), valid, metas)
        return [(b.tensor.cpu(), s.cpu(), l.cpu()) for b, s, l in res]

    multi = dist.is_available() and dist.is_initialized()
    # N > 1: --shard samples (default; BASELINE configs 4 / 5: every rank its own scenes, one all-gather of padded detections per step,
    # weak scaling) needs the public call through the native handle; --shard views (one scene set, views split, strong scaling)
    if multi and not view_sharded and not (args.api == 'simple_test' and getattr(model, '_native', None) is not None and model.head_2d is None):
        raise SystemExit('--config other than kitti on N > 1 ranks: --shard samples needs the native handle (fp32 storage), or use --shard views')
    # default: the public call.  model.simple_test runs trunk + unprojection + neck (+ the anchor head and tail for nuScenes)
    # through the native model handle; stage times come from its coarse trace (neck stages individually, trunk as one span)
    public = args.api == 'simple_test' and getattr(model, '_native', None) is not None and not view_sharded
    traced_run = os.environ.get('IVX_BENCH_TRACE', '1') != '0'
    if public:
        step(0)                                  # one composed step: the neck's direct / executed FLOP counts

        def step(i, traced=False):               # noqa: F811
            res = model.simple_test(img, metas, gather=multi)
            if res is None:                      # N > 1: the collected list exists on rank 0 only (as collect_results)
                return []
            return [(r['boxes_3d'].tensor, r['scores_3d'], r['labels_3d']) for r in res]
        model._native.trace(0)      # the timed steps run without stage events: these steps are short (4-30 ms, ~100 launches)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(args.warmup + i)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rccl_ranks, rank_ms = None, None
    if multi:
        # self-check of the N > 1 line, as on the KITTI line: every rank's own step time and the rank count an actual all-reduce sees
        tl = torch.tensor([dt], device=dev, dtype=torch.float64)
        allt = torch.empty((dist.get_world_size(),), device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allt, tl)
        rank_ms = [round(float(v) / args.steps * 1e3, 3) for v in allt.cpu()]
        ones = torch.ones((1,), device=dev, dtype=torch.float32)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rccl_ranks = int(ones.item())
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if rank != 0:
            return
    trunk_note = ''
    if public and not traced_run:                # throughput only
        neck_ms, t2d, t2d_ms, t2d_flops, nt = 1e9, [], 0.0, 0.0, 0
    elif public and multi:                       # (the gathered step holds a collective: no extra single-rank steps)
        neck_ms, t2d, t2d_ms, t2d_flops, nt = 1e9, [], 0.0, 0.0, 0
    elif public:
        nt = min(3, args.steps)                  # stage events (native coarse trace) in extra steps after the timed region
        model._native.trace(1)
        for k in range(nt):
            step(k)
        torch.cuda.synchronize()
        recs = model._native.trace_records()
        model._native.trace(0)
        n_per = len(recs) // nt
        neck_ms = t2d_ms = t2d_flops = 0.0
        for k in range(nt):
            rows = recs[k * n_per:(k + 1) * n_per]
            t3 = [r for r in rows if r['is3d'] and r['stage'] <= 3]
            neck_ms += (max(r['start_ms'] + r['ms'] for r in t3) - min(r['start_ms'] for r in t3)) / nt
            t2d_ms += sum(r['ms'] for r in rows if r['stage'] == 6) / nt
            t2d_flops += sum(r['flops'] for r in rows if r['stage'] == 6) / nt
        t2d = [1]
        trunk_note = 'one event pair around the whole trunk (native coarse trace) in %d extra steps after the timed region' % nt
    else:
        neck_ms = sum(ev[args.warmup + i][0].elapsed_time(ev[args.warmup + i][1]) for i in range(args.steps)) / args.steps
        # 2-D trunk roofline: per-launch events would make these short, host-bound steps slower, so the trunk's launches are
        # bracketed in `nt` EXTRA steps after the timed region (same inputs, same kernels)
        nt = 0 if multi else min(3, args.steps)          # (the view-sharded step holds a collective: all ranks or none)
        for k in range(nt):
            step(k, traced=True)
        torch.cuda.synchronize()
        t2d = [t for k in range(nt) for t in trunk_tr[k]]
        t2d_ms = sum(t[1].elapsed_time(t[2]) for t in t2d) / max(nt, 1)
        t2d_flops = sum(t[3] for t in t2d if t[0] in ('direct', 'wino_gemm')) / max(nt, 1)
        trunk_note = '%d launches/step, event-bracketed incl. their Winograd transform / split-K passes, in %d extra steps after the timed region' % (
            len(t2d) // max(nt, 1), nt)
    ach = neck_exec[0] / (neck_ms * 1e-3) / 1e12      # executed FLOPs over the whole neck time (transform kernels included)
    # operand modes of this run: fp16 (hi, lo) pairs issue THREE 16-bit MFMA products per fp32 multiply-add and are priced against the dense
    # 16-bit peak; fp32 / bf16 storage against their own MFMA peaks
    pair_neck = FusedConv.wino_operands == 4 and args.storage != 'bf16'
    pair_trunk = FusedConv.trunk_operands == 4 and args.storage != 'bf16'
    pk = PEAK_BF16_MFMA_TFLOPS if args.storage == 'bf16' else PEAK_F32_MFMA_TFLOPS
    nk_mul, nk_pk = (3.0, PEAK_BF16_MFMA_TFLOPS) if pair_neck else (1.0, pk)
    neck_roof = {'bound': 'mfma',
                 'kernel': ('Winograd-domain GEMMs on fp16 (hi, lo) pair operands (conv_wino_halo_kernel / conv_igemm_v4_kernel<pair>) + their transforms + the '
                            'direct layers of the 3-D neck' if pair_neck else 'conv_igemm_v4_kernel<%s> (3-D neck)' % ('__bf16' if args.storage == 'bf16' else 'float')),
                 'flops_counted': ('executed multiply-adds x 3 (every fp16 product issued), over the whole neck time incl. the transform kernels' if pair_neck
                                   else 'executed FLOPs over the whole neck time incl. the transform kernels'),
                 'achieved': round(ach * nk_mul, 2), 'peak': nk_pk, 'unit': 'TFLOP/s', 'frac': round(ach * nk_mul / nk_pk, 4),
                 'fp32_equivalent_tflops': round(ach, 2), 'traffic': None,
                 'neck_gflop': round(neck_exec[0] / 1e9, 1), 'neck_direct_gflop': round(neck_flops[0] / 1e9, 1),
                 'direct_equivalent_tflops': round(neck_flops[0] / (neck_ms * 1e-3) / 1e12, 2), 'neck_ms_per_step': round(neck_ms, 3)}
    trunk_roof = None
    if t2d:
        t_ach = t2d_flops / (t2d_ms * 1e-3) / 1e12
        t_mul, t_pk = (3.0, PEAK_BF16_MFMA_TFLOPS) if pair_trunk else (1.0, pk)
        trunk_roof = {'bound': 'mfma', 'kernel': 'conv_igemm_v4_kernel%s (ResNet-50 + FPN level 0 over %d views; %s)' % (
                          '<fp16 pair activations chained between the layers>' if pair_trunk else '', B * V, trunk_note),
                      'flops_counted': ('every fp16 MFMA product issued: 3 per fp32 multiply-add (the stem alone runs on fp32 MFMA); most of these layers are bound by '
                                        'HBM / launch latency, not by the matrix pipe' if pair_trunk else 'executed FLOPs'),
                      'achieved': round(t_ach * t_mul, 2), 'peak': t_pk, 'unit': 'TFLOP/s', 'frac': round(t_ach * t_mul / t_pk, 4),
                      'fp32_equivalent_tflops': round(t_ach, 2), 'ms_per_step': round(t2d_ms, 3), 'executed_gflop_per_step': round(t2d_flops / 1e9, 1)}
    rec = {'metric': f'images/sec/node ({args.config}: {V} view(s) 3x{H}x{W}, {"x".join(map(str, cfg["n_voxels"]))} vox)',
           'value': round(B * V * (1 if view_sharded else world) * args.steps / dt, 3), 'unit': 'images/s',
           'scenes_per_s': round(B * (1 if view_sharded else world) * args.steps / dt, 3), 'n_gpus': world,
           'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
           'scaling': 'strong' if view_sharded else 'weak', 'vs_baseline': None, 'dtype': args.storage + ((' + fp8 2-D conv (bf16 residual stream)' if args.fp8_residual == 'bf16' else ' + fp8 trunk storage') if args.trunk_fp8 else ''), 'data': 'synthetic',
           'config': {'workload': args.config, 'views': V, 'batch_per_gpu': B, 'global_batch': B * (1 if view_sharded else world), 'parallelism': f'dp{world}' if not view_sharded else f'views/{world}',
                      'shard': args.shard, 'rccl_ranks': rccl_ranks, 'ms_per_step_by_rank': rank_ms,
                      'collective': (None if not multi else ('reduce-scatter of the partial volume over x-slabs (+ halo) + all-gather of the neck rows' if (view_sharded and args.exchange != 'all_reduce' and hasattr(model.neck_3d, 'strides'))
                                                             else ('all-reduce of the partial volume sums / view counts' if view_sharded else 'one all_gather_into_tensor of padded detections per step (RCCL)'))),
                      'trunk_fp8': fp8_note, 'api': 'simple_test (native handle)' if public else 'composed',
                      'wino_operands': 'fp16 pairs' if (FusedConv.wino_operands == 4 and args.storage != 'bf16') else 'storage type',
                      'detections_last_step': int(sum(len(r[1]) for r in last))},
           'roofline': neck_roof,
           'roofline_trunk_2d': None if not t2d else trunk_roof}
    if emit:
        print(json.dumps(rec))
    return rec


def self_launch(n):
    """Re-exec this command under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1, a free port)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def bench_dry(args, rank, world):
    """--dry: the launch / sharding / collection plumbing of the N > 1 path without a GPU (CPU test with --backend gloo), for the
    workload `--config` names: every rank takes its slice of a synthetic global batch of detections (max_num rows of that config), the
    slices are all-gathered exactly as in the timed step, rank 0 checks the reassembled batch and prints the JSON line (value is NOT a
    measurement).  With --shard views the exchange of the view-sharded mode runs on synthetic partial volumes in BOTH forms -- the
    all-reduce and the reduce-scatter over x-slabs with the neck's halo (dist.exchange_volume_slabs) -- and every rank checks that its
    slab of the latter equals the same rows of the former, then the neck rows are all-gathered (dist.all_gather_rows)."""
    from imvoxelnet_amd import dist as ivx_dist
    B = args.batch
    M = {'kitti': 50, 'nuscenes': 500, 'scannet_fast': 3000, 'scannet_v1': 3000, 'sunrgbd_fast': 1000}.get(args.config, 50)
    g = torch.Generator().manual_seed(7)
    gb, gs = torch.randn(B * world, M, 7, generator=g), torch.rand(B * world, M, generator=g)
    gl, gc = torch.randint(0, 3, (B * world, M), generator=g), torch.randint(0, M + 1, (B * world,), generator=g, dtype=torch.int32)
    a, b = ivx_dist.shard_range(B * world, rank, world)
    multi = dist.is_available() and dist.is_initialized()
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        boxes, scores, labels, count = ivx_dist.all_gather_detections(gb[a:b], gs[a:b], gl[a:b], gc[a:b])
    if multi:
        dist.barrier()
    dt = time.perf_counter() - t0
    ok = bool(torch.equal(boxes, gb) and torch.equal(scores, gs) and torch.equal(labels, gl) and torch.equal(count, gc))
    slab_ok, slab_note = None, None
    if args.shard == 'views':
        import imvoxelnet_amd as ia
        neck = (ia.KittiImVoxelNeck if args.config == 'kitti' else ia.NuScenesImVoxelNeck)(8, 16)
        X, Y, Z, C = 48, 5, 4, 8
        plans = [ivx_dist.StackNeckSlabs(neck, X, world, r) for r in range(world)]
        gp = torch.Generator().manual_seed(100 + rank)                     # this rank's partial view sum / view count over the whole volume
        part = torch.randn(B, X, Y, Z, C, generator=gp)
        cnt = torch.randint(0, 3, (B, X, Y, Z), generator=gp, dtype=torch.int32)
        sv, sc = ivx_dist.exchange_volume_slabs(part, cnt, plans, rank=rank)
        full_v, full_c = ivx_dist.all_reduce_volume(part.clone(), cnt.clone())
        me = plans[rank]
        # the rank-ordered sum of the slab form against the all-reduce's: equal to fp32 rounding of the additions (bit-equal at 2 ranks)
        slab_ok = bool(torch.equal(sc, full_c[:, me.ea:me.eb]) and torch.allclose(sv, full_v[:, me.ea:me.eb], rtol=0, atol=1e-5)
                       and (world > 2 or torch.equal(sv, full_v[:, me.ea:me.eb])))
        rows = torch.arange(me.oa, me.ob, dtype=torch.float32).reshape(1, -1, 1, 1, 1).expand(B, -1, 3, 1, 2).contiguous()
        allrows = ivx_dist.all_gather_rows(rows, plans, rank=rank)
        slab_ok = slab_ok and bool(torch.equal(allrows[0, :, 0, 0, 0], torch.arange(me.Xo, dtype=torch.float32)))
        flag = torch.tensor([1.0 if slab_ok else 0.0])
        if multi:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        slab_ok = bool(flag.item() == 1.0)
        slab_note = 'x-slabs (oa, ob, ea, eb) of a %d-row volume: %s' % (X, [(p.oa, p.ob, p.ea, p.eb) for p in plans])
    if rank == 0:
        print(json.dumps({'metric': 'images/sec/node (KITTI 3x384x1280, 216x248x12 vox)' if args.config == 'kitti' else f'images/sec/node ({args.config})',
                          'value': 0.0, 'unit': 'images/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / (args.steps + args.warmup) * 1e3, 3),
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                          'dry': True, 'gathered_ok': ok, 'slab_exchange_ok': slab_ok,
                          'config': {'workload': 'plumbing only (no GPU work): ' + args.config, 'batch_per_gpu': B, 'global_batch': B * world,
                                     'parallelism': f'dp{world}', 'backend': args.backend, 'max_num': M, 'shard': args.shard, 'slabs': slab_note}}))
    if multi:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit('dry run: the all-gathered batch differs from the global batch')
    if slab_ok is False:
        raise SystemExit('dry run: the reduce-scatter exchange over x-slabs differs from the all-reduce form')


def physical_cores():
    """Distinct (package, core) pairs of the host (`cores` of the CPU legs is the THREAD count used; SMT siblings share a core)."""
    try:
        seen, pkg, core = set(), None, None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                pkg = line.split(':')[1].strip()
            elif line.startswith('core id'):
                core = line.split(':')[1].strip()
            elif not line.strip():
                if core is not None:
                    seen.add((pkg, core))
                pkg = core = None
        if core is not None:
            seen.add((pkg, core))
        return len(seen) or None
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='samples per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-cpu-cabi', action='store_true', help='skip the second CPU leg (the C-ABI over the CPU restatement, oracle/cpu_abi)')
    ap.add_argument('--cpu-repeats', type=int, default=3, help='timed images of the cpu_baseline leg (after one warm-up image)')
    ap.add_argument('--config', default='kitti', choices=['kitti', 'nuscenes', 'scannet_fast', 'sunrgbd_fast', 'scannet_v1', 'lift_nuscenes', 'lift_scannet'],
                    help='BASELINE.json workload; the headline metric is quoted on kitti (configs[1]), the default')
    ap.add_argument('--views', type=int, default=0, help='views per scene for the indoor configs (default: reference test value)')
    ap.add_argument('--shard', default='samples', choices=['samples', 'views'],
                    help="multi-GPU partition: 'samples' (default; weak scaling, the headline mode) or 'views' (indoor multi-view "
                         "configs: the views of each scene are split over the ranks, one RCCL all-reduce of the partial volume; strong scaling)")
    ap.add_argument('--fp8-stages', type=int, default=None, help='--trunk-fp8: how many leading ResNet stages store e4m3 (default: all four)')
    ap.add_argument('--fp8-residual', default='bf16', choices=['bf16', 'fp8'], help="--trunk-fp8: 'bf16' (default) keeps the residual stream in bf16 and stores the bottleneck interiors as e4m3; 'fp8' stores every trunk activation as e4m3 (bandwidth stress mode, ~10 %% feature noise)")
    ap.add_argument('--fp8-variant', default='conv3', choices=['conv3', 'full'], help="--trunk-fp8: 'conv3' (default) = e4m3 on conv3 of ResNet stages 3 - 4 only (FPN within ~2.2 %% of fp32); 'full' = every bottleneck interior e4m3 (3.6 %%)")
    ap.add_argument('--trunk-fp8', action='store_true', help='with --storage bf16 and an indoor --config: e4m3 storage of the 2-D trunk (calibrated on the bench batch)')
    ap.add_argument('--storage', default='f32', choices=['f32', 'bf16'],
                    help='f32 (default) = the reference precision and the headline metric; bf16 = optional reduced-precision '
                         'storage mode (kitti only), reported with dtype "bf16" and priced against the bf16 MFMA peak')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help="process-group backend: 'nccl' (= RCCL; GPU runs) or 'gloo' (CPU plumbing test with --dry)")
    ap.add_argument('--dry', action='store_true', help='no GPU work: exercise the N-rank launch, batch sharding and detection all-gather only')
    ap.add_argument('--api', default='simple_test', choices=['simple_test', 'composed'],
                    help="what a timed step calls: 'simple_test' (default) = the drop-in public call ImVoxelNet.simple_test(img, img_metas) "
                         "(reference: tools/benchmark.py:74 model(return_loss=False, rescale=True, **data)); 'composed' = the same stages "
                         "called one by one with the packed D2H of round 1")
    ap.add_argument('--wino-operands', default='f16pair', choices=['f16pair', 'f32'],
                    help="operands of the Winograd-domain GEMMs of the neck: 'f16pair' (default) = every fp32 value as an fp16 (hi, lo) pair, three "
                         "fp16 MFMA products per pair, fp32 accumulation (22-bit operands: error at the level of the fp32 form's own rounding); "
                         "'f32' = fp32 MFMA (exact fp32 products).  The default run also times the other mode after the timed region and "
                         "reports it as exact_fp32_mfma")
    ap.add_argument('--exchange', default='auto', choices=['auto', 'all_reduce', 'reduce_scatter'],
                    help="--shard views: 'all_reduce' = whole partial volumes all-reduced, neck replicated; 'reduce_scatter' / 'auto' = x-slabs (+ halo) of the "
                         "partial volumes exchanged, every rank convolves its slab (stack necks with an anchor head: nuScenes), neck rows all-gathered")
    ap.add_argument('--trunk-operands', default='f16pair', choices=['f16pair', 'f32'],
                    help="2-D trunk (ResNet-50 + FPN): 'f16pair' (default) = activations chained as fp16 (hi, lo) pair tensors with device-side "
                         "power-of-two scales (ivx_conv_fwd_pio: three fp16 MFMA products per multiply-add, fp32 accumulation, no conversion "
                         "passes); 'f32' = fp32 MFMA")
    args = ap.parse_args()

    # One process per GPU.  Under torch.distributed.run (the reference's launcher is tools/dist_test.sh:9-10,
    # `torch.distributed.launch --nproc_per_node=$GPUS`) RANK / WORLD_SIZE are set; a bare `python bench.py --gpus N`
    # launches the N ranks itself.
    if args.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        return self_launch(args.gpus)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    force_dist = os.environ.get('IVX_BENCH_FORCE_DIST') == '1' and 'RANK' in os.environ   # exercise the RCCL path at world size 1
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'gloo':
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            try:
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
            except TypeError:      # older torch: no device_id argument
                dist.init_process_group('nccl', rank=rank, world_size=world)
        world = dist.get_world_size()      # n_gpus reported = the ranks the process group actually has
    elif not args.dry:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print(f'# note: --gpus {args.gpus} but the process group has {world} rank(s); reporting {world}', file=sys.stderr)
    if args.dry:
        return bench_dry(args, rank, world)
    dev = torch.device('cuda', torch.cuda.current_device())

    import imvoxelnet_amd as ia
    from imvoxelnet_amd import dist as ivx_dist, ops
    from imvoxelnet_amd import workloads as kc
    from imvoxelnet_amd.workloads import kitti_model_cfg, KITTI_TEST_CFG, kitti_meta
    from imvoxelnet_amd.conv import FusedConv

    if os.environ.get('IVX_NARROW_EPILOGUE') == '1':       # A/B of the conv epilogue (default: LDS-transposed wide stores)
        from imvoxelnet_amd import _lib
        _lib.lib().ivx_conv_set_epilogue_mode(1)
    if args.wino_operands == 'f32':
        FusedConv.wino_operands = 0            # fp32 MFMA in the Winograd domain for every workload of this run
    if args.trunk_operands == 'f32':
        FusedConv.trunk_operands = 0           # fp32 MFMA in the 2-D trunk
    if args.config.startswith('lift_'):
        return bench_lift(args, ia, kc, dev)
    if args.config != 'kitti':
        bench_other(args, ia, kc, dev, rank, world)
        if world > 1 or force_dist:
            dist.destroy_process_group()
        return
    if args.trunk_fp8:
        raise SystemExit('--trunk-fp8 is wired for the multi-view indoor configs (--config scannet_v1 | scannet_fast), where the 2-D trunk is the step')
    model = ia.build_detector(kitti_model_cfg(), test_cfg=KITTI_TEST_CFG)
    ia.randomize_(model, 0)
    with torch.no_grad():   # trained-net-like head statistics so the NMS tail has real work
        model.bbox_head.conv_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(5))
        model.bbox_head.conv_cls.bias.fill_(-2.0)
        model.bbox_head.conv_reg.weight.normal_(0, 0.002, generator=torch.Generator().manual_seed(6))
        model.bbox_head.conv_dir_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(7))
    bf16 = args.storage == 'bf16'
    pair = args.wino_operands == 'f16pair' and not bf16
    FusedConv.wino_operands = 4 if pair else 0
    ceil = measured_ceilings(dev, bf16 or pair) if os.environ.get('IVX_BENCH_UBENCH', '1') != '0' else {}
    model.prepare(dev, dtype=torch.bfloat16 if bf16 else torch.float32)
    # the pair form issues fp16 MFMAs (same rate as bf16): priced against the dense 16-bit MFMA peak, counting every product it issues
    peak = PEAK_BF16_MFMA_TFLOPS if (bf16 or pair) else PEAK_F32_MFMA_TFLOPS
    peak_meas = ceil.get('mfma_bf16_tflops' if (bf16 or pair) else 'mfma_f32_tflops')
    hbm_meas = ceil.get('hbm_copy_gbps')
    esz = 2 if bf16 else 4

    B = args.batch
    g = torch.Generator().manual_seed(1000 + rank)
    img_host = torch.randn(B, 1, 3, 384, 1280, generator=g)
    img = img_host.to(dev)
    metas = [kitti_meta(t=(0.01 * b, 0.0, 0.0), box_type=ia.LiDARInstance3DBoxes) for b in range(B)]

    multi = world > 1 or force_dist
    nsteps = args.steps + args.warmup
    traces = [[] for _ in range(nsteps)]      # per step: the conv stage launches (FusedConv.trace)
    lifts = [[] for _ in range(nsteps)]       # per step: the unprojection launch (ops.stage_trace)

    native_trace = False
    if args.api == 'simple_test':
        native_trace = model._native is not None and os.environ.get('IVX_BENCH_TRACE', '1') != '0'   # stage events are recorded inside the native handle
        if native_trace:
            model._native.trace(1)     # coarse: neck stages, unprojection, tail individually; the 2-D trunk as one span

        def step(i):
            """The drop-in call, as tools/benchmark.py:74 times it: returns the list of result dicts on the host."""
            FusedConv.trace, ops.stage_trace = traces[i], lifts[i]
            try:
                return model.simple_test(img, metas, gather=multi)
            finally:
                FusedConv.trace, ops.stage_trace = None, None

        def n_det(out):
            return int(sum(len(r['scores_3d']) for r in out)) if out is not None else 0
    else:
        def step(i):
            FusedConv.trace, ops.stage_trace = traces[i], lifts[i]
            p0 = model.features_2d_cl(img)
            proj, new_origin, crop = model._camera_setup(metas, 4, p0.device)   # host camera set-up + 3 small H2D copies
            vol, _ = ops.backproject_mean(p0, proj, new_origin, crop, model.voxel_size, model.n_voxels)
            y = model.neck_3d.forward_cl(vol)
            FusedConv.trace, ops.stage_trace = None, None
            h = model.bbox_head.forward_cl(y)
            boxes, scores, labels, count = model.bbox_head.get_bboxes_cl(h, y.shape[2], y.shape[1], metas, hw_transposed=True)
            if multi:
                boxes, scores, labels, count = ivx_dist.all_gather_detections(boxes, scores, labels, count)
            return ivx_dist.pack_detections(boxes, scores, labels, count).cpu()      # one packed D2H copy

        def n_det(out):
            return int(out[:, -1].sum().item())

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    # Stage events.  Inside the timed region only the launches of the `roofline` entry carry HIP events (the nine Winograd-domain GEMM launches of a step:
    # trace level 3); every other section of the line (transforms, unprojection, trunk span, tail) is measured over POST_STEPS traced steps AFTER the timed
    # region (level 1) -- an event between two launches drains the command processor's look-ahead, and ~60 of them cost a 15 ms step ~0.15 ms.
    # IVX_BENCH_TRACE_TIMED=1 restores the round-5 form (level 1 inside the timed region).
    trace_timed_all = os.environ.get('IVX_BENCH_TRACE_TIMED', '0') == '1'
    if native_trace:
        model._native.trace(0)
        model._native.trace(1 if trace_timed_all else 3)             # drop the warm-up records; the event pool they created is kept
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(args.warmup + i)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    recs_timed, n_post = None, 0
    if native_trace and not trace_timed_all:
        recs_timed = model._native.trace_records()          # stage-2 records of the timed steps
        n_post = min(args.steps, 10)
        model._native.trace(0)
        model._native.trace(1)
        for _ in range(n_post):
            model.simple_test(img, metas, gather=multi)
        torch.cuda.synchronize()
    rank_ms, rccl_ranks, all_gather_us = [round(dt / args.steps * 1e3, 3)], 1, None
    if multi:
        # self-check of the N > 1 line: every rank's own step time (all-gather) and the rank count an actual RCCL all-reduce sees
        tl = torch.tensor([dt], device=dev, dtype=torch.float64)
        allt = torch.empty((dist.get_world_size(),), device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allt, tl)
        rank_ms = [round(float(v) / args.steps * 1e3, 3) for v in allt.cpu()]
        ones = torch.ones((1,), device=dev, dtype=torch.float32)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rccl_ranks = int(ones.item())
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the step's one collective on its own (after the timed region): all-gather of the padded detection block, event-timed
        M_ = int(KITTI_TEST_CFG['max_num'])
        zb, zs = torch.zeros(B, M_, 7, device=dev), torch.zeros(B, M_, device=dev)
        zl, zc = torch.zeros(B, M_, dtype=torch.int64, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
        for _ in range(3):
            ivx_dist.all_gather_detections(zb, zs, zl, zc)
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        g0.record()
        for _ in range(20):
            ivx_dist.all_gather_detections(zb, zs, zl, zc)
        g1.record()
        torch.cuda.synchronize()
        all_gather_us = round(g0.elapsed_time(g1) / 20 * 1e3, 1)

    # the same steps with fp32 MFMA in the transformed domain (exact fp32 products): a second native handle, timed after the region above
    alt = None
    if pair and args.api == 'simple_test' and model._native is not None and not multi and os.environ.get('IVX_BENCH_ALT', '1') != '0':
        from imvoxelnet_amd import engine
        recs_keep = model._native.trace_records() if native_trace else None
        trunk_keep = FusedConv.trunk_operands
        FusedConv.wino_operands = FusedConv.trunk_operands = 0
        keep, model._native = model._native, engine.NativeModel(model, dev)
        FusedConv.wino_operands, FusedConv.trunk_operands = 4, trunk_keep
        try:
            for _ in range(min(3, max(1, args.warmup))):
                out_alt = model.simple_test(img, metas)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(args.steps):
                out_alt = model.simple_test(img, metas)
            torch.cuda.synchronize()
            ta = time.perf_counter() - ta
        finally:
            alt_native, model._native = model._native, keep
            if alt_native is not keep:
                alt_native.close()
        same = [bool(len(a['scores_3d']) == len(b['scores_3d']) and torch.equal(a['labels_3d'], b['labels_3d'])
                     and torch.allclose(a['scores_3d'], b['scores_3d'], atol=1e-4)) for a, b in zip(last, out_alt)]
        alt = {'value': round(B * args.steps / ta, 3), 'unit': 'images/s', 'ms_per_step': round(ta / args.steps * 1e3, 3),
               'note': 'bench.py --wino-operands f32 --trunk-operands f32: fp32 MFMA everywhere (exact fp32 products; the round-2 arithmetic), same weights and images',
               'detections_last_step': n_det(out_alt), 'same_detections_as_default': all(same)}

    ev_ids = list(range(args.warmup, args.warmup + args.steps))
    # The nine conv layers of KittiImVoxelNeck run in the F(6x6,3x3) minimal-filtering form (csrc/winograd.hip): input
    # transform -> ONE grouped launch of the implicit-GEMM kernel -> output transform.  The roofline entry is the implicit-GEMM kernel over its launches on the 3-D neck with the
    # FLOPs the kernel EXECUTES (64/324 of the direct count for an F(6x6,3x3) layer) over the event-bracketed duration of
    # exactly those launches; direct_equivalent_tflops divides the direct-convolution FLOPs of the whole neck by the whole
    # neck time (it may exceed the MFMA peak).
    flops_step = neck_flops_per_sample((216, 248, 12), 64, 256) * B
    # uniform stage records (kind, ms, start_ms, flops, bytes, is_3d) per traced step, from either source
    KIND = {0: 'direct', 1: 'wino_input', 2: 'wino_gemm', 3: 'wino_output', 4: 'lift', 5: 'tail', 6: 'trunk2d'}
    per_step = []
    if native_trace:
        recs = recs_keep if alt is not None else model._native.trace_records()
        n_traced = n_post if recs_timed is not None else args.steps              # one record list per step
        n_per = len(recs) // n_traced
        for k in range(n_traced):
            per_step.append([(KIND[r['stage']], r['ms'], r['start_ms'], r['flops'], r['bytes'], r['is3d']) for r in recs[k * n_per:(k + 1) * n_per]])
    else:
        for i in ev_ids:
            if not traces[i]:
                continue
            first = traces[i][0][1]
            rows = [(t[0], t[1].elapsed_time(t[2]), first.elapsed_time(t[1]), t[3], t[4], t[5]) for t in traces[i]]
            rows += [('lift', l[1].elapsed_time(l[2]), first.elapsed_time(l[1]), 0.0, 0.0, True) for l in lifts[i]]
            per_step.append(rows)
    untraced = not per_step or not per_step[0]
    if untraced:          # IVX_BENCH_TRACE=0: throughput only (no stage events in the timed region)
        per_step = [[('wino_gemm', 1e-9, 0.0, 0.0, 0.0, True), ('lift', 1e-9, 0.0, 0.0, 0.0, True), ('direct', 1e-9, 0.0, 0.0, 0.0, False)]]
    nst = len(per_step)
    neck_ms = []
    for rows in per_step:
        t3 = [r for r in rows if r[5] and r[0] != 'lift']              # the 3-D neck layers (the 2-D trunk is traced too)
        neck_ms.append(max(r[2] + r[1] for r in t3) - min(r[2] for r in t3))   # first neck launch -> the last one to finish
    tr = [r for rows in per_step for r in rows if r[5] and r[0] != 'lift']
    neck_ms_avg = sum(neck_ms) / nst
    lift_ms = sum(r[1] for rows in per_step for r in rows if r[0] == 'lift') / nst
    # unprojection: algorithmic bytes = features read once + volume written once + mask (173.1 MB/sample)
    lift_bytes = B * (1 * 64 * 96 * 320 * esz + 64 * 216 * 248 * 12 * esz + 216 * 248 * 12)
    mfma = [t for t in tr if t[0] in ('direct', 'wino_gemm')]
    mfma_ms = sum(t[1] for t in mfma) / nst
    mfma_flops = sum(t[3] for t in mfma) / nst
    n_launch = max(1, round(len(mfma) / nst))
    mfma_post_ms = None
    if recs_timed:       # the roofline entry from the events of the TIMED steps; the post-pass value beside it as a cross-check
        g_t = [r for r in recs_timed if r['stage'] == 2 and r['is3d']]
        if g_t and len(g_t) == n_launch * args.steps:
            mfma_post_ms = round(mfma_ms, 4)
            mfma_ms = sum(r['ms'] for r in g_t) / args.steps
            mfma_flops = sum(r['flops'] for r in g_t) / args.steps
    achieved = mfma_flops / (mfma_ms * 1e-3) / 1e12
    xf = [t for t in tr if t[0] in ('wino_input', 'wino_output')]
    xf_ms = sum(t[1] for t in xf) / nst
    xf_bytes = sum(t[4] for t in xf) / nst
    t2d = [r for rows in per_step for r in rows if not r[5] and r[0] != 'tail']   # ResNet-50 + FPN level 0 + the head conv
    t2d_ms = sum(t[1] for t in t2d) / nst
    t2d_flops = sum(t[3] for t in t2d if t[0] in ('direct', 'wino_gemm', 'trunk2d')) / nst

    # HBM traffic of the neck conv launches: PMC counters cannot be read from inside the process, so the value comes
    # from the committed rocprofv3 --pmc summary of this same command (tools/pmc_bench.sh -> profiles/*_bench_pmc.json;
    # FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as reported), averaged per launch; null if absent.
    traffic, traffic_src = None, None
    for name in (('r06_bench_pmc.json', 'r05_bench_pmc.json', 'r04_bench_pmc.json', 'r03b_bench_pmc.json') if pair else ('r03_bench_pmc.json', 'r02_bench_pmc.json', 'r01_bench_pmc.json')):
        pj = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(pj) and not bf16:
            try:   # the summary covers the main launch of every neck layer: HBM bytes averaged per launch
                ks = [v for v in json.load(open(pj)).values() if 'hbm_bytes' in v.get('derived', {})]
                nl = sum(v['launches'] for v in ks)
                traffic = round(sum(v['derived']['hbm_bytes'] * v['launches'] for v in ks) / nl / 1e9, 3) if nl else None
                traffic_src = 'profiles/' + name
            except Exception:
                traffic = None
            break

    trunk_pair = FusedConv.trunk_operands == 4 and not bf16
    peak2d = PEAK_BF16_MFMA_TFLOPS if (bf16 or trunk_pair) else PEAK_F32_MFMA_TFLOPS
    # The whole step against its two floors (round-5 verdict, item 5): every matrix-core product the step issues / the dense MFMA peak of the
    # operand type, and the bytes its launches must move once (inputs + outputs + filters of every launch as executed) / the HBM peak.
    # counter_GB: HBM bytes per step by the PMC counters of the committed profile of this command (all kernels of the step), or null.
    t2d_bytes = sum(t[4] for t in t2d) / nst
    step_flops = mfma_flops + t2d_flops
    neck_alg_bytes = neck_algorithmic_bytes_per_sample((216, 248, 12), 64, 256) * B       # activations + filters; V / M of the three-stage form excluded
    step_alg_bytes = neck_alg_bytes + t2d_bytes + lift_bytes
    counter_gb, counter_src = None, None
    pj_all = os.path.join(ROOT, 'profiles', 'r06_bench_pmc_all.json')
    if os.path.exists(pj_all) and pair and not bf16:
        try:
            ja = json.load(open(pj_all))
            nsteps_prof = float(ja.get('_meta', {}).get('steps', 3))
            skip = ('ubench', 'rocclr', 'at::native', '_meta')
            counter_gb = round(sum(v['derived'].get('hbm_bytes', 0.0) * v['launches'] for k, v in ja.items()
                                   if not any(q in k for q in skip) and isinstance(v, dict) and 'derived' in v) / nsteps_prof / 1e9, 2)
            counter_src = 'profiles/r06_bench_pmc_all.json'
        except Exception:
            counter_gb = None
    peak_step = PEAK_BF16_MFMA_TFLOPS if (pair or bf16) else PEAK_F32_MFMA_TFLOPS
    ms_step = dt / args.steps * 1e3
    t_mfma_floor = step_flops / (peak_step * 1e12) * 1e3
    t_hbm_floor = step_alg_bytes / 8e12 * 1e3
    whole_step = {'issued_pflop': round(step_flops / 1e15, 4), 'counter_GB': counter_gb, 'counter_source': counter_src,
                  'algorithmic_GB': round(step_alg_bytes / 1e9, 2), 'algorithmic_GB_neck': round(neck_alg_bytes / 1e9, 2), 'algorithmic_GB_trunk': round(t2d_bytes / 1e9, 2),
                  'moved_GB_neck_three_stage_form': round(sum(t[4] for t in tr) / nst / 1e9, 2), 't_mfma_floor_ms': round(t_mfma_floor, 3), 't_hbm_floor_ms': round(t_hbm_floor, 3),
                  'ms_per_step': round(ms_step, 3), 'frac': round(max(t_mfma_floor, t_hbm_floor) / ms_step, 4),
                  'note': 'floors: products issued / %.0f TFLOP/s (dense 16-bit MFMA; the pair form issues 3 per multiply-add) and algorithmic bytes / 8 TB/s; '
                          'frac = the larger floor / the measured step' % peak_step}
    if rank == 0:
        total_images = B * world * args.steps
        rec = {
            'metric': 'images/sec/node (KITTI 3x384x1280, 216x248x12 vox)',
            'value': round(total_images / dt, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('f32 (fp16 (hi, lo) pair MFMA operands: %s)' % ' and '.join(
                (['the Winograd-domain neck GEMMs'] if pair else []) + (['the chained 2-D trunk'] if (FusedConv.trunk_operands == 4 and not bf16) else [])))
            if (pair or (FusedConv.trunk_operands == 4 and not bf16)) else args.storage, 'data': 'synthetic',
            'config': {'workload': 'kitti_mono_1x3x384x1280_vox216x248x12_resnet50_fpn64_kittineck_anchor3dhead',
                       'batch_per_gpu': B, 'global_batch': B * world, 'parallelism': f'dp{world}', 'hip_graph': False,
                       'api': 'ImVoxelNet.simple_test(img, img_metas)' if args.api == 'simple_test' else 'composed stages',
                       'device_side': 'native model handle (ivx_model_detect)' if model._native is not None and args.api == 'simple_test' else 'layer-by-layer over the op-level C-ABI',
                       'stage_events': ('HIP event pairs around every launch of every timed step' if recs_timed is None else
                                        'timed steps: HIP event pairs around the %d Winograd-domain GEMM launches of a step (roofline.achieved); transforms / unprojection / '
                                        'trunk span / tail: %d traced steps after the timed region' % (n_launch, n_post)),
                       'neck_gemm_operands': ('fp16 (hi, lo) pairs of fp32 values: 3 fp16 MFMA products per pair, fp32 accumulate, device-side power-of-two scales '
                                              '(ivx_conv_desc.wino_operands = IVX_F16_PAIR)') if pair else ('bf16' if bf16 else 'fp32 MFMA'),
                       'trunk_operands': ('fp16 (hi, lo) pair ACTIVATIONS chained between the layers (ivx_conv_fwd_pio: device-side power-of-two scales '
                                          'from a bound of each output, three fp16 MFMA products per multiply-add, no conversion passes; '
                                          'ivx_model_cfg.trunk_operands = IVX_F16_PAIR)') if (FusedConv.trunk_operands == 4 and not bf16) else ('bf16' if bf16 else 'fp32 MFMA'),
                       'detections_last_step': n_det(last), 'rccl_ranks': rccl_ranks, 'ms_per_step_by_rank': rank_ms,
                       'collective': 'one all_gather_into_tensor of padded detections per step (RCCL)' if multi else None,
                       'all_gather_us': all_gather_us},
            'measured_ceilings': dict(ceil, note='csrc/ubench.hip at start-up: MFMA issue rate of the conv kernel\'s instruction, HBM copy rate '
                                                 '(read + written bytes) over 2 x 1 GiB') if ceil else None,
            'exact_fp32_mfma': alt, 'exact_fp32_mfma_images_per_s': alt['value'] if alt else None,
            'roofline': {'bound': 'mfma', 'kernel': ('conv_wino_halo_kernel / conv_wino_zblk_kernel<fp16 pair operands> (the 8 Winograd-domain GEMMs of the stride-1 / '
                                                    'z-stride-2 neck layers; z-blocked tiles on the 3-slice columns) + conv_igemm_v4_kernel<fp16 pair operands> (the last, '
                                                    'pad-0 layer): %d launches/step' % n_launch) if pair else
                                                   'conv_igemm_v4_kernel<%s> (3-D neck, %d launches/step)' % ('__bf16' if bf16 else 'float', n_launch),
                         'flops_counted': ('every fp16 MFMA product issued: 3 per fp32-equivalent multiply-add (hi*hi + hi*lo + lo*hi), priced against the dense '
                                           '16-bit MFMA peak; the z-blocked launches skip the taps outside the column and are counted at the 7/9 they issue '
                                           '(ivx_conv_winograd_issued_fraction)') if pair else 'one MFMA multiply-add per algorithmic multiply-add',
                         'fp32_equivalent_tflops': round(achieved / 3, 2) if pair else None,
                         'hbm_GBps_algorithmic': round(sum(t[4] for t in mfma) / nst / (mfma_ms * 1e-3) / 1e9, 1) if mfma_ms > 0 else None,
                         'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': round(achieved / peak, 4),
                         'peak_measured': peak_meas, 'frac_of_measured': round(achieved / peak_meas, 4) if peak_meas else None,
                         'frac_whole_neck': round(mfma_flops / (neck_ms_avg * 1e-3) / 1e12 / peak, 4),
                         'traffic': traffic, 'traffic_unit': 'GB/launch', 'traffic_source': f'committed profile {traffic_src} (rocprofv3 --pmc pass of this command; '
                                                                                             'PMC counters cannot be read in-process)' if traffic is not None else None,
                         'algorithmic_gflop_per_launch': round(mfma_flops / n_launch / 1e9, 2),
                         'avg_launch_ms': round(mfma_ms / n_launch, 4), 'launches_per_step': n_launch,
                         'mfma_launch_ms_per_step': round(mfma_ms, 3),
                         'events': ('of the timed steps' if recs_timed is not None and mfma_post_ms is not None else 'of the traced steps'),
                         'mfma_launch_ms_per_step_post_pass': mfma_post_ms,
                         'neck_ms_per_step': round(neck_ms_avg, 3), 'neck_direct_gflop_per_step': round(flops_step / 1e9, 1),
                         'neck_executed_tflops': round(mfma_flops / (neck_ms_avg * 1e-3) / 1e12, 2),
                         'direct_equivalent_tflops': round(flops_step / (neck_ms_avg * 1e-3) / 1e12, 2),
                         'winograd_gemm_launches_per_step': len([t for t in tr if t[0] == 'wino_gemm']) // nst,
                         'whole_step': whole_step},
            'roofline_winograd_transforms': None if not xf else {
                'bound': 'hbm', 'kernel': 'wino_input_kernel + wino_output_kernel (%d launches/step, event-bracketed)' % (len(xf) // nst),
                'achieved': round(xf_bytes / (xf_ms * 1e-3) / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
                'frac': round(xf_bytes / (xf_ms * 1e-3) / 8e12, 4), 'peak_measured': hbm_meas,
                'frac_of_measured': round(xf_bytes / (xf_ms * 1e-3) / 1e9 / hbm_meas, 4) if hbm_meas else None, 'ms_per_step': round(xf_ms, 3),
                'algorithmic_GB_per_step': round(xf_bytes / 1e9, 2)},
            'roofline_unprojection': {'bound': 'hbm', 'kernel': 'backproject_single_view_kernel (1 launch/step, event-bracketed)',
                                      'achieved': round(lift_bytes / (lift_ms * 1e-3) / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
                                      'frac': round(lift_bytes / (lift_ms * 1e-3) / 8e12, 4), 'peak_measured': hbm_meas,
                                      'frac_of_measured': round(lift_bytes / (lift_ms * 1e-3) / 1e9 / hbm_meas, 4) if hbm_meas else None,
                                      'ms': round(lift_ms, 4),
                                      'algorithmic_MB': round(lift_bytes / 1e6, 1)},
            'roofline_trunk_2d': {'bound': 'mfma', 'kernel': 'stem_pool_pair_kernel + bottleneck_pio_kernel (first block of stage 1 with its shortcut conv, identity blocks of stages 1-2) + conv_igemm_v4_kernel (ResNet-50 + FPN level 0 + head conv; %s)' % (
                                      'one event pair around the whole trunk + one around the head conv' if native_trace else
                                      '%d launches/step, event-bracketed incl. their transform / split-K passes' % (len(t2d) // nst)),
                                  'achieved': round(t2d_flops / (t2d_ms * 1e-3) / 1e12, 2) if t2d_ms > 0 else None, 'peak': peak2d, 'unit': 'TFLOP/s',
                                  'frac': round(t2d_flops / (t2d_ms * 1e-3) / 1e12 / peak2d, 4) if t2d_ms > 0 else None,
                                  'flops_counted': ('every fp16 MFMA product issued by the chained pair form: 3 per fp32 multiply-add, '
                                                    'priced against the dense 16-bit MFMA peak; these layers are bound by HBM / launch latency, not by the matrix pipe') if trunk_pair else
                                                   ('matrix-core products of the executed form: one per multiply-add on the fp32 MFMA layers, three on the 2-D 3x3 layers that run '
                                                    'as Winograd on fp16-pair operands (priced against the fp32 MFMA peak all the same: a mixed span)'),
                                  'fp32_equivalent_tflops': round(t2d_flops / 3 / (t2d_ms * 1e-3) / 1e12, 2) if (trunk_pair and t2d_ms > 0) else None,
                                  'ms_per_step': round(t2d_ms, 3), 'executed_gflop_per_step': round(t2d_flops / 1e9, 1),
                                  'algorithmic_GB_per_step': round(t2d_bytes / 1e9, 3) if t2d_bytes > 0 else None,
                                  'hbm_GBps_algorithmic': round(t2d_bytes / (t2d_ms * 1e-3) / 1e9, 1) if (t2d_bytes > 0 and t2d_ms > 0) else None},
        }
        if untraced:
            for k in ('roofline', 'roofline_winograd_transforms', 'roofline_unprojection', 'roofline_trunk_2d'):
                rec[k] = None
            rec['note'] = 'IVX_BENCH_TRACE=0: stage events disabled, throughput only'
        if bf16:
            rec['note'] = 'reduced-precision storage mode (bf16 activations/weights, fp32 accumulate); NOT the headline metric, which is quoted at fp32'
        if world == 1 and not multi and args.api == 'simple_test' and not bf16 and os.environ.get('IVX_BENCH_EXTRA', '1') != '0':
            # The other BASELINE.json workloads on the driver's line (round-4 verdict, item 4): a few timed steps each of the public call, same
            # timing method (bench_other), after the KITTI region -- parity-test configurations, NOT the headline metric.
            import subprocess
            rec['extra_configs'] = []
            for cname, views in (('nuscenes', 0), ('scannet_fast', 20), ('scannet_v1', 50), ('sunrgbd_fast', 0)):
                cmd = [sys.executable, os.path.abspath(__file__), '--config', cname, '--steps', '5', '--warmup', '2', '--batch', str(BATCH_PER_GPU)]
                if views:
                    cmd += ['--views', str(views)]
                if not pair:
                    cmd += ['--wino-operands', 'f32']
                if FusedConv.trunk_operands != 4:
                    cmd += ['--trunk-operands', 'f32']
                try:      # its own process, bounded: a fault or a hang in an extra configuration cannot cost the headline line (round-5 advisor)
                    pr = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
                    line = [l for l in pr.stdout.splitlines() if l.startswith('{"metric')]
                    if pr.returncode != 0 or not line:
                        raise RuntimeError('exit code %d: %s' % (pr.returncode, (pr.stderr or '')[-200:]))
                    r2 = json.loads(line[-1])
                    rf = r2['roofline']
                    rec['extra_configs'].append({
                        'workload': r2['metric'], 'value': r2['value'], 'unit': r2['unit'], 'ms_per_step': r2['ms_per_step'],
                        'steps': r2['steps'], 'warmup': r2['warmup'], 'batch_per_gpu': r2['config']['batch_per_gpu'], 'views': r2['config']['views'],
                        'api': r2['config']['api'], 'detections_last_step': r2['config']['detections_last_step'],
                        'neck_frac_of_mfma_peak': rf['frac'], 'neck_ms_per_step': rf['neck_ms_per_step'],
                        'trunk_2d_ms_per_step': (r2['roofline_trunk_2d'] or {}).get('ms_per_step')})
                except Exception as e:
                    rec['extra_configs'].append({'workload': cname, 'error': repr(e)[:300]})
            rec['extra_configs_note'] = ('five timed steps each of the public call in a subprocess of this script (same dtype / operand modes as the headline); '
                                         'parity-test configurations, not the headline metric')
        if world == 1 and not args.no_cpu_baseline:
            from oracle import imvoxel_oracle as orc
            sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            cfg = dict(n_voxels=(216, 248, 12), voxel_size=(.32, .32, .32), neck='kitti', num_classes=1, test_cfg=KITTI_TEST_CFG,
                       anchor=dict(ranges=[[0, -39.68, -1.78, 69.12 - .32, 39.68 - .32, -1.78]], sizes=[[1.6, 3.9, 1.56]],
                                   rotations=[0, 1.57]))
            ts = []
            for k in range(1 + args.cpu_repeats):            # one untimed warm-up image, then `cpu_repeats` timed ones
                tc = time.perf_counter()
                orc.simple_test_anchor(img_host[k % B:k % B + 1], metas[k % B:k % B + 1], sd, cfg)
                ts.append(time.perf_counter() - tc)
            ts = ts[1:]
            tc = sum(ts) / len(ts)
            rec['cpu_baseline'] = {'value': round(1.0 / tc, 4), 'unit': 'images/s', 'cores': torch.get_num_threads(),
                                   'kind': 'port', 'host_cpus': os.cpu_count(), 'physical_cores': physical_cores(),
                                   'sample': '%d images (1x3x384x1280 -> 216x248x12 each, one at a time) through the oracle port (torch-CPU fp32 '
                                             'convs + C unprojection/NMS) after one untimed warm-up image: %s s, mean %.2f s'
                                             % (len(ts), ' / '.join('%.2f' % t for t in ts), tc)}
            if not args.no_cpu_cabi:
                # SURVEY 8d baseline (ii): the build's own CPU restatement behind the SAME C-ABI (oracle/cpu_abi: csrc/model.cpp over
                # plain-loop OpenMP ops), same weights, same image, through ivx_model_forward on host memory
                import importlib.util
                os.environ.setdefault('OMP_NUM_THREADS', str(torch.get_num_threads()))
                spec = importlib.util.spec_from_file_location('ivx_cpu_abi_host', os.path.join(ROOT, 'oracle', 'cpu_abi', 'host.py'))
                host = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(host)
                cm = host.CpuModel(model)
                try:
                    tcs = []
                    for k in range(1 + max(1, args.cpu_repeats - 1)):       # one untimed warm-up image, then cpu_repeats - 1 timed ones
                        tc0 = time.perf_counter()
                        cdet = cm.forward(img_host[k % B:k % B + 1], metas[k % B:k % B + 1])
                        tcs.append(time.perf_counter() - tc0)
                finally:
                    cm.close()
                tcs = tcs[1:]
                rec['cpu_baseline_cabi'] = {'value': round(len(tcs) / sum(tcs), 4), 'unit': 'images/s', 'cores': int(os.environ['OMP_NUM_THREADS']),
                                            'kind': 'port', 'host_cpus': os.cpu_count(), 'physical_cores': physical_cores(),
                                            'detections_last_image': int(len(cdet[0][1])),
                                            'sample': '%d images through ivx_model_forward of oracle/_cpuabi/libimvoxel_cpu.so (the model-level C-ABI '
                                                      'over the CPU restatement: direct convolutions, no Winograd form) after one untimed warm-up '
                                                      'image: %s s' % (len(tcs), ' / '.join('%.2f' % t for t in tcs))}
        print(json.dumps(rec))
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
