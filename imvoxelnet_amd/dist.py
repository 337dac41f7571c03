"""Multi-GPU inference: scenes are independent, so a batch is partitioned across ranks (one process per
GPU, torch.distributed: backend 'nccl' == RCCL over xGMI on the MI355X node, 'gloo' in CPU tests) and the
only exchange is one all-gather of fixed-size padded detections per batch.  It replaces mmdet's pickle
based collect_results_cpu/gpu used by the reference's multi_gpu_test (tools/test.py:131-136).

Payload per sample: max_num x (7 box + score + label) fp32 + count  (KITTI: 50 x 9 x 4 B = 1.8 kB), so the
collective is latency-bound; the sample-sharded mode deliberately has no other data-path collective.

Second, optional mode for single-scene latency with many views: VIEW sharding.  Each rank runs the 2-D
trunk and the partial unprojection on its slice of the views, the partial view sums and view counts are all-reduced
(the one real exchange step of the path: 26-300 MB fp32 per scene, ring all-reduce over xGMI), every rank normalises
and continues with the (replicated) 3-D neck and head.  See view_sharded_lift / ImVoxelNet.simple_test_view_sharded.

SURVEY 8e's preferred form of that exchange for the stack necks (KittiImVoxelNeck / NuScenesImVoxelNeck): a REDUCE-SCATTER OVER X-SLABS
WITH A HALO (exchange_volume_slabs): rank r receives only the totals of its own slab of the volume, widened by the receptive field of
the neck along x, normalises and convolves that slab alone, and the ranks all-gather the cropped neck outputs (one x-slab of
[B,X',Y',1,C] each: 8x smaller than the volume).  Half the bytes of the all-reduce on the wire, and the 3-D neck -- the other half of a
multi-view step -- is divided by the number of ranks as well (minus the halo's redundant rows).  See StackNeckSlabs.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous partition of n_items over `world` ranks (sizes differ by at most 1)."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def _rank_world(rank, world, group=None):
    """Defaults from the process group (`group`: a sub-group, whose ranks are numbered 0 .. size-1; None: the default group);
    (0, 1) when torch.distributed is not initialised (single process)."""
    on = dist.is_available() and dist.is_initialized()
    if rank is None:
        rank = dist.get_rank(group) if on else 0
        if rank < 0:       # torch returns -1 for a process outside `group`: indexing shard plans with it would silently pick the last one
            raise RuntimeError('this process is not a member of the process group passed as `group`; call the sharded entry points from member ranks only')
    return (rank, (dist.get_world_size(group) if on else 1) if world is None else world)


def _peer(group, r):
    """Global rank of group rank r (point-to-point ops address peers by GLOBAL rank, also inside a sub-group)."""
    return r if group is None else dist.get_global_rank(group, r)


def is_collecting_rank(group=None):
    """True on the rank that builds the collected result list (rank 0 of the group, or any single process)."""
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank(group) == 0


def shard_batch(img, img_metas, rank=None, world=None):
    rank, world = _rank_world(rank, world)
    a, b = shard_range(len(img_metas), rank, world)
    return img[a:b], img_metas[a:b]


def pack_detections(boxes, scores, labels, count):
    """[B,M,7],[B,M],[B,M] int64,[B] int32 -> one fp32 tensor [B, M*9 + 1] (labels/count are small ints:
    exact in fp32)."""
    B, M = scores.shape
    body = torch.cat([boxes, scores.unsqueeze(-1), labels.to(torch.float32).unsqueeze(-1)], dim=-1).reshape(B, M * 9)
    return torch.cat([body, count.to(torch.float32).unsqueeze(-1)], dim=-1).contiguous()


def unpack_detections(packed, max_num):
    B = packed.shape[0]
    body = packed[:, :max_num * 9].reshape(B, max_num, 9)
    return body[..., :7], body[..., 7], body[..., 8].to(torch.int64), packed[:, -1].to(torch.int32)


def all_gather_detections(boxes, scores, labels, count, group=None, global_batch=None):
    """Every rank contributes its [B_local, ...] detections (same max_num everywhere) and receives the detections of the
    whole batch in rank order.  global_batch: size of the batch shard_batch() partitioned -- when it does not divide by
    the world size the shards differ by one sample (shard_range), so every rank pads its slice to ceil(global / world)
    rows with count 0 before the fixed-size all-gather and the padding rows are dropped afterwards (mmdet's
    collect_results handles ragged shards the same way: pad to the longest, trim to len(dataset)).  Without
    global_batch all ranks must hold the same B_local (checked)."""
    packed = pack_detections(boxes, scores, labels, count)
    M = scores.shape[1]
    if not (dist.is_available() and dist.is_initialized()):
        return unpack_detections(packed, M)
    world = dist.get_world_size(group)
    if global_batch is None:
        sizes = torch.tensor([packed.shape[0], -packed.shape[0]], device=packed.device, dtype=torch.int64)
        dist.all_reduce(sizes, op=dist.ReduceOp.MAX, group=group)          # max(B_local), -min(B_local)
        if int(sizes[0]) != -int(sizes[1]):
            raise ValueError(f'ranks hold between {-int(sizes[1])} and {int(sizes[0])} samples: pass global_batch= for ragged shards')
        rows = packed.shape[0]
    else:
        rows = (int(global_batch) + world - 1) // world
        if packed.shape[0] > rows:
            raise ValueError(f'this rank holds {packed.shape[0]} samples, more than ceil({global_batch} / {world})')
        if packed.shape[0] < rows:                                           # padding rows: all zeros, i.e. count 0
            packed = torch.cat([packed, packed.new_zeros((rows - packed.shape[0], packed.shape[1]))])
    out = torch.empty((world * rows, packed.shape[1]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, packed.contiguous(), group=group)
    if global_batch is not None:
        keep = [r * rows + i for r in range(world) for i in range(shard_range(int(global_batch), r, world)[1] - shard_range(int(global_batch), r, world)[0])]
        out = out[torch.tensor(keep, device=out.device, dtype=torch.int64)]
    return unpack_detections(out, M)


def shard_views(img, img_metas, rank=None, world=None):
    """img [B,V,3,H,W], metas with V extrinsics each -> this rank's contiguous slice of the views (same slice for
    every sample).  Returns (img_local [B,V_local,...], metas_local, (v0, v1))."""
    rank, world = _rank_world(rank, world)
    v0, v1 = shard_range(img.shape[1], rank, world)
    metas = []
    for m in img_metas:
        m2 = dict(m)
        l2i = dict(m['lidar2img'])
        l2i['extrinsic'] = list(m['lidar2img']['extrinsic'])[v0:v1]
        m2['lidar2img'] = l2i
        metas.append(m2)
    return img[:, v0:v1].contiguous(), metas, (v0, v1)


def all_reduce_volume(vol_sum, count, group=None):
    """Sum the per-rank partial view sums [B,X,Y,Z,C] fp32 and view counts [B,X,Y,Z] int32 over the ranks, in place.
    Two collectives per batch; a no-op without an initialised process group."""
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(vol_sum, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(count, op=dist.ReduceOp.SUM, group=group)
    return vol_sum, count


def view_sharded_lift(model, img, img_metas, rank=None, world=None, group=None):
    """The exchange step of the view-sharded mode.  img [B,V,3,H,W] and img_metas are the FULL inputs (present on every
    rank); returns the complete (volume [B,X,Y,Z,C], valid [B,X,Y,Z] bool) on every rank."""
    from . import ops
    rank, world = _rank_world(rank, world, group)
    img_l, metas_l, (v0, v1) = shard_views(img, img_metas, rank, world)
    if v1 > v0:
        p0 = model.features_2d_cl(img_l)
        proj, origin, crop = model._camera_setup(metas_l, 4, p0.device)
        vol, cnt = ops.backproject_sum(p0, proj, origin, crop, model.voxel_size, model.n_voxels)
    else:     # more ranks than views: this rank contributes zeros
        B = img.shape[0]
        cf = model.neck.out_channels
        vol = torch.zeros((B,) + tuple(model.n_voxels) + (cf,), device=img.device, dtype=torch.float32)
        cnt = torch.zeros((B,) + tuple(model.n_voxels), device=img.device, dtype=torch.int32)
    all_reduce_volume(vol, cnt, group)
    return ops.volume_normalize_(vol, cnt)


# ------------------------------------------------------------------ reduce-scatter over X-slabs (+ halo) for the stack necks
class StackNeckSlabs:
    """Index arithmetic of running a stack neck (block, conv, block, conv, block, conv: necks3d._StackNeck) on one x-slab.
    Along x every layer is a k = 3 convolution with its own stride / padding; output index o of a layer reads the inputs
    o*s - p .. o*s - p + 2.  For the output slab [oa, ob) of rank r the constructor walks the layers backwards to the input
    rows [lo, hi] those outputs depend on, widens the slab to [ea, eb) = [floor(lo) to a multiple of the total x stride, hi + 1)
    (so that local and global indices of every strided layer stay congruent), clipped to the volume.  Outputs of the local run
    that depend on the artificial zero padding at ea / eb lie outside [oa, ob) by construction and are cropped away."""

    def __init__(self, neck, X, world, rank):
        from .necks3d import BasicBlock3d
        layers, k = [], 0
        for m in neck.model:
            if isinstance(m, BasicBlock3d):
                layers += [(1, 1), (1, 1)]
            else:
                layers.append((int(neck.strides[k][0]), int(neck.paddings[k][0])))
                k += 1
        self.layers = layers
        n = X
        self.S = 1
        for s_, p_ in layers:
            n = (n + 2 * p_ - 3) // s_ + 1
            self.S *= s_
        self.X, self.Xo = X, n
        if X % self.S:
            raise ValueError(f'the x extent {X} of the volume is not a multiple of the neck\'s x stride {self.S}')
        self.oa, self.ob = shard_range(self.Xo, rank, world)
        if self.ob <= self.oa:
            raise ValueError(f'more ranks ({world}) than output rows ({self.Xo}): use the all-reduce exchange')
        lo, hi = self.oa, self.ob - 1
        for s_, p_ in reversed(layers):
            lo, hi = lo * s_ - p_, hi * s_ - p_ + 2
        self.ea = max(0, lo // self.S * self.S)
        self.eb = min(X, hi + 1)
        self.off = self.ea // self.S                     # global output index of local output row 0

    def crop(self, y_local):
        """[B, x_local, Y', 1, C] of the local run -> this rank's output rows [oa, ob)."""
        return y_local[:, self.oa - self.off:self.ob - self.off].contiguous()


def exchange_volume_slabs(vol_sum, count, plans, group=None, rank=None):
    """The exchange step in its reduce-scatter form.  vol_sum [B,X,Y,Z,C] fp32 / count [B,X,Y,Z] int32: this rank's partial
    view sum and view count over the WHOLE volume; plans[r] = StackNeckSlabs of rank r (every rank builds all of them: pure
    arithmetic).  Every rank sends rank r its partial rows [ea_r, eb_r) and adds up what it receives in rank order (a fixed order:
    deterministic) -- grouped point-to-point transfers, i.e. an all-to-all over xGMI, which both RCCL and gloo provide.
    Returns (sum_ext [B, eb-ea, Y, Z, C], count_ext [B, eb-ea, Y, Z]) of this rank's widened slab, totals over all ranks."""
    on = dist.is_available() and dist.is_initialized()
    rank, world = _rank_world(rank, len(plans) if not on else None, group)
    if world != len(plans):
        raise ValueError(f'{len(plans)} slab plans for a group of {world} ranks')
    me = plans[rank]
    mine_v = vol_sum[:, me.ea:me.eb].contiguous()
    mine_c = count[:, me.ea:me.eb].contiguous()
    if not on or world == 1:
        return mine_v, mine_c
    send, recv_v, recv_c, ops_ = [], {}, {}, []
    for r in range(world):
        if r == rank:
            continue
        pr = plans[r]
        sv, sc = vol_sum[:, pr.ea:pr.eb].contiguous(), count[:, pr.ea:pr.eb].contiguous()
        send += [sv, sc]                               # keep alive until the transfers complete
        recv_v[r], recv_c[r] = torch.empty_like(mine_v), torch.empty_like(mine_c)
        pr_g = _peer(group, r)                         # plans are indexed by GROUP rank; the transfer names the peer's global rank
        ops_ += [dist.P2POp(dist.isend, sv, pr_g, group), dist.P2POp(dist.isend, sc, pr_g, group),
                 dist.P2POp(dist.irecv, recv_v[r], pr_g, group), dist.P2POp(dist.irecv, recv_c[r], pr_g, group)]
    for q in dist.batch_isend_irecv(ops_):
        q.wait()
    tot_v, tot_c = None, None
    for r in range(world):                             # rank order, whoever we are
        v, c = (mine_v, mine_c) if r == rank else (recv_v[r], recv_c[r])
        tot_v, tot_c = (v.clone(), c.clone()) if tot_v is None else (tot_v.add_(v), tot_c.add_(c))
    return tot_v, tot_c


def all_gather_rows(y, plans, group=None, rank=None):
    """Cropped neck outputs [B, ob-oa, Y', 1, C] of every rank -> the whole map [B, X', Y', 1, C] on every rank (slabs differ by at
    most one row: padded to the longest for the fixed-size all-gather)."""
    on = dist.is_available() and dist.is_initialized()
    if not on or len(plans) == 1:
        return y
    rows = max(p.ob - p.oa for p in plans)
    B = y.shape[0]
    pad = y if y.shape[1] == rows else torch.cat([y, y.new_zeros((B, rows - y.shape[1]) + tuple(y.shape[2:]))], 1)
    out = torch.empty((len(plans) * B,) + tuple(pad.shape[1:]), dtype=y.dtype, device=y.device)      # concatenation along dim 0 (both backends)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    out = out.view((len(plans), B) + tuple(pad.shape[1:]))
    return torch.cat([out[r][:, :plans[r].ob - plans[r].oa] for r in range(len(plans))], 1).contiguous()


def view_sharded_neck_slabs(model, img, img_metas, group=None, rank=None, world=None):
    """View-sharded step up to the neck output with the reduce-scatter exchange (stack necks only): -> [B, X', Y', 1, C] on every rank."""
    from . import ops
    rank, world = _rank_world(rank, world, group)
    X = int(model.n_voxels[0])
    plans = [StackNeckSlabs(model.neck_3d, X, world, r) for r in range(world)]
    img_l, metas_l, (v0, v1) = shard_views(img, img_metas, rank, world)
    if v1 > v0:
        p0 = model.features_2d_cl(img_l)
        proj, origin, crop = model._camera_setup(metas_l, 4, p0.device)
        vol, cnt = ops.backproject_sum(p0, proj, origin, crop, model.voxel_size, model.n_voxels)
    else:
        B = img.shape[0]
        vol = torch.zeros((B,) + tuple(model.n_voxels) + (model.neck.out_channels,), device=img.device, dtype=torch.float32)
        cnt = torch.zeros((B,) + tuple(model.n_voxels), device=img.device, dtype=torch.int32)
    sv, sc = exchange_volume_slabs(vol, cnt, plans, group, rank)
    slab, _ = ops.volume_normalize_(sv, sc)
    y = plans[rank].crop(model.neck_3d.forward_cl(slab))
    return all_gather_rows(y, plans, group, rank)
