"""Multi-GPU inference: scenes are independent, so a batch is partitioned across ranks (one process per
GPU, torch.distributed: backend 'nccl' == RCCL over xGMI on the MI355X node, 'gloo' in CPU tests) and the
only exchange is one all-gather of fixed-size padded detections per batch.  It replaces mmdet's pickle
based collect_results_cpu/gpu used by the reference's multi_gpu_test (tools/test.py:131-136).

Payload per sample: max_num x (7 box + score + label) fp32 + count  (KITTI: 50 x 9 x 4 B = 1.8 kB), so the
collective is latency-bound; there is deliberately no other data-path collective.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous partition of n_items over `world` ranks (sizes differ by at most 1)."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shard_batch(img, img_metas, rank=None, world=None):
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
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


def all_gather_detections(boxes, scores, labels, count, group=None):
    """Every rank contributes its [B_local, ...] detections (same B_local and max_num on every rank) and
    receives the detections of the whole batch in rank order."""
    packed = pack_detections(boxes, scores, labels, count)
    if not (dist.is_available() and dist.is_initialized()):
        return unpack_detections(packed, scores.shape[1])
    world = dist.get_world_size(group)
    out = torch.empty((world * packed.shape[0], packed.shape[1]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    return unpack_detections(out, scores.shape[1])
