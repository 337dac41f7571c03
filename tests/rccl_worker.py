"""Worker of tests/test_gpu_engine.py::test_rccl_two_gpus_* (one process per GPU under torch.distributed.run, backend 'nccl' = RCCL).
Rank r takes GPU r.  Part 1: the sample-sharded step -- every rank runs simple_test(gather=True) on its slice of a global batch and
rank 0 compares the collected list with simple_test over the whole batch on its own GPU.  Part 2: the view-sharded step with the
reduce-scatter exchange over x-slabs against the single-GPU call.  Every check is an assertion (a first multi-GPU run fails loudly), and
rank 0 prints one JSON line for the calling test."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def same(a, b, atol):
    return bool(len(a['scores_3d']) == len(b['scores_3d']) and torch.equal(a['labels_3d'], b['labels_3d']) and
                torch.allclose(a['scores_3d'], b['scores_3d'], atol=atol) and torch.allclose(a['boxes_3d'].tensor, b['boxes_3d'].tensor, atol=atol * 10))


def main():
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    try:
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    except TypeError:
        dist.init_process_group('nccl', rank=rank, world_size=world)
    import imvoxelnet_amd as ia
    from imvoxelnet_amd import dist as ivd, workloads as kc
    out = {'world': world}
    # ---- part 1: samples sharded, one all-gather of the padded detections
    model = ia.build_detector(kc.kitti_model_cfg(n_voxels=(104, 120, 12)), test_cfg=dict(kc.KITTI_TEST_CFG))
    ia.randomize_(model, 11)
    with torch.no_grad():
        model.bbox_head.conv_cls.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(1))
        model.bbox_head.conv_cls.bias.fill_(-1.0)
        model.bbox_head.conv_reg.weight.normal_(0, 0.002, generator=torch.Generator().manual_seed(2))
    model.prepare(dev)
    hw = (192, 640)
    ones = torch.ones((1,), device=dev)
    dist.all_reduce(ones)
    out['rccl_ranks'] = int(ones.item())
    assert out['rccl_ranks'] == world, f"the RCCL group has {out['rccl_ranks']} ranks, WORLD_SIZE is {world}"      # fail loudly, do not just report
    # equal shards (2 per rank) and ragged ones (one rank holds a sample more): the collected list must equal the single-process result
    for GB, ragged in ((2 * world, False), (2 * world + 1, True)):
        img = torch.randn(GB, 1, 3, *hw, generator=torch.Generator().manual_seed(3 + GB)).to(dev)
        metas = [kc.kitti_meta(img_hw=hw, t=(0.01 * b, 0, 0), box_type=ia.LiDARInstance3DBoxes) for b in range(GB)]
        img_l, metas_l = ivd.shard_batch(img, metas, rank, world)
        assert len(metas_l) == ivd.shard_range(GB, rank, world)[1] - ivd.shard_range(GB, rank, world)[0]
        got = model.simple_test(img_l.contiguous(), metas_l, gather=True, global_batch=GB if ragged else None)
        if rank == 0:
            want = model.simple_test(img, metas)
            ok = len(got) == GB and all(same(a, b, 1e-5) for a, b in zip(got, want))
            assert ok, f'global batch {GB} over {world} ranks: the gathered detections differ from the single-process result'
            if not ragged:
                out['gather_len'] = len(got)
                out['gather_same'] = ok
                out['gather_detections'] = int(sum(len(r['scores_3d']) for r in got))
            else:
                out['ragged_gather_len'] = len(got)
                out['ragged_gather_same'] = ok
        else:
            assert got is None
    # ---- part 2: views sharded, reduce-scatter of x-slabs + all-gather of the neck rows (stack neck, anchor head: the nuScenes family)
    m2 = ia.build_detector(kc.nuscenes_model_cfg(n_voxels=(48, 48, 12), dcn=False), test_cfg=dict(kc.NUSCENES_TEST_CFG))
    ia.randomize_(m2, 12)
    with torch.no_grad():
        m2.bbox_head.conv_cls.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(4))
        m2.bbox_head.conv_cls.bias.fill_(-1.0)
    m2.prepare(dev)
    V, hw2 = 6, (160, 256)
    img2 = torch.randn(1, V, 3, *hw2, generator=torch.Generator().manual_seed(5)).to(dev)
    meta2 = [kc.nuscenes_meta(img_hw=hw2, box_type=ia.LiDARInstance3DBoxes)]
    r_slab = m2.simple_test_view_sharded(img2, meta2, exchange='reduce_scatter')
    r_allr = m2.simple_test_view_sharded(img2, meta2, exchange='all_reduce')
    want2 = m2.simple_test(img2, meta2)
    # the order of the view sum differs between the forms (fp32 rounding): same kept boxes, values to 1e-3
    out_ok = torch.tensor([1.0 if (same(r_slab[0], want2[0], 1e-3) and same(r_allr[0], want2[0], 1e-3)) else 0.0], device=dev)
    dist.all_reduce(out_ok, op=dist.ReduceOp.MIN)
    out['view_sharded_same'] = bool(out_ok.item() == 1.0)
    assert out['view_sharded_same'], 'the view-sharded step (reduce-scatter / all-reduce exchange) differs from the single-GPU call'
    out['view_detections'] = int(len(want2[0]['scores_3d']))
    if rank == 0:
        print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
