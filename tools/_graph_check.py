import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, imvoxelnet_amd as ia, kitti_cfg as kc
from imvoxelnet_amd.engine import NativeModel
m = ia.build_detector(kc.kitti_model_cfg(), test_cfg=kc.KITTI_TEST_CFG); ia.randomize_(m, 0)
with torch.no_grad():
    m.bbox_head.conv_cls.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(5)); m.bbox_head.conv_cls.bias.fill_(-2.0)
m.prepare(torch.device('cuda'), native=False)
B = 4
img = torch.randn(B, 1, 3, 384, 1280, generator=torch.Generator().manual_seed(1)).cuda()
metas = [kc.kitti_meta(t=(0.01 * b, 0, 0), box_type=ia.LiDARInstance3DBoxes) for b in range(B)]
proj, no, crop = m._camera_setup(metas, 4, img.device)
x = img.reshape(B, 3, 384, 1280).contiguous()
for graph in (False, True):
    nat = NativeModel(m, torch.device('cuda'), graph=graph)
    for trace in (False, True):
        nat.trace(trace)
        outs = []
        for i in range(4):
            o = nat.forward(x, B, 1, 384, 1280, proj, no, crop)
            outs.append([t.clone() for t in o])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            o = nat.forward(x, B, 1, 384, 1280, proj, no, crop)
            c = o[3].cpu()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10 * 1e3
        same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[3]))
        line = f'graph={graph} trace={trace}: {dt:.3f} ms/step, replay == eager first call: {same}, detections {int(o[3].sum())}'
        if trace:
            recs = nat.trace_records()
            g = [r for r in recs if r['stage'] == 2 and r['is3d']]
            line += f' | {len(recs)} records, neck gemm sum {sum(r["ms"] for r in g):.3f} ms, all stages sum {sum(r["ms"] for r in recs):.3f} ms'
        print(line, flush=True)
    ref = outs
