import sys, torch
sys.path.insert(0, '/root/repo')
from imvoxelnet_amd import ops
def t(f, it=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
g = torch.Generator(device='cuda').manual_seed(0)
for (Z, C) in ((3, 256), (6, 128), (12, 64)):
    for nb in (64, 4):
        x = torch.randn(nb, 108, 124, Z, C, device='cuda', generator=g)
        w = torch.randn(C, 1, 1, 3, C, device='cuda', generator=g) * 0.02
        y = ops.conv_fwd(x, w, None, None, (1, 1, 3), (1, 1, 1), (0, 0, 1))
        ms = t(lambda: ops.conv_fwd(x, w, None, None, (1, 1, 3), (1, 1, 1), (0, 0, 1), out=y))
        fl = 2.0 * y.numel() * C * 3
        print(f'xi-GEMM Z{Z} C{C} batch {nb}: {ms:8.3f} ms  {fl / ms / 1e9:6.1f} TF   (x16/{64 // nb if nb == 4 else 1}: {ms * (16 if nb == 4 else 1):.2f} ms total for B=4)', flush=True)
        del y
    xs = torch.randn(4, 216, 248, Z, C, device='cuda', generator=g)
    ms = t(lambda: torch.mul(x, 2.0, out=x))
    print(f'   in-place scale of V ({x.numel() * 4 / 1e9:.2f} GB r+w): {ms:.3f} ms  {2 * x.numel() * 4 / ms / 1e9:.0f} GB/s', flush=True)
    w3 = torch.randn(C, 3, 3, 3, C, device='cuda', generator=g) * 0.02
    y3 = ops.conv_fwd(xs, w3, None, None, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    ms = t(lambda: ops.conv_fwd(xs, w3, None, None, (3, 3, 3), (1, 1, 1), (1, 1, 1), out=y3))
    print(f'   direct 3x3x3: {ms:.3f} ms {2.0 * y3.numel() * C * 27 / ms / 1e9:.1f} TF', flush=True)
    del x, xs, y3
