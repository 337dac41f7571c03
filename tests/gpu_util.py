import numpy as np
import torch


def stats(name, got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    d = np.abs(got - ref)
    return (f'{name}: shape {got.shape} max|d| {d.max() if d.size else 0:.3e} mean|d| {d.mean() if d.size else 0:.3e} '
            f'max|ref| {np.abs(ref).max() if ref.size else 0:.3e} nan {int(np.isnan(got).sum())}')


def assert_close(name, got, ref, rtol, atol):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    assert got.shape == ref.shape, f'{name}: shape {got.shape} vs {ref.shape}'
    ok = np.allclose(got, ref, rtol=rtol, atol=atol)
    msg = stats(name, got, ref)
    print(msg)
    if not ok:
        bad = ~np.isclose(got, ref, rtol=rtol, atol=atol)
        idx = np.argwhere(bad)[:5]
        raise AssertionError(msg + f' | {int(bad.sum())} elements out of tolerance, first at {idx.tolist()}: '
                             f'got {got[bad][:5]} ref {ref[bad][:5]}')


def cl(x):
    """[B,C,*sp] cpu/np -> channels-last [B,D,H,W,C] device tensor (layout change done on the host)."""
    t = torch.as_tensor(np.asarray(x), dtype=torch.float32)
    if t.dim() == 4:
        t = t.unsqueeze(2)
    return t.permute(0, 2, 3, 4, 1).contiguous().cuda()


def uncl(y):
    """channels-last device [B,D,H,W,C] -> numpy [B,C,D,H,W]."""
    return y.permute(0, 4, 1, 2, 3).contiguous().cpu().numpy()


class direct_conv_only:
    """with direct_conv_only(): FusedConv runs every layer on the direct implicit-GEMM kernel (tests of that kernel's
    tiles, split-K and tail plans; the Winograd form has its own tests)."""

    def __enter__(self):
        from imvoxelnet_amd.conv import FusedConv
        self._old, FusedConv.winograd = FusedConv.winograd, False
        return self

    def __exit__(self, *exc):
        from imvoxelnet_amd.conv import FusedConv
        FusedConv.winograd = self._old
        return False
