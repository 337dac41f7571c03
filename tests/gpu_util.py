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
        self._old_pair, FusedConv.pair_mode = FusedConv.pair_mode, 0      # ... and not the split-operand form of the 3x3x3 layers (conv.py pair_mode)
        return self

    def __exit__(self, *exc):
        from imvoxelnet_amd.conv import FusedConv
        FusedConv.winograd = self._old
        FusedConv.pair_mode = self._old_pair
        return False


def match_rows(kept, cand):
    """Index of every row of `kept` inside `cand` by exact (bitwise) equality; both [*, d] tensors on one device."""
    kept, cand = torch.as_tensor(kept), torch.as_tensor(cand)
    if len(kept) == 0:
        return torch.zeros((0,), dtype=torch.int64)
    eq = (kept[:, None, :] == cand[None, :, :]).all(-1)
    assert bool(eq.any(1).all()), f'{int((~eq.any(1)).sum())} kept rows are not among the candidates'
    return eq.float().argmax(1).cpu()


def assert_same_kept(name, got_ids, got_scores, ref_ids, ref_scores, tie_rel=1e-5, boundary=False):
    """north_star parity clause "identical box indices after NMS": the sequences of kept candidate ids (descending score)
    must be identical.  The scores of the two implementations agree to ~1e-6 relative (different fp32 summation orders;
    asserted separately), so two candidates whose scores are closer than `tie_rel` have no defined mutual order: such
    swaps are the ONLY tolerated difference, and every one is counted and printed.  boundary=True (a top-k list): an id
    may additionally drop out / come in at the cut if its score is tied with the last kept score."""
    got_ids, ref_ids = [tuple(np.atleast_1d(g).tolist()) for g in got_ids], [tuple(np.atleast_1d(r).tolist()) for r in ref_ids]
    gs, rs = np.asarray(got_scores, np.float64), np.asarray(ref_scores, np.float64)
    assert len(got_ids) == len(ref_ids), f'{name}: kept {len(got_ids)} vs reference {len(ref_ids)}'
    if got_ids == ref_ids:
        print(f'{name}: all {len(ref_ids)} kept indices identical, in identical order')
        return 0

    def tied(a, b):
        return abs(a - b) <= tie_rel * max(abs(a), abs(b), 1e-30)
    gpos, rpos = {g: i for i, g in enumerate(got_ids)}, {r: i for i, r in enumerate(ref_ids)}
    only_g, only_r = [g for g in got_ids if g not in rpos], [r for r in ref_ids if r not in gpos]
    if only_g or only_r:
        assert boundary, f'{name}: kept sets differ: only here {only_g[:5]}, only reference {only_r[:5]}'
        for g in only_g:
            assert tied(gs[gpos[g]], gs.min()), f'{name}: {g} is kept here only and is not tied with the cut ({gs[gpos[g]]!r} vs {gs.min()!r})'
        for r in only_r:
            assert tied(rs[rpos[r]], rs.min()), f'{name}: {r} is kept by the reference only and is not tied with the cut'
    moved = [(i, gpos[r]) for i, r in enumerate(ref_ids) if r in gpos and gpos[r] != i]
    for i, j in moved:
        lo, hi = min(i, j), max(i, j)
        assert tied(rs[lo], rs[hi]) and tied(gs[lo], gs[hi]), \
            f'{name}: the id at reference position {i} sits at {j} here and the scores in between are not tied ({rs[lo]!r} .. {rs[hi]!r})'
    print(f'{name}: {len(ref_ids)} kept; {len(only_r)} swapped at the cut, {len(moved)} positions permuted among score ties '
          f'(|ds| <= {tie_rel:g} rel): {moved[:8]}')
    return len(moved) + len(only_r)
