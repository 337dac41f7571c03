#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max, and
per-(kernel, grid) rows for the conv kernels.  Usage: rocpd_summary.py results.db [steps] [> profiles/xxx.md]
With `steps` (timed + warm-up steps of the profiled bench.py run) it also prints the per-step total of the conv launches
of the 3-D neck -- every launch >= 0.3 ms of the two kernels that run the Winograd-domain GEMMs, `conv_wino_halo_kernel` (the
stride-1 / z-stride-2 layers on fp16 pairs) and `conv_igemm_v4_kernel` (the last layer, and every layer with fp32 operands) --
plus the transforms: the numbers bench.py's event-bracketed `roofline.mfma_launch_ms_per_step` / `neck_ms_per_step` must agree with."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from kname import pretty
import sqlite3
import sys


def main(path, steps=0):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
    name = 'name' if 'name' in cols else 'kernel_name'
    q = f'select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name} order by 3 desc'
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows)
    print(f'# rocprofv3 --kernel-trace summary ({path.split("/")[-1]})\n')
    print('| kernel | calls | total ms | avg us | min us | max us | % |')
    print('|---|---|---|---|---|---|---|')
    for n, k, t, a, mn, mx in rows:
        print(f'| `{pretty(n)[:110]}` | {k} | {t / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * t / tot:.2f} |')
    gcols = [x for x in ('grid_x', 'grid_size_x', 'grid_size') if x in cols]
    gx = gy = None
    if gcols:
        gx = gcols[0]
        gy = gx.replace('x', 'y') if 'x' in gx else None
        sel = f'{name}, {gx}' + (f', {gy}' if gy and gy in cols else '')
        print('\n## conv kernel by launch shape\n')
        print('| kernel | grid | calls | avg us |')
        print('|---|---|---|---|')
        for r in c.execute(f"select {sel}, count(*), avg(end-start) from kernels where ({name} like '%conv_igemm%' or {name} like '%conv_wino_halo%' or {name} like '%conv_wino_zblk%') group by {sel} order by avg(end-start) desc"):
            print(f'| `{pretty(r[0])[:60]}` | {r[1:-2]} | {r[-2]} | {r[-1] / 1e3:.1f} |')


    if steps:
        gemm = f"({name} like '%conv_igemm%' or {name} like '%conv_wino_halo%' or {name} like '%conv_wino_zblk%')"
        big = c.execute(f"select count(*), sum(end-start) from kernels where {gemm} and end-start >= 300000").fetchone()
        halo = c.execute(f"select count(*), sum(end-start) from kernels where {name} like '%conv_wino_halo%' or {name} like '%conv_wino_zblk%'").fetchone()
        gy_ok = gcols and gy and gy in cols
        tail = c.execute(f"select count(*), sum(end-start) from kernels where {name} like '%conv_igemm%' and end-start < 300000 and {gy} > 1 and {gx} <= 65536").fetchone() if gy_ok else (0, 0)
        red = c.execute(f"select count(*), sum(end-start) from kernels where {name} like '%splitk_reduce%'").fetchone()
        xin = c.execute(f"select count(*), sum(end-start) from kernels where {name} like '%wino_input_kernel%' and end-start >= 100000").fetchone()
        xout = c.execute(f"select count(*), sum(end-start) from kernels where ({name} like '%wino_output_kernel%' or {name} like '%wino_output_buf_kernel%') and end-start >= 100000").fetchone()
        print(f'\n## 3-D neck reconciliation ({steps} steps profiled)\n')
        print('| launches | per step | ms per step | avg ms |')
        print('|---|---|---|---|')
        print(f'| GEMM launches >= 0.3 ms of conv_wino_halo_kernel / conv_wino_zblk_kernel + conv_igemm_v4_kernel (the 9 neck layers: grouped Winograd-domain GEMMs, or direct convs) | {big[0] / steps:.1f} | '
              f'{(big[1] or 0) / 1e6 / steps:.3f} | {(big[1] or 0) / 1e6 / max(big[0], 1):.4f} |')
        print(f'| ... of which conv_wino_halo_kernel + conv_wino_zblk_kernel (all their launches) | {halo[0] / steps:.1f} | {(halo[1] or 0) / 1e6 / steps:.3f} | {(halo[1] or 0) / 1e6 / max(halo[0], 1):.4f} |')
        print(f'| wino_input_kernel launches >= 0.1 ms | {xin[0] / steps:.1f} | {(xin[1] or 0) / 1e6 / steps:.3f} | {(xin[1] or 0) / 1e6 / max(xin[0], 1):.4f} |')
        print(f'| wino_output_kernel / wino_output_buf_kernel launches >= 0.1 ms | {xout[0] / steps:.1f} | {(xout[1] or 0) / 1e6 / steps:.3f} | {(xout[1] or 0) / 1e6 / max(xout[0], 1):.4f} |')
        print(f'| K-split launches (small 2-D layers) | {tail[0] / steps:.1f} | {(tail[1] or 0) / 1e6 / steps:.3f} | |')
        print(f'| split-K reductions | {red[0] / steps:.1f} | {(red[1] or 0) / 1e6 / steps:.3f} | |')
        tot_n = ((big[1] or 0) + (xin[1] or 0) + (xout[1] or 0)) / 1e6 / steps
        print(f'\nneck kernel time per step ~ {tot_n:.2f} ms = GEMM launches + transforms; bench.py brackets the same launches with HIP events: '
              '`roofline.mfma_launch_ms_per_step` is the first row (`roofline.avg_launch_ms` its avg ms column, `roofline.launches_per_step` its count), '
              '`roofline_winograd_transforms.ms_per_step` the sum of the transform rows, `roofline.neck_ms_per_step` the total.')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
