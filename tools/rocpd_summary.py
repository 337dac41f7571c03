#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max, and
per-(kernel, grid) rows for the conv kernel.  Usage: rocpd_summary.py results.db [> profiles/xxx.md]"""
import sqlite3
import sys


def main(path):
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
        print(f'| `{n[:110]}` | {k} | {t / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * t / tot:.2f} |')
    gcols = [x for x in ('grid_x', 'grid_size_x', 'grid_size') if x in cols]
    if gcols:
        gx = gcols[0]
        gy = gx.replace('x', 'y') if 'x' in gx else None
        sel = f'{name}, {gx}' + (f', {gy}' if gy and gy in cols else '')
        print('\n## conv kernel by launch shape\n')
        print('| kernel | grid | calls | avg us |')
        print('|---|---|---|---|')
        for r in c.execute(f"select {sel}, count(*), avg(end-start) from kernels where {name} like '%conv_igemm%' group by {sel} order by avg(end-start) desc"):
            print(f'| `{r[0][:60]}` | {r[1:-2]} | {r[-2]} | {r[-1] / 1e3:.1f} |')


if __name__ == '__main__':
    main(sys.argv[1])
