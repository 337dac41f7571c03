#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV passes (tools/pmc_bench.sh / pmc_conv.sh) per kernel name.
  pmc_summary.py <dir with pass*/p_counter_collection.csv> [--min-ms 1.0] [--json out.json] > profiles/xxx.md
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte
requests at 64 B, so the read side is doubled before use; WRITE_SIZE is used as reported (it matches the
algorithmic output bytes of the conv launches exactly)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from kname import pretty
import argparse
import collections
import csv
import glob
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('--min-ms', type=float, default=0.0)
    ap.add_argument('--match', default='conv_igemm,conv_wino_halo,conv_wino_zblk', help='comma-separated substrings of the kernel names to keep')
    ap.add_argument('--json', default=None)
    ap.add_argument('--steps', type=int, default=0, help='model steps the profiled command ran (written to the JSON as _meta.steps: per-step sums)')
    a = ap.parse_args()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for p in sorted(glob.glob(os.path.join(a.dir, 'pass*'))):
        if not os.path.isdir(p):
            continue
        trace = glob.glob(os.path.join(p, '*kernel_trace.csv'))
        cc = glob.glob(os.path.join(p, '*counter_collection.csv'))
        if not trace or not cc:
            continue
        d = {}
        for r in csv.DictReader(open(trace[0])):
            d[r['Dispatch_Id']] = ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r['Kernel_Name'])
        for r in csv.DictReader(open(cc[0])):
            ms, name = d.get(r['Dispatch_Id'], (0.0, r['Kernel_Name']))
            name = pretty(name)
            if not any(mm in name for mm in a.match.split(',')) or ms < a.min_ms:
                continue
            key = name.replace('(anonymous namespace)::', '').split('(')[0][:80]
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] in ('GRBM_GUI_ACTIVE', 'FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS_VALU'):
                dur[(key, r['Counter_Name'])].append(ms)
    out = {}
    print(f'# rocprofv3 --pmc summary ({a.dir}; kernels matching "{a.match}", launches >= {a.min_ms} ms)\n')
    for key, cs in agg.items():
        m = {k: sum(v) / len(v) for k, v in cs.items()}
        n = {k: len(v) for k, v in cs.items()}
        print(f'## `{key}`  ({max(n.values())} launches averaged)\n')
        print('| counter | mean per launch |')
        print('|---|---|')
        for k in sorted(m):
            print(f'| {k} | {m[k]:.4g} |')
        der = {}
        if 'GRBM_GUI_ACTIVE' in m and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
            cyc = m['GRBM_GUI_ACTIVE'] / 8.0                 # summed over 8 XCDs
            der['mfma_pipe_busy_frac'] = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024)   # 256 CUs x 4 SIMDs
            ms = sum(dur[(key, 'GRBM_GUI_ACTIVE')]) / len(dur[(key, 'GRBM_GUI_ACTIVE')])
            der['avg_launch_ms_profiled'] = ms
            der['effective_clock_ghz'] = cyc / (ms * 1e-3) / 1e9
        if 'FETCH_SIZE' in m:
            der['hbm_read_bytes'] = 2.0 * m['FETCH_SIZE'] * 1024
        if 'WRITE_SIZE' in m:
            der['hbm_write_bytes'] = m['WRITE_SIZE'] * 1024
        if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
            der['hbm_bytes'] = der['hbm_read_bytes'] + der['hbm_write_bytes']
        if 'TCC_HIT_sum' in m and 'TCC_MISS_sum' in m:
            der['l2_hit_rate'] = m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum'])
        if 'SQ_LDS_BANK_CONFLICT' in m and m.get('SQ_LDS_IDX_ACTIVE'):
            der['lds_bank_conflict_frac'] = m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']
        if 'SQ_INSTS_VALU' in m and m.get('SQ_INSTS_MFMA'):
            der['valu_per_mfma'] = m['SQ_INSTS_VALU'] / m['SQ_INSTS_MFMA']
        print('\n| derived | value |')
        print('|---|---|')
        for k, v in der.items():
            print(f'| {k} | {v:.4g} |')
        print()
        out[key] = dict(counters=m, derived=der, launches=max(n.values()))
    if a.steps:
        out['_meta'] = {'steps': a.steps}
    if a.json:
        json.dump(out, open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
