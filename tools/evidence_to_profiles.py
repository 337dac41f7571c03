#!/usr/bin/env python
"""Copy the judged summaries of a tools/evidence_r3b.sh bundle into profiles/ under a round prefix and write the bench-lines table.
  python tools/evidence_to_profiles.py gpurun_out/evidence_r03c r03b"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    E, pre = sys.argv[1], sys.argv[2]
    P = os.path.join(ROOT, 'profiles')
    for src, dst in (('kernel_trace.md', 'bench_kernel_trace.md'), ('pmc.md', 'bench_pmc.md'), ('pmc.json', 'bench_pmc.json'),
                     ('pmc_wino.md', 'bench_pmc_wino.md'), ('pmc_wino.json', 'bench_pmc_wino.json'), ('other.jsonl', 'bench_other_configs.jsonl'),
                     ('other_f32_operands.jsonl', 'bench_other_configs_f32_operands.jsonl'), ('trunk_layers_kitti.md', 'trunk_layers_kitti.md')):
        if os.path.exists(os.path.join(E, src)):
            shutil.copy(os.path.join(E, src), os.path.join(P, f'{pre}_{dst}'))
    if os.path.exists(os.path.join(E, 'pair_ab.log')):
        with open(os.path.join(P, f'{pre}_pair_ab.log'), 'w') as f:
            f.writelines(l for l in open(os.path.join(E, 'pair_ab.log')) if 'amdgpu.ids' not in l)
    rows = [('default: `python bench.py --steps 20 --warmup 5` (simple_test -> ivx_model_detect, eager; Winograd-domain GEMMs on fp16 pair operands)', 'bench_default.json'),
            ('`--wino-operands f32` (fp32 MFMA in the Winograd domain: the round-2 arithmetic)', 'bench_f32_operands.json'),
            ('--api composed (layer by layer over the op-level ABI; every input stage reduces its tensor itself)', 'bench_composed.json'),
            ('--storage bf16 (optional reduced-precision mode; NOT the headline)', 'bench_bf16.json'),
            ('IVX_BENCH_FORCE_DIST=1 under torch.distributed.run, world size 1 (RCCL all-gather in every step)', 'bench_dist1.json'),
            (f'the run under rocprofv3 --kernel-trace --stats (profiles/{pre}_bench_kernel_trace.md)', 'bench_profiled.json')]
    out = [f'# Round 3, second half (split-operand MFMA): bench lines (one MI355X box, same session; tools/evidence_r3b.sh -> {E})', '',
           '| run | images/s | ms/step | GEMM: products TFLOP/s (frac of its MFMA peak) | fp32-equivalent TFLOP/s | GEMM ms | neck ms | transforms ms (GB/s) | trunk ms |',
           '|---|---|---|---|---|---|---|---|---|']
    for name, f in rows:
        try:
            r = json.load(open(os.path.join(E, f)))
        except Exception:
            continue
        ro, xf, t2 = r.get('roofline') or {}, r.get('roofline_winograd_transforms') or {}, r.get('roofline_trunk_2d') or {}
        out.append(f"| {name} | {r['value']} | {r['ms_per_step']} | {ro.get('achieved')} ({ro.get('frac')} of {ro.get('peak')}) | "
                   f"{ro.get('fp32_equivalent_tflops') or ro.get('achieved')} | {ro.get('mfma_launch_ms_per_step')} | {ro.get('neck_ms_per_step')} | "
                   f"{xf.get('ms_per_step')} ({xf.get('achieved')}) | {t2.get('ms_per_step')} |")
    d = json.load(open(os.path.join(E, 'bench_default.json')))
    alt = d.get('exact_fp32_mfma') or {}
    out += ['', f"The default run times the fp32-MFMA form after its timed region (`exact_fp32_mfma`): {alt.get('value')} images/s, same detections: "
                f"{alt.get('same_detections_as_default')}.", '',
            'Earlier boxes of this half-round, default command: 170.9 (first pair GEMMs, tensor-wide max-reduction with per-wave atomics), 186.1 (reduction '
            'rewritten), 194.6 (one atomic per workgroup, 256x64 tile for Cout 64), 199.7 / 200.2 / 198.8 (maxima handed over by the producing output '
            'transform), 203.1 / 204.8 (z-halo kernel for the stride-1 layers), 215.1 (overlapping halo tiles: a fourth workgroup per CU; input transform '
            'without the redundant saturation).', '',
            f'## Other workloads (images/s; `profiles/{pre}_bench_other_configs{{,_f32_operands}}.jsonl`)', '',
            '| workload | fp16-pair operands (default) | fp32 MFMA operands |', '|---|---|---|']
    a = [json.loads(l) for l in open(os.path.join(E, 'other.jsonl'))]
    b = [json.loads(l) for l in open(os.path.join(E, 'other_f32_operands.jsonl'))]
    for r in a:
        m = [q for q in b if q['config']['workload'] == r['config']['workload'] and q['config'].get('views') == r['config'].get('views')]
        out.append(f"| {r['config']['workload']} x{r['config'].get('views')} views | {r['value']} ({r['ms_per_step']} ms/scene) | {m[0]['value'] if m else '-'} |")
    out += ['', '## The default line in full', '', '```json', json.dumps(d, indent=1), '```', '']
    with open(os.path.join(P, f'{pre}_bench_lines.md'), 'w') as f:
        f.write('\n'.join(out))
    print('wrote', os.path.join(P, f'{pre}_bench_lines.md'))


if __name__ == '__main__':
    main()
