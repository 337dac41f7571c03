#!/usr/bin/env python
"""Copy the judged summaries of a tools/evidence_r4.sh bundle into profiles/ under a round prefix and write the bench-lines table.
  python tools/evidence_to_profiles.py gpurun_out/evidence_r04 r04"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COPIES = (('kernel_trace_kitti.md', 'bench_kernel_trace.md'), ('kernel_trace_scannet_v1.md', 'kernel_trace_scannet_v1.md'),
          ('kernel_trace_nuscenes.md', 'kernel_trace_nuscenes.md'), ('pmc.md', 'bench_pmc.md'), ('pmc.json', 'bench_pmc.json'),
          ('pmc_wino.md', 'bench_pmc_wino.md'), ('pmc_wino.json', 'bench_pmc_wino.json'), ('pmc_trunk.md', 'bench_pmc_trunk.md'), ('pmc_all.md', 'bench_pmc_all.md'), ('pmc_all.json', 'bench_pmc_all.json'),
          ('bottleneck_ab.md', 'bottleneck_ab.md'), ('stem_ab.md', 'stem_ab.md'), ('other_no_fusion.jsonl', 'bench_other_configs_no_fusion.jsonl'),
          ('pmc_scannet_v1.md', 'pmc_scannet_v1.md'), ('pmc_scannet_v1.json', 'pmc_scannet_v1.json'),
          ('pmc_nuscenes.md', 'pmc_nuscenes.md'), ('pmc_nuscenes.json', 'pmc_nuscenes.json'),
          ('other.jsonl', 'bench_other_configs.jsonl'), ('other_f32_operands.jsonl', 'bench_other_configs_f32_operands.jsonl'),
          ('other_bf16.jsonl', 'bench_other_configs_bf16.jsonl'), ('trunk_layers_kitti.md', 'trunk_layers_kitti.md'),
          ('trunk_layers_scannet_v1.md', 'trunk_layers_scannet_v1.md'), ('host.txt', 'host.txt'),
          ('other_no_split_form.jsonl', 'bench_other_configs_no_split_form.jsonl'), ('neck_layers_scannet_v1.md', 'neck_layers_scannet_v1.md'),
          ('neck_layers_scannet_fast.md', 'neck_layers_scannet_fast.md'), ('neck_layers_sunrgbd_fast.md', 'neck_layers_sunrgbd_fast.md'),
          ('neck_layers_nuscenes.md', 'neck_layers_nuscenes.md'))


def load(E, f):
    try:
        return json.load(open(os.path.join(E, f)))
    except Exception:
        return None


def main():
    E, pre = sys.argv[1], sys.argv[2]
    P = os.path.join(ROOT, 'profiles')
    for src, dst in COPIES:
        if os.path.exists(os.path.join(E, src)):
            with open(os.path.join(P, f'{pre}_{dst}'), 'w') as f:
                f.writelines(l for l in open(os.path.join(E, src)) if 'amdgpu.ids' not in l)
    rows = [('default: `python bench.py --steps 20 --warmup 5` (simple_test -> ivx_model_detect; neck GEMMs and 2-D trunk on fp16 pair operands)', 'bench_default.json'),
            ('`--trunk-operands f32` (round 3: the 2-D trunk on fp32 MFMA, the neck GEMMs on pairs)', 'bench_trunk_f32.json'),
            ('`--wino-operands f32 --trunk-operands f32` (fp32 MFMA everywhere: the round-2 arithmetic)', 'bench_f32_operands.json'),
            ('--api composed (layer by layer over the op-level ABI)', 'bench_composed.json'),
            ('`IVX_FUSE_BOTTLENECK=0` (the five identity blocks of stages 1-2 as three launches each)', 'bench_no_fused_bottleneck.json'),
            ('`IVX_FUSE_STEM=0` (layout change, fp32-MFMA stem, max-pool as three launches)', 'bench_no_fused_stem.json'),
            ('`IVX_FUSE_STEM=0 IVX_FUSE_BOTTLENECK=0` (the round-5 trunk)', 'bench_no_fusion.json'),
            ('`IVX_BENCH_TRACE_TIMED=1` (round-5 placement of the stage events: a pair around every launch group of the TIMED steps)', 'bench_events_in_timed_region.json'),
            ('--storage bf16 (optional reduced-precision mode; NOT the headline)', 'bench_bf16.json'),
            ('IVX_BENCH_FORCE_DIST=1 under torch.distributed.run, world size 1 (RCCL all-gather in every step)', 'bench_dist1.json'),
            (f'the run under rocprofv3 --kernel-trace --stats (profiles/{pre}_bench_kernel_trace.md)', 'bench_profiled_kitti.json')]
    out = [f'# Round {int(pre[1:])}: bench lines (one MI355X box, one session; tools/evidence_r{int(pre[1:])}.sh -> {E})', '',
           '| run | images/s | ms/step | GEMM: products TFLOP/s (frac of its MFMA peak) | fp32-equivalent TFLOP/s | GEMM ms | neck ms | transforms ms (GB/s) | trunk ms |',
           '|---|---|---|---|---|---|---|---|---|']
    for name, f in rows:
        r = load(E, f)
        if not r:
            continue
        ro, xf, t2 = r.get('roofline') or {}, r.get('roofline_winograd_transforms') or {}, r.get('roofline_trunk_2d') or {}
        out.append(f"| {name} | {r['value']} | {r['ms_per_step']} | {ro.get('achieved')} ({ro.get('frac')} of {ro.get('peak')}) | "
                   f"{ro.get('fp32_equivalent_tflops') or ro.get('achieved')} | {ro.get('mfma_launch_ms_per_step')} | {ro.get('neck_ms_per_step')} | "
                   f"{xf.get('ms_per_step')} ({xf.get('achieved')}) | {t2.get('ms_per_step')} |")
    d = load(E, 'bench_default.json')
    alt = d.get('exact_fp32_mfma') or {}
    cb, cc = d.get('cpu_baseline') or {}, d.get('cpu_baseline_cabi') or {}
    out += ['', f"The default run times the all-fp32-MFMA form after its timed region (`exact_fp32_mfma`): {alt.get('value')} images/s, same detections: "
                f"{alt.get('same_detections_as_default')}.",
            f"CPU legs on this box ({cb.get('host_cpus')} hardware threads, {cb.get('physical_cores')} physical cores; {cb.get('cores')} threads used): oracle port "
            f"{cb.get('value')} images/s, C-ABI over the CPU restatement {cc.get('value')} images/s.", '',
            f'## Other workloads (images/s; `profiles/{pre}_bench_other_configs{{,_f32_operands,_bf16}}.jsonl`)', '',
            '| workload | default (pair operands: neck GEMMs + chained trunk) | ms/scene | trunk ms | neck ms | fp32 MFMA operands everywhere |', '|---|---|---|---|---|---|']
    a = [json.loads(l) for l in open(os.path.join(E, 'other.jsonl'))]
    b = [json.loads(l) for l in open(os.path.join(E, 'other_f32_operands.jsonl'))]
    for r in a:
        m = [q for q in b if q['config']['workload'] == r['config']['workload'] and q['config'].get('views') == r['config'].get('views')]
        out.append(f"| {r['config']['workload']} x{r['config'].get('views')} views | {r['value']} | {r['ms_per_step']} | {(r.get('roofline_trunk_2d') or {}).get('ms_per_step')} | "
                   f"{(r.get('roofline') or {}).get('neck_ms_per_step')} | {m[0]['value'] if m else '-'} |")
    if os.path.exists(os.path.join(E, 'other_no_fusion.jsonl')):
        out += ['', '| workload, `IVX_FUSE_STEM=0 IVX_FUSE_BOTTLENECK=0` (the round-5 trunk) | images/s | ms/scene | trunk ms |', '|---|---|---|---|']
        for l in open(os.path.join(E, 'other_no_fusion.jsonl')):
            r = json.loads(l)
            out.append(f"| {r['config']['workload']} x{r['config'].get('views')} views | {r['value']} | {r['ms_per_step']} | {(r.get('roofline_trunk_2d') or {}).get('ms_per_step')} |")
    if os.path.exists(os.path.join(E, 'other_no_split_form.jsonl')):
        out += ['', '| workload, `IVX_CONV_PAIR=0 IVX_CONV_SKINNY=0` (the 3x3x3 neck layers outside the Winograd form on fp32 MFMA, no K split for the Cout <= 32 head convs: before the last change of round 6) | images/s | ms/scene | neck ms |', '|---|---|---|---|']
        for l in open(os.path.join(E, 'other_no_split_form.jsonl')):
            r = json.loads(l)
            out.append(f"| {r['config']['workload']} x{r['config'].get('views')} views | {r['value']} | {r['ms_per_step']} | {(r.get('roofline') or {}).get('neck_ms_per_step')} |")
    if os.path.exists(os.path.join(E, 'other_bf16.jsonl')):
        out += ['', '| optional storage mode | images/s | ms/scene |', '|---|---|---|']
        for l in open(os.path.join(E, 'other_bf16.jsonl')):
            r = json.loads(l)
            out.append(f"| {r['config']['workload']} x{r['config'].get('views')} views, {r['dtype']} | {r['value']} | {r['ms_per_step']} |")
    out += ['', '## N-rank launch path at world size 1 (RCCL in the step; this pool has one GPU per box)', '',
            '| command | images/s | rccl_ranks | collective |', '|---|---|---|---|']
    for name, f in (('kitti --gpus 1', 'bench_dist1.json'), ('nuscenes --batch 1 (BASELINE config 4 as sharded)', 'bench_dist1_nuscenes.json'),
                    ('scannet_v1 --batch 2 (config 5 as sharded)', 'bench_dist1_scannet_v1.json'),
                    ('nuscenes --batch 1 --shard views (slab reduce-scatter path)', 'bench_dist1_nuscenes_views.json')):
        r = load(E, f)
        if r:
            out.append(f"| {name} | {r['value']} | {r['config'].get('rccl_ranks')} | {r['config'].get('collective')} |")
    out += ['', '## Runs under rocprofv3 (kernel trace)', '', '| workload | images/s under the profiler |', '|---|---|']
    for n in ('kitti', 'scannet_v1', 'nuscenes'):
        r = load(E, f'bench_profiled_{n}.json')
        if r:
            out.append(f"| {n} | {r['value']} |")
    out += ['', '## The default line in full', '', '```json', json.dumps(d, indent=1), '```', '']
    with open(os.path.join(P, f'{pre}_bench_lines.md'), 'w') as f:
        f.write('\n'.join(out))
    print('wrote', os.path.join(P, f'{pre}_bench_lines.md'))


if __name__ == '__main__':
    main()
