"""Build libimvoxel_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libimvoxel_hip.so')
ARCH = 'gfx950'

# (source, extra flags[, object name]).  The geometry / index kernels must keep the reference's operation order.
# conv_igemm.hip is compiled once per kernel family (IVX_CONV_TU, see the top of that file): the instantiations of the LDS-DMA kernel
# dominate the build time and the families compile in parallel.
SOURCES = [
    ('conv_igemm.hip', ['-DIVX_CONV_TU=0'], 'conv_igemm.o'),
    ('conv_igemm.hip', ['-DIVX_CONV_TU=1'], 'conv_igemm_f32.o'),
    ('conv_igemm.hip', ['-DIVX_CONV_TU=2'], 'conv_igemm_lowp.o'),
    ('conv_igemm.hip', ['-DIVX_CONV_TU=3'], 'conv_igemm_pair_bf16.o'),
    ('conv_igemm.hip', ['-DIVX_CONV_TU=4'], 'conv_igemm_pair_f16.o'),
    ('conv_igemm.hip', ['-DIVX_CONV_TU=5'], 'conv_igemm_halo.o'),
    ('bottleneck.hip', []),
    ('stem.hip', []),
    ('fold4w.hip', []),
    ('winograd.hip', []),
    ('pool_layout.hip', []),
    ('backproject.hip', ['-ffp-contract=off']),
    ('anchor_tail.hip', ['-ffp-contract=off']),
    ('dcn.hip', []),
    ('ubench.hip', []),
    ('api_common.cpp', []),
    ('kitti_eval.cpp', []),
    ('model.cpp', ['-ffp-contract=off']),
]


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found; libimvoxel_hip.so cannot be built')
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, 'ivx_common.h'),
               os.path.join(os.path.dirname(os.path.dirname(CSRC)), 'include', 'imvoxel.h')]
    objs, jobs = [], []
    for entry in SOURCES:
        src, extra = entry[0], entry[1]
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, entry[2] if len(entry) > 2 else os.path.splitext(src)[0] + '.o')
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc, f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-c', s, '-o', o] + extra
            if src.endswith('.cpp'):
                cmd = [hipcc, '-O2', '-std=c++17', '-fPIC', '-c', s, '-o', o] + extra
            jobs.append(cmd)
    if jobs:
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
            list(ex.map(run, jobs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(verbose=True))
