"""Build libimvoxel_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libimvoxel_hip.so')
ARCH = 'gfx950'

# (source, extra flags).  The geometry / index kernels must keep the reference's operation order.
SOURCES = [
    ('conv_igemm.hip', []),
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
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + '.o')
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc, f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-c', s, '-o', o] + extra
            if src.endswith('.cpp'):
                cmd = [hipcc, '-O2', '-std=c++17', '-fPIC', '-c', s, '-o', o] + extra
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(verbose=True))
