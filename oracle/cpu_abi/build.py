"""TEST INFRASTRUCTURE.  Build oracle/_cpuabi/libimvoxel_cpu.so: the model-level handle of the product (csrc/model.cpp, compiled
UNCHANGED against the host-memory stand-in of the HIP runtime in oracle/cpu_abi/hip/) over the CPU restatement of the op-level
entry points (cpu_ops.cpp + oracle/ivx_oracle.c), and tests/c/e2e_small_cpu / e2e_indoor_cpu: the same C host programs as
tests/c/e2e_small.c / e2e_indoor.c, linked against it.  Only tests/ use the result; the product loads csrc/libimvoxel_hip.so and has no CPU fallback."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'imvoxelnet_amd', 'csrc')
OUT = os.path.join(ROOT, 'oracle', '_cpuabi')
LIB = os.path.join(OUT, 'libimvoxel_cpu.so')
EXE = os.path.join(ROOT, 'tests', 'c', 'e2e_small_cpu')


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = [os.path.join(CSRC, 'model.cpp'), os.path.join(CSRC, 'api_common.cpp'), os.path.join(HERE, 'cpu_ops.cpp')]
    oracle_c = os.path.join(ROOT, 'oracle', 'ivx_oracle.c')
    deps = srcs + [oracle_c, os.path.join(HERE, 'hip', 'hip_runtime_api.h'), os.path.join(ROOT, 'include', 'imvoxel.h')]
    if force or _stale(LIB, deps):
        obj = os.path.join(OUT, 'ivx_oracle.o')
        subprocess.check_call(['gcc', '-O2', '-fPIC', '-std=c11', '-ffp-contract=off', '-fno-fast-math', '-fopenmp', '-c', oracle_c, '-o', obj])
        subprocess.check_call(['g++', '-O3', '-mavx2', '-fPIC', '-shared', '-std=c++17', '-ffp-contract=off', '-fno-fast-math', '-fopenmp', f'-I{HERE}'] + srcs +
                              [obj, '-o', LIB, '-lm', '-Wl,--no-undefined'])
    for name in ('e2e_small', 'e2e_indoor'):         # the Python-free C hosts of tests/c, linked against the CPU restatement
        c_src, exe = os.path.join(ROOT, 'tests', 'c', name + '.c'), os.path.join(ROOT, 'tests', 'c', name + '_cpu')
        if force or _stale(exe, [c_src, LIB]):
            subprocess.check_call(['gcc', '-std=gnu11', '-O1', f'-I{HERE}', c_src, '-o', exe, f'-L{OUT}', '-limvoxel_cpu', '-lm', f'-Wl,-rpath,{OUT}',
                                   '-Wl,-rpath,$ORIGIN/../../oracle/_cpuabi'])
    return LIB, EXE


if __name__ == '__main__':
    print(build(force=True))
