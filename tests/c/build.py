"""Build the C test program(s) of tests/c against the in-tree libimvoxel_hip.so (plain gcc; no Python at run time)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'imvoxelnet_amd', 'csrc')
ROCM = os.environ.get('ROCM_PATH', '/opt/rocm')


def build(name='e2e_small', force=False):
    src, exe = os.path.join(HERE, name + '.c'), os.path.join(HERE, name)
    lib = os.path.join(CSRC, 'libimvoxel_hip.so')
    if not force and os.path.exists(exe) and os.path.getmtime(exe) > max(os.path.getmtime(src), os.path.getmtime(lib)):
        return exe
    cmd = ['gcc', '-std=c11', '-O1', '-D__HIP_PLATFORM_AMD__', f'-I{ROCM}/include', src, '-o', exe, f'-L{CSRC}', '-limvoxel_hip',
           f'-L{ROCM}/lib', '-lamdhip64', '-lm', f'-Wl,-rpath,{CSRC}', f'-Wl,-rpath,{ROCM}/lib', '-Wl,-rpath,$ORIGIN/../../imvoxelnet_amd/csrc']
    subprocess.check_call(cmd)
    return exe


if __name__ == '__main__':
    print(build(force=True))
