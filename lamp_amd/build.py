"""Build liblamp_hip.so (the C-ABI HIP library) for gfx950 with hipcc, in-tree.

    python -m lamp_amd.build            # rebuild if sources are newer than the .so
    python -m lamp_amd.build --force
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'liblamp_hip.so')
SOURCES = ['gemm.hip', 'gemm_gen.hip', 'attention.hip', 'attention_general.hip', 'pointwise.hip', 'backward.hip', 'api.hip']
HEADERS = [os.path.join(CSRC, 'lamp_kernels.h'), os.path.join(HERE, '..', 'include', 'lamp_hip.h')]


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link the shared library."""
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-Wno-unused-result', '-o', LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + r.stdout + r.stderr)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
