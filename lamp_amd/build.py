"""Build liblamp_hip.so (the C-ABI HIP library) for gfx950 with hipcc, in-tree.

    python -m lamp_amd.build            # rebuild what is older than its sources
    python -m lamp_amd.build --force

Two libraries come out of the same sources:
  liblamp_hip.so         the product: exactly the entry points include/lamp_hip.h declares.
  liblamp_hip_tuning.so  the same code compiled with -DLAMP_TUNING: additionally exports the lamp_debug_* hooks
                         (force a GEMM tile / attention variant, per-workgroup timelines) that tools/bench_kernels.py
                         and the every-variant tests use.  Nothing in lamp_amd/ loads it.
Translation units are compiled in parallel (one hipcc process each) and linked afterwards.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'liblamp_hip.so')
LIB_TUNING = os.path.join(HERE, 'liblamp_hip_tuning.so')
SOURCES = ['gemm.hip', 'gemm_gen.hip', 'attention.hip', 'attention_tile.hip', 'attention_small.hip', 'attention_general.hip', 'pointwise.hip',
           'backward.hip', 'chain.hip', 'api.hip']
TUNING_SOURCES = {'gemm.hip', 'attention.hip', 'attention_tile.hip', 'attention_small.hip', 'chain.hip'}
TUNING_ONLY = ['slab.hip']   # experiments kept bit-identical and benchmarkable, not part of the product library   # the units that contain LAMP_TUNING code
HEADERS = [os.path.join(CSRC, 'lamp_kernels.h'), os.path.join(CSRC, 'lamp_asm.h'), os.path.join(HERE, '..', 'include', 'lamp_hip.h')]
REMARKS = ['-Rpass-analysis=kernel-resource-usage']   # per-kernel VGPR / AGPR / scratch report, saved beside each object
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-fvisibility-inlines-hidden', '-Wno-unused-result']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    deps = [os.path.join(CSRC, s) for s in SOURCES + TUNING_ONLY] + HEADERS
    return _newer(LIB, deps) or _newer(LIB_TUNING, deps) or not all(os.path.exists(resources_path(s)) for s in SOURCES)


def _run(cmd, verbose, remarks=None):
    if verbose:
        print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + ' '.join(cmd) + '\n' + r.stdout + r.stderr)
    if remarks:   # the compiler's per-kernel register / scratch report of THIS object (tests/test_kernel_resources.py)
        with open(remarks, 'w') as f:
            f.write(r.stderr)


def resources_path(source, tuning=False):
    """Where build() keeps hipcc's -Rpass-analysis=kernel-resource-usage report of one translation unit."""
    return os.path.join(OBJ, source.replace('.hip', '.tuning.resources.txt' if tuning else '.resources.txt'))


def kernel_resources(source, tuning=False):
    """{demangled kernel name: {'vgpr', 'agpr', 'sgpr', 'scratch', 'occupancy', 'lds'}} of one built translation unit, parsed
    from the report build() saved next to its object file."""
    import re
    rows, cur = {}, None
    with open(resources_path(source, tuning)) as f:
        text = f.read()
    names = re.findall(r'Function Name: (\S+)', text)
    demangled = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    table = dict(zip(names, demangled))
    for line in text.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            name = table.get(m.group(1), m.group(1)).replace('(anonymous namespace)::', '')
            cur = rows.setdefault(name.split('(')[0].replace('void ', ''), {})
            continue
        for key, pat in (('sgpr', r'TotalSGPRs: (\d+)'), ('vgpr', r' VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'),
                         ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'), ('occupancy', r'Occupancy \[waves/SIMD\]: (\d+)'),
                         ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return rows


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link both shared libraries."""
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    cc = _hipcc()
    jobs = []   # (object, command)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        deps = [src] + HEADERS
        obj = os.path.join(OBJ, s.replace('.hip', '.o'))
        if force or _newer(obj, deps) or not os.path.exists(resources_path(s)):
            jobs.append(([cc] + FLAGS + REMARKS + ['-c', src, '-o', obj], resources_path(s)))
        if s in TUNING_SOURCES:
            tobj = os.path.join(OBJ, s.replace('.hip', '.tuning.o'))
            if force or _newer(tobj, deps) or not os.path.exists(resources_path(s, True)):
                jobs.append(([cc] + FLAGS + REMARKS + ['-DLAMP_TUNING', '-c', src, '-o', tobj], resources_path(s, True)))
    for s in TUNING_ONLY:
        src = os.path.join(CSRC, s)
        tobj = os.path.join(OBJ, s.replace('.hip', '.tuning.o'))
        if force or _newer(tobj, [src] + HEADERS) or not os.path.exists(resources_path(s, True)):
            jobs.append(([cc] + FLAGS + REMARKS + ['-DLAMP_TUNING', '-c', src, '-o', tobj], resources_path(s, True)))
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as ex:
        list(ex.map(lambda j: _run(j[0], verbose, j[1]), jobs))
    objs = [os.path.join(OBJ, s.replace('.hip', '.o')) for s in SOURCES]
    tobjs = [os.path.join(OBJ, s.replace('.hip', '.tuning.o' if s in TUNING_SOURCES else '.o')) for s in SOURCES]
    tobjs += [os.path.join(OBJ, s.replace('.hip', '.tuning.o')) for s in TUNING_ONLY]
    _run([cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs, verbose)
    _run([cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_TUNING] + tobjs, verbose)
    return LIB


if __name__ == '__main__':
    try:
        print(build(force='--force' in sys.argv, verbose=True))
    except RuntimeError as e:   # the LAST lines must say so (callers pipe this through tail)
        print(str(e)[-4000:], file=sys.stderr)
        print('BUILD FAILED')
        sys.exit(1)
