"""Build liblamp_hip.so (the C-ABI HIP library) for gfx950 with hipcc, in-tree.

    python -m lamp_amd.build            # rebuild what is older than its sources
    python -m lamp_amd.build --force

Two libraries come out of the same sources:
  liblamp_hip.so         the product: exactly the entry points include/lamp_hip.h declares.
  liblamp_hip_tuning.so  the same code compiled with -DLAMP_TUNING: additionally exports the lamp_debug_* hooks
                         (force a GEMM tile / attention variant, per-workgroup timelines) that tools/bench_kernels.py
                         and the every-variant tests use.  Nothing in lamp_amd/ loads it.
Translation units are compiled in parallel (one hipcc process each) and linked afterwards.

The build FAILS -- no library is linked -- when a hand-scheduled kernel is not sound as THIS hipcc compiled it:
chain.hip / attention_tile.hip / slab.hip issue loads from inline assembly with hand-counted waits, which the compiler cannot
see.  After compiling, every rebuilt guarded unit goes through lamp_amd/isa_guard.py (no instruction touches an in-flight
register before its wait, no vector-written scalar feeds an inline-assembly load too early, m0 belongs to the LDS-DMA
requests alone) and through the compiler's own resource report (no scratch, no AGPRs in those kernels); the verdict is
stamped beside the objects together with `hipcc --version` (lamp_amd/build/toolchain.txt; bench.py reports it).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'liblamp_hip.so')
LIB_TUNING = os.path.join(HERE, 'liblamp_hip_tuning.so')
SOURCES = ['gemm.hip', 'gemm_gen.hip', 'attention.hip', 'attention_tile.hip', 'attention_small.hip', 'attention_general.hip', 'attention_sparse.hip',
           'pointwise.hip',
           'backward.hip', 'chain.hip', 'api.hip']
TUNING_SOURCES = {'gemm.hip', 'attention.hip', 'attention_tile.hip', 'attention_small.hip', 'attention_sparse.hip', 'chain.hip'}
TUNING_ONLY = ['experiments/slab.hip']   # experiments kept bit-identical and benchmarkable, never part of the product library
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
    return (_newer(LIB, deps) or _newer(LIB_TUNING, deps) or not all(os.path.exists(resources_path(s)) for s in SOURCES) or
            built_toolchain() is None)


def _run(cmd, verbose, remarks=None):
    if verbose:
        print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + ' '.join(cmd) + '\n' + r.stdout + r.stderr)
    if remarks:   # the compiler's per-kernel register / scratch report of THIS object (tests/test_kernel_resources.py)
        with open(remarks, 'w') as f:
            f.write(r.stderr)


def _stem(source):
    return os.path.basename(source)


def resources_path(source, tuning=False):
    """Where build() keeps hipcc's -Rpass-analysis=kernel-resource-usage report of one translation unit."""
    return os.path.join(OBJ, _stem(source).replace('.hip', '.tuning.resources.txt' if tuning else '.resources.txt'))


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


def toolchain():
    """`hipcc --version` of the compiler build() uses, first line + the clang line (recorded in build/toolchain.txt)."""
    try:
        out = subprocess.run([_hipcc(), '--version'], capture_output=True, text=True).stdout.splitlines()
    except OSError as e:
        return 'unknown (%s)' % e
    keep = [l.strip() for l in out if 'HIP version' in l or 'clang version' in l]
    return '; '.join(keep) or (out[0].strip() if out else 'unknown')


def built_toolchain():
    """The toolchain line stamped by the build that produced the libraries in the tree (None before the first build)."""
    try:
        with open(os.path.join(OBJ, 'toolchain.txt')) as f:
            return f.read().strip()
    except OSError:
        return None


GUARD_KERNELS = ('chain', 'slab', 'attn_tile')   # kernels with inline-assembly loads: registers must stay where the loads land


def _resource_problems(source, tuning):
    out = []
    for name, r in kernel_resources(source, tuning).items():
        if any(k in name for k in GUARD_KERNELS) and 'kernel' in name and (r.get('agpr', 0) or r.get('scratch', 0)):
            out.append('%s%s: %s uses %d AGPRs / %d bytes of scratch' % (source, ' (tuning)' if tuning else '', name,
                                                                         r.get('agpr', 0), r.get('scratch', 0)))
    return out


def _guard_stamp(source, tuning, tool):
    h = hashlib.sha256(tool.encode())
    for f in [os.path.join(CSRC, source)] + HEADERS + [os.path.join(HERE, 'isa_guard.py')]:
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return os.path.join(OBJ, _stem(source).replace('.hip', '.tuning.guard' if tuning else '.guard')), h.hexdigest()


def _guard(source, tuning, tool):
    """ISA + resource rules of one guarded unit; the verdict is stamped (sources + checker + toolchain), so an unchanged
    unit is not disassembled again.  -> list of problems."""
    from . import isa_guard
    stamp, digest = _guard_stamp(source, tuning, tool)
    try:
        with open(stamp) as f:
            if f.read().strip() == digest:
                return []
    except OSError:
        pass
    problems = isa_guard.guard_unit(os.path.join(CSRC, source), ('-DLAMP_TUNING',) if tuning else ())
    problems += _resource_problems(source, tuning)
    if not problems:
        with open(stamp, 'w') as f:
            f.write(digest)
    return problems


def verify(verbose=False):
    """Run the guards over every guarded unit of both libraries (stamped verdicts are reused) -> problems."""
    from . import isa_guard
    tool = toolchain()
    jobs = []
    for s in SOURCES + TUNING_ONLY:
        if _stem(s) not in isa_guard.GUARDED:
            continue
        if s in SOURCES:
            jobs.append((s, False))
        if s in TUNING_SOURCES or s in TUNING_ONLY:
            jobs.append((s, True))
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as ex:
        results = list(ex.map(lambda j: _guard(j[0], j[1], tool), jobs))
    problems = [p for r in results for p in r]
    if verbose:
        print('isa guard: %d units, %d problems (%s)' % (len(jobs), len(problems), tool), flush=True)
    return problems


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950, check the hand-scheduled kernels, link both shared libraries."""
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
        tobj = os.path.join(OBJ, _stem(s).replace('.hip', '.tuning.o'))
        if force or _newer(tobj, [src] + HEADERS) or not os.path.exists(resources_path(s, True)):
            jobs.append(([cc] + FLAGS + REMARKS + ['-DLAMP_TUNING', '-c', src, '-o', tobj], resources_path(s, True)))
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as ex:
        list(ex.map(lambda j: _run(j[0], verbose, j[1]), jobs))
    objs = [os.path.join(OBJ, s.replace('.hip', '.o')) for s in SOURCES]
    tobjs = [os.path.join(OBJ, s.replace('.hip', '.tuning.o' if s in TUNING_SOURCES else '.o')) for s in SOURCES]
    tobjs += [os.path.join(OBJ, _stem(s).replace('.hip', '.tuning.o')) for s in TUNING_ONLY]
    problems = verify(verbose)
    if problems:
        for lib in (LIB, LIB_TUNING):   # never leave a library of unsound kernels behind
            if os.path.exists(lib):
                os.remove(lib)
        raise RuntimeError('ISA guard: the hand-scheduled kernels are NOT sound as this hipcc compiled them (%s):\n  ' %
                           toolchain() + '\n  '.join(problems[:20]))
    _run([cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs, verbose)
    _run([cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_TUNING] + tobjs, verbose)
    with open(os.path.join(OBJ, 'toolchain.txt'), 'w') as f:
        f.write(toolchain() + '\n')
    return LIB


if __name__ == '__main__':
    try:
        print(build(force='--force' in sys.argv, verbose=True))
    except RuntimeError as e:   # the LAST lines must say so (callers pipe this through tail)
        print(str(e)[-4000:], file=sys.stderr)
        print('BUILD FAILED')
        sys.exit(1)
