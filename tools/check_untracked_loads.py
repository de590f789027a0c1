#!/usr/bin/env python3
"""ISA check of the hand-scheduled kernels: no instruction may read the destination registers of an inline-assembly load
before a wait follows that load.

chain.hip issues its W stream, its LDS fragment reads and its LayerNorm operand reads from inline assembly with hand-counted
s_waitcnt: the compiler does not know these registers are in flight.  The failure this guards against (seen twice: round 4's
four-wave geometry, round 5's LayerNorm operands read under an `if`) is a register COPY -- a phi at a control-flow merge, an
AGPR park, a spill -- that the compiler places right behind the load, before the data has landed: silently wrong, not even
repeatable.  The check walks the device assembly (hipcc -S): inside every kernel whose name matches, for every load between
;;#ASMSTART / ;;#ASMEND markers, no later instruction of the same basic block may mention the load's destination registers
as a SOURCE until an s_waitcnt has been passed.  (A wait does not prove the right count -- the bit-identity tests do that --
but a read with NO wait in between is always wrong.)

    python tools/check_untracked_loads.py [file.s | file.hip] [kernel-name substring ...]      exit 1 on a finding
"""
import re
import subprocess
import sys
import tempfile

LOAD = re.compile(r'^\s*(buffer_load_dword\w*|global_load_dword\w*|ds_read\w*)\s+(v\[\d+:\d+\]|v\d+)\s*,\s*(.*)$')
REG = re.compile(r'v\[(\d+):(\d+)\]|v(\d+)')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def device_asm(path, extra=()):
    if path.endswith('.s'):
        return open(path).read()
    with tempfile.NamedTemporaryFile(suffix='.s') as f:
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', f.name, path,
                        *extra], check=True, capture_output=True)
        return open(f.name).read()


def check(asm, wanted=('chain',)):
    """-> (kernels checked, loads checked, findings: [(kernel, load line no, load, reader line no, reader)])"""
    findings, n_loads, kernels = [], 0, 0
    lines = asm.split('\n')
    kernel, in_asm = None, False
    pending = []   # (dest regs, line no, text) of loads no wait has followed yet
    for no, line in enumerate(lines, 1):
        t = line.strip()
        m = re.match(r'^(_Z\w+):', line)
        if m:
            name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
            kernel = name if any(w in name for w in wanted) and 'kernel' in name else None
            kernels += kernel is not None
            pending = []
            continue
        if kernel is None or not t or t.startswith(';') and 'ASM' not in t:
            continue
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if t.startswith('.') or t.endswith(':'):          # label: a new basic block (the scan is per block)
            pending = []
            continue
        if t.startswith('s_endpgm'):
            kernel = None
            continue
        if t.startswith('s_waitcnt'):
            pending = []
            continue
        if t.startswith(('s_cbranch', 's_branch', 's_barrier')):
            if t.startswith('s_barrier'):
                continue
            pending = []
            continue
        m = LOAD.match(t)
        if m:
            dest, rest = regs_of(m.group(2)), m.group(3)
            for regs, lno, ltxt in pending:          # the address operands of this load are sources too
                if regs & regs_of(rest.split(';')[0]):
                    findings.append((kernel, lno, ltxt, no, t))
            # a later load that re-targets the same registers simply replaces the entry (write after write)
            pending = [(r, l, x) for r, l, x in pending if not (r & dest)]
            if in_asm:
                pending.append((dest, no, t))
                n_loads += 1
            continue
        if not pending:
            continue
        ops = t.split(None, 1)
        if len(ops) < 2:
            continue
        operands = ops[1].split(';')[0]
        parts = [p.strip() for p in operands.split(',')]
        # first operand is the destination for VALU / MFMA / DS-read style instructions; stores have sources only
        is_store = ops[0].startswith(('ds_write', 'buffer_store', 'global_store', 'scratch_store'))
        srcs = regs_of(','.join(parts if is_store else parts[1:]))
        dsts = set() if is_store else regs_of(parts[0])
        for regs, lno, ltxt in pending:
            if regs & srcs:
                findings.append((kernel, lno, ltxt, no, t))
            elif regs & dsts and not ops[0].startswith('v_mfma'):
                # overwriting a register whose load is still in flight (the landing data would clobber the new value)
                findings.append((kernel, lno, ltxt, no, t + '   ; WRITES a register in flight'))
    return kernels, n_loads, findings


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else 'lamp_amd/csrc/chain.hip'
    wanted = tuple(sys.argv[2:]) or ('chain',)
    kernels, n_loads, findings = check(device_asm(path), wanted)
    print('%d kernels, %d inline-assembly loads checked, %d findings' % (kernels, n_loads, len(findings)))
    for k, lno, ltxt, no, t in findings[:40]:
        print('  %s\n    line %d: %s\n    line %d: %s' % (k[:90], lno, ltxt, no, t))
    sys.exit(1 if findings else 0)


if __name__ == '__main__':
    main()
