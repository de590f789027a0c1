"""ISA check of the hand-scheduled kernels: no instruction may read the destination registers of an inline-assembly load
before a wait follows that load.  lamp_amd.build runs it (guard_unit) on every build of chain.hip / attention_tile.hip /
slab.hip and FAILS the build on a finding: a different hipcc may schedule around the untracked loads differently, and such a
miscompile passes or fails parity nondeterministically.  tools/check_untracked_loads.py is the command line.

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
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get('HIPCC') or '/opt/rocm/bin/hipcc'

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
        subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', f.name, path,
                        *extra], check=True, capture_output=True)
        return open(f.name).read()


WAIT = re.compile(r'(vmcnt|lgkmcnt)\((\d+)\)')
SREG = re.compile(r's\[(\d+):(\d+)\]|\bs(\d+)\b')
VALU_SGPR_WAIT_STATES = 5   # VALU writes an SGPR (v_readlane, v_readfirstlane, carry / compare results) -> VMEM reads it


def sregs_of(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def pk_sources(parts):
    """Registers a packed instruction (v_pk_mul_f32 v[d:d+1], v[a:a+1], v[b:b+1] op_sel:[..] op_sel_hi:[..]) really reads: of a
    two-register source the low one feeds a result half whose select bit is 0, the high one a half whose bit is 1 (defaults
    op_sel = 0, op_sel_hi = 1) -- `v[92:93] ... op_sel_hi:[0,1]` broadcasts v92 and never touches v93."""
    text = ','.join(parts[1:])
    sel = {'op_sel': None, 'op_sel_hi': None}
    for key in sel:
        m = re.search(key + r':\[([01,]+)\]', text)
        if m:
            sel[key] = [int(x) for x in m.group(1).split(',')]
    text = re.sub(r'op_sel(_hi)?:\[[01,]+\]', '', text)
    out = set()
    for k, src in enumerate([x.strip() for x in text.split(',') if x.strip()]):
        m = re.match(r'v\[(\d+):(\d+)\]', src)
        if not m or int(m.group(2)) - int(m.group(1)) != 1:
            out |= regs_of(src)
            continue
        lo, hi = int(m.group(1)), int(m.group(2))
        a = sel['op_sel'][k] if sel['op_sel'] and k < len(sel['op_sel']) else 0
        b = sel['op_sel_hi'][k] if sel['op_sel_hi'] and k < len(sel['op_sel_hi']) else 1
        if a == 0 or b == 0:
            out.add(lo)
        if a == 1 or b == 1:
            out.add(hi)
    return out


def check(asm, wanted=('chain', 'slab')):
    """-> (kernels checked, loads checked, findings: [(kernel, load line no, load, reader line no, reader)]).

    Two in-order queues of inline-assembly loads, followed along the FALL-THROUGH path (reset at unconditional branches): vector memory (buffer / global loads) and LDS reads.  A wait
    `vmcnt(N)` / `lgkmcnt(N)` retires all but the N youngest of its queue (loads of a kind return in order; other operations the
    compiler issued only make the hardware count higher, i.e. retire less than assumed here -- the model errs towards missing a
    finding, never towards inventing one).  Until retired, a load's destination registers may be neither read nor written."""
    findings, n_loads, kernels = [], 0, 0
    lines = asm.split('\n')
    kernel, in_asm = None, False
    pend = {'vm': [], 'lgkm': []}   # [(dest regs, line no, text)]
    fresh = {}                      # SGPR -> (wait states since a VALU instruction wrote it, line no, text)

    def all_pending():
        return pend['vm'] + pend['lgkm']

    for no, line in enumerate(lines, 1):
        t = line.strip()
        m = re.match(r'^(_Z\w+):', line)
        if m:
            name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
            kernel = name if any(w in name for w in wanted) and 'kernel' in name else None
            kernels += kernel is not None
            pend = {'vm': [], 'lgkm': []}
            continue
        if kernel is None or not t or t.startswith(';') and 'ASM' not in t:
            continue
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if t.startswith('.') or t.endswith(':'):          # label: reached by fall-through with the loads of this path still in flight
            continue
        # ---- rule 4: scalar registers written by the vector unit, and the inline-assembly memory instructions that read them
        op0 = t.split(None, 1)
        if in_asm and op0[0].startswith(('buffer_', 'global_', 'ds_', 'scratch_')) and len(op0) > 1:
            for r in sregs_of(op0[1].split(';')[0]):
                if r in fresh and fresh[r][0] < VALU_SGPR_WAIT_STATES:
                    findings.append((kernel, fresh[r][1], fresh[r][2], no, t + '   ; reads s%d %d wait state(s) after the vector unit wrote it' % (r, fresh[r][0])))
        step = 1
        mnop = re.match(r's_nop\s+(\d+)', t)
        if mnop:
            step = int(mnop.group(1)) + 1
        fresh = {r: (a + step, l, x) for r, (a, l, x) in fresh.items() if a + step < 16}
        if op0[0].startswith('v_') and len(op0) > 1:
            first = op0[1].split(',')[0].strip()
            if first.startswith('s') and not first.startswith('src'):
                for r in sregs_of(first):
                    fresh[r] = (0, no, t)
        if t.startswith('s_endpgm'):
            kernel = None
            continue
        if t.startswith('s_waitcnt'):
            for which, n in WAIT.findall(t):
                q = 'vm' if which == 'vmcnt' else 'lgkm'
                n = int(n)
                pend[q] = pend[q][len(pend[q]) - n:] if n else []
            continue
        if t.startswith('s_cbranch'):                     # the fall-through path goes on with the same loads in flight
            continue
        if t.startswith(('s_branch', 's_setpc', 's_swappc')):   # what follows is reached from elsewhere: unknown state
            pend = {'vm': [], 'lgkm': []}
            continue
        m = LOAD.match(t)
        if m and not t.split()[-1] == 'lds':
            dest, rest = regs_of(m.group(2)), m.group(3)
            for regs, lno, ltxt in all_pending():          # the address operands of this load are sources too
                if regs & regs_of(rest.split(';')[0]):
                    findings.append((kernel, lno, ltxt, no, t))
            q = 'lgkm' if t.startswith('ds_') else 'vm'
            # a later load of the same kind that re-targets the registers replaces the entry (in-order return: the later one wins);
            # one of the OTHER kind racing an in-flight load is a finding
            other = 'vm' if q == 'lgkm' else 'lgkm'
            for regs, lno, ltxt in pend[other]:
                if regs & dest:
                    findings.append((kernel, lno, ltxt, no, t + '   ; WRITES a register in flight'))
            pend[q] = [(r, l, x) for r, l, x in pend[q] if not (r & dest)]
            if in_asm:
                pend[q].append((dest, no, t))
                n_loads += 1
            continue
        if not all_pending():
            continue
        ops = t.split(None, 1)
        if len(ops) < 2 or ops[0].startswith('s_'):
            continue
        operands = ops[1].split(';')[0]
        parts = [p.strip() for p in operands.split(',')]
        is_store = ops[0].startswith(('ds_write', 'buffer_store', 'global_store', 'scratch_store', 'buffer_load'))
        srcs = pk_sources(parts) if ops[0].startswith('v_pk_') else regs_of(','.join(parts if is_store else parts[1:]))
        dsts = set() if is_store else regs_of(parts[0])
        for regs, lno, ltxt in all_pending():
            if regs & srcs:
                findings.append((kernel, lno, ltxt, no, t))
            elif regs & dsts:
                # overwriting a register whose load is still in flight: the landing data would clobber the new value
                findings.append((kernel, lno, ltxt, no, t + '   ; WRITES a register in flight'))
    return kernels, n_loads, findings


def m0_findings(asm):
    """attention_tile.hip sets m0 (the LDS address of an LDS-DMA request) inside inline assembly without declaring the clobber
    (hipcc rejects m0 in a clobber list as reserved): sound as long as nothing else in the unit uses m0 -- every mention of it
    in the ISA must be an `s_mov_b32 m0` of ours, followed by s_nop and the buffer_load ... lds it belongs to."""
    lines = [l.strip() for l in asm.split('\n')]
    uses = [i for i, l in enumerate(lines) if re.search(r'\bm0\b', l) and not l.startswith(';')]
    bad = []
    for i in uses:
        ok = (lines[i].startswith('s_mov_b32 m0, s') and i + 2 < len(lines) and lines[i + 1].startswith('s_nop') and
              lines[i + 2].startswith('buffer_load_dwordx4') and lines[i + 2].endswith('lds'))
        if not ok:
            bad.append((i + 1, lines[i]))
    n_dma = sum(l.endswith(' lds') and l.startswith('buffer_load') for l in lines)
    if n_dma != len(uses):
        bad.append((0, '%d LDS-DMA loads for %d m0 writes' % (n_dma, len(uses))))
    return len(uses), bad


# translation unit -> (kernel-name substrings to walk, minimum kernels, minimum inline-assembly loads, m0 rule)
GUARDED = {'chain.hip': (('chain',), 12, 500, False), 'slab.hip': (('slab',), 1, 500, False),
           'attention_tile.hip': (('attn_tile',), 2, 3, True)}


def guard_unit(path, flags=()):
    """The rules above over one guarded translation unit as hipcc compiles it NOW -> list of problem strings (empty = sound)."""
    unit = os.path.basename(path)
    wanted, min_kernels, min_loads, owns_m0 = GUARDED[unit]
    asm = device_asm(path, flags)
    kernels, loads, findings = check(asm, wanted)
    out = ['%s: line %d `%s` <- line %d `%s`' % (k[:80], lno, ltxt, no, t) for k, lno, ltxt, no, t in findings]
    if kernels < min_kernels or loads <= min_loads:
        out.append('%s: only %d kernels / %d inline-assembly loads seen: the checker no longer recognises the unit' % (unit, kernels, loads))
    if owns_m0:
        n, bad = m0_findings(asm)
        out += ['%s: m0 at line %d: %s' % (unit, no, t) for no, t in bad]
        if n <= 20:
            out.append('%s: only %d m0 writes seen' % (unit, n))
    return out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else 'lamp_amd/csrc/chain.hip'
    wanted = tuple(sys.argv[2:]) or ('chain', 'slab')
    kernels, n_loads, findings = check(device_asm(path), wanted)
    print('%d kernels, %d inline-assembly loads checked, %d findings' % (kernels, n_loads, len(findings)))
    for k, lno, ltxt, no, t in findings[:40]:
        print('  %s\n    line %d: %s\n    line %d: %s' % (k[:90], lno, ltxt, no, t))
    sys.exit(1 if findings else 0)


if __name__ == '__main__':
    main()
