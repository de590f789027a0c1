"""Build-time guards of the hand-scheduled kernels (CPU only: hipcc cross-compiles gfx950 without a GPU).

chain.hip issues loads from inline assembly with hand-counted waits; the compiler does not know those registers are in
flight.  Two failure modes have produced silently wrong results in this repository's history (profiles/r04_rejected_experiments.txt
#4): spilled / AGPR-parked W-stream registers overwritten by loads still in flight, and register copies placed
right behind an untracked load.  Neither is visible to a parity test that happens to pass; both are visible in the compiler's
own resource report and in the ISA:
  * every kernel lamp_forward can reach has ScratchSize 0 (two known, harmless exceptions are pinned by name and size), and the
    chain kernels use no AGPRs (lamp_amd.build keeps hipcc's -Rpass-analysis=kernel-resource-usage report beside each object);
  * tools/check_untracked_loads.py finds no instruction that touches the destination of an inline-assembly load before a wait.
"""
import os
import re
import sys

import pytest

from conftest import ROOT

from lamp_amd import build as B

sys.path.insert(0, os.path.join(ROOT, 'tools'))
from lamp_amd import isa_guard as CUL  # noqa: E402   (tools/check_untracked_loads.py is its command line)
import count_loop_valu as CLV  # noqa: E402

# kernels allowed to spill: (translation unit, name) -> max bytes per lane.  Neither uses inline-assembly loads.
KNOWN_SCRATCH = {
    ('attention.hip', 'lamp::attn_kernel<128, 2, 0, 1>'): 16,        # u8-mask, two key shares: only d_v = 4 (mod 8) at <= 128 queries gets here
                                                                      # (the unsplit variant behind a bare lamp_sdpa_fwd no longer spills)
    ('backward.hip', 'lamp::layernorm_bwd_kernel<16, true, 3>'): 160,  # training only
}
FORWARD_UNITS = ['gemm.hip', 'chain.hip', 'attention.hip', 'attention_tile.hip', 'attention_small.hip', 'attention_general.hip', 'pointwise.hip']


@pytest.mark.parametrize('tuning', [False, True])
def test_no_kernel_spills_and_the_chain_uses_no_agprs(tuning):
    seen = 0
    for unit in B.SOURCES + (B.TUNING_ONLY if tuning else []):
        if unit == 'api.hip' or (tuning and unit not in B.TUNING_SOURCES and unit not in B.TUNING_ONLY):
            continue
        res = B.kernel_resources(unit, tuning)
        assert res, unit
        for name, r in res.items():
            seen += 1
            allowed = KNOWN_SCRATCH.get((unit, name), 0)
            if tuning and 'gemm_pair_kernel' in name:
                allowed = 96   # (84 since the straight-line epilogue) the rejected counter-chained FFN pair (profiles/r04_rejected_experiments.txt #11): tuning build only
            assert r.get('scratch', 0) <= allowed, (unit, name, r)
            if ('chain' in name or 'slab' in name or 'attn_tile' in name) and 'kernel' in name:
                # the W stream's registers must stay where the in-flight loads will write them
                assert r.get('agpr', 0) == 0 and r.get('scratch', 0) == 0, (name, r)
            if 'attn_tile_kernel' in name:   # two workgroups per CU: the partner hides barrier and LDS latencies
                assert r.get('occupancy', 0) >= 2 and r.get('vgpr', 999) <= 256, (name, r)
    assert seen > 100
    # the production chain kernels exist in the product build (what lamp_forward launches at batch 32)
    prod = B.kernel_resources('chain.hip', False)
    for want in ('lamp::chain_kernel<2, 16, 32, 2, 1>', 'lamp::chain_packed_kernel<2, 16, 32, 4>', 'lamp::chain_rows4_kernel<2, 3>'):
        assert want in prod, sorted(prod)
        assert prod[want]['occupancy'] >= (2 if 'rows4' in want else 4), prod[want]


def test_the_build_runs_the_guards_and_stamps_its_toolchain(tmp_path, monkeypatch):
    """lamp_amd.build.build() itself refuses to link a library whose hand-scheduled kernels fail the ISA / resource
    rules (tools/check_untracked_loads.py is only the command line of the same checker), and records the compiler."""
    B.build()
    assert B.verify() == []                       # stamped verdicts of the build in the tree
    assert B.built_toolchain() == B.toolchain() and 'clang version' in B.toolchain()
    # a finding fails the build before anything is linked (the guard's verdict is forced; nothing is compiled here)
    monkeypatch.setattr(B, 'verify', lambda verbose=False: ['chain.hip: forced finding'])
    monkeypatch.setattr(B, 'needs_build', lambda: True)
    monkeypatch.setattr(B, 'LIB', str(tmp_path / 'liblamp_hip.so'))
    monkeypatch.setattr(B, 'LIB_TUNING', str(tmp_path / 'liblamp_hip_tuning.so'))
    (tmp_path / 'liblamp_hip.so').write_bytes(b'stale')
    with pytest.raises(RuntimeError, match='ISA guard'):
        B.build()
    assert not (tmp_path / 'liblamp_hip.so').exists()
    # the resource rule: AGPRs or scratch in a guarded kernel
    monkeypatch.setattr(B, 'kernel_resources', lambda s, t: {'lamp::chain_rows4_kernel<2, 3>': {'agpr': 4, 'scratch': 0},
                                                              'lamp::gemm_nt_kernel<1>': {'agpr': 64, 'scratch': 0}})
    assert len(B._resource_problems('chain.hip', False)) == 1


def test_checker_sees_a_copy_behind_an_untracked_load():
    """The exact shape of round 5's bug (LayerNorm operands read under `if`): the merge's copies right behind the read."""
    bad = '''
_ZN4lamp12chain_kernelILi2ELi16ELi32ELi2ELi1EEEvNS_11ChainParamsE:
	s_cbranch_vccnz .LBB0_2
	;;#ASMSTART
	ds_read_b128 v[28:31], v0
	;;#ASMEND
	s_nop 0
	v_mov_b32_e32 v38, v29
.LBB0_2:
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_add_f32_e32 v1, v28, v2
	s_endpgm
'''
    kernels, loads, findings = CUL.check(bad)
    assert kernels == 1 and loads == 1 and len(findings) == 1 and 'v_mov_b32' in findings[0][4]
    good = bad.replace('\tv_mov_b32_e32 v38, v29\n', '')
    assert CUL.check(good)[2] == []
    # a register in flight overwritten before its wait
    clobber = bad.replace('v_mov_b32_e32 v38, v29', 'v_mov_b32_e32 v30, v2')
    assert len(CUL.check(clobber)[2]) == 1


def test_checker_sees_a_scalar_written_by_the_vector_unit_in_front_of_a_load():
    """Round 5, slab.hip: the compiler reloaded the stream's scalar offset with v_readlane right in front of an inline-assembly
    load (no hazard padding for instructions it cannot see): the load went out with the old offset."""
    bad = '''
_ZN4lamp16slab_gemm_kernelILi10EEEvNS_10SlabParamsE:
	v_readlane_b32 s8, v209, 5
	;;#ASMSTART
	buffer_load_dwordx4 v[46:49], v85, s[16:19], s8 offen offset:0
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	s_endpgm
'''
    assert len(CUL.check(bad)[2]) == 1
    assert CUL.check(bad.replace('\t;;#ASMSTART\n\tbuffer_load', '\ts_nop 4\n\t;;#ASMSTART\n\tbuffer_load'))[2] == []
    assert len(CUL.check(bad.replace('v_readlane_b32 s8, v209, 5', 'v_readfirstlane_b32 s18, v86'))[2]) == 1   # a descriptor word


@pytest.mark.parametrize('unit,flags,n_kernels,n_loads', [('chain.hip', (), 12, 500), ('chain.hip', ('-DLAMP_TUNING',), 12, 500),
                                                          ('experiments/slab.hip', ('-DLAMP_TUNING',), 1, 500), ('attention_tile.hip', (), 2, 3)])
def test_no_instruction_touches_an_inline_assembly_load_before_a_wait(unit, flags, n_kernels, n_loads):
    """... nor does one overwrite a register in flight, nor does an inline-assembly memory instruction read a scalar the vector
    unit wrote fewer than five wait states earlier (tools/check_untracked_loads.py: the three rules)."""
    asm = CUL.device_asm(os.path.join(ROOT, 'lamp_amd', 'csrc', unit), flags)
    kernels, loads, findings = CUL.check(asm, ('chain', 'slab', 'attn_tile'))
    assert kernels >= n_kernels and loads > n_loads
    assert findings == [], findings[:5]


def test_checker_reads_the_half_selects_of_packed_instructions():
    """`v_pk_mul_f32 v[14:15], v[92:93], v[14:15] op_sel_hi:[0,1]` broadcasts v92: it does not read v93 (where hipcc likes to
    keep an in-flight mask word); with the default selects it does."""
    tmpl = '''_ZN4lamp12_GLOBAL__N_116attn_tile_kernelILi3EEEvNS_10AttnParamsE:
	;;#ASMSTART
	buffer_load_dword v93, v64, s[12:15], 0 offen
	;;#ASMEND
	v_pk_mul_f32 v[14:15], v[92:93], v[14:15]%s
	s_waitcnt vmcnt(0)
	s_endpgm
'''
    assert CUL.check(tmpl % ' op_sel_hi:[0,1]', ('attn_tile',))[2] == []
    assert len(CUL.check(tmpl % '', ('attn_tile',))[2]) == 1


def test_the_tile_attention_kernel_owns_m0():
    """attention_tile.hip sets m0 (the LDS address of an LDS-DMA request) inside inline assembly without declaring the clobber
    (hipcc rejects m0 in a clobber list as reserved): sound as long as nothing else in the kernel uses m0 -- every mention of
    it in the ISA must be an `s_mov_b32 m0` of ours, followed by s_nop and the buffer_load ... lds it belongs to."""
    asm = CUL.device_asm(os.path.join(ROOT, 'lamp_amd', 'csrc', 'attention_tile.hip'))
    n, bad = CUL.m0_findings(asm)
    assert n > 20 and bad == [], bad[:5]
    # ... and the rule sees an m0 write that is not followed by its LDS-DMA load
    assert CUL.m0_findings('s_mov_b32 m0, s5\ns_nop 0\nv_mov_b32_e32 v1, v2\n')[1]


# ---- vector instructions in the MFMA loops: matrix-pipe time on gfx950 (profiles/r05_mfma_chain.txt) ----
def _loops(unit, want, flags=()):
    asm = CLV.device_asm(os.path.join(ROOT, 'lamp_amd', 'csrc', unit), flags)
    found = {n: CLV.mfma_loops(l) for n, l in CLV.kernels(asm).items() if want in n}
    assert found, (unit, want)
    return asm, found


def test_gemm_main_loops_carry_no_vector_instructions():
    """Forward tile GEMMs (LDS staging through registers / LDS-DMA: offsets advance in scalar registers) and the general GEMM
    (loop-invariant per-lane offsets, the k-tile's base in the scalar offset): the innermost loop of every product-path
    instantiation holds MFMAs, LDS / memory instructions and scalar bookkeeping only.  (Before round 5 gemm_gen carried 36-66
    vector instructions per two k-tiles: +17...+63 % on the MFMAs' time.)  The TAIL instantiations (last template argument true:
    a row operand whose K leaves a partial k-tile) also hold that one tile's per-element bounds -- statically inside the loop,
    behind a scalar branch that only the last k-tile takes."""
    for want, n_plain, n_tail in (('gemm_gen_kernel<', 8, 6), ('gemm_group_kernel<', 4, 3)):
        _, gen = _loops('gemm_gen.hip', want)
        plain = {n: l for n, l in gen.items() if n.split('>')[0].endswith('false')}
        tail = {n: l for n, l in gen.items() if n.split('>')[0].endswith('true')}
        assert len(plain) == n_plain and len(tail) == n_tail, (want, sorted(gen))
        for name, loops in plain.items():
            inner = loops[0][2]
            assert inner['valu'] + inner['trans'] <= 6, (name, dict(inner))     # one instantiation keeps a short waterfall
        assert sum(l[0][2]['valu'] == 0 for l in plain.values()) >= n_plain - 1
        for name, loops in tail.items():
            inner = loops[0][2]
            assert 0 < inner['valu'] <= 44 and inner['trans'] == 0, (name, dict(inner))
    _, nt = _loops('gemm.hip', 'gemm_nt_kernel<')
    for name, loops in nt.items():
        if ', 16, 2, 2, false, ' in name:   # the 16-deep tiles of the forward (64x64x16, 128x64x16), K a multiple of the tile depth
            assert loops[0][2]['valu'] == 0 and loops[0][2]['mfma'] >= 16, (name, dict(loops[0][2]))


def test_attention_key_loops_stay_within_their_vector_instruction_budget():
    """attention_tile.hip: two tile steps per loop trip, 256 MFMAs; the executed path carries ~79 vector instructions + 16 v_exp
    per tile with the bit mask (the static count below also holds the rescale branch's 32 v_pk_mul per step).  attn16_kernel:
    <= 40 per 64-MFMA tile (91 before round 5)."""
    _, tile = _loops('attention_tile.hip', 'attn_tile_kernel<')
    for name, loops in tile.items():
        main = [c for a, b, c in loops if c['mfma'] in (256, 384)]   # without a mask: the unbiased QK^T and the last tile's biased one
        assert main, name
        c = min(main, key=lambda c: c['valu'])
        assert c['valu'] - c['v_pk_mul_f32'] <= 180 and c['trans'] <= 36, (name, dict(c))
        assert c['v_cndmask_b32_e32'] + c['v_cndmask_b32_e64'] <= 8, (name, dict(c))    # the mask is NOT applied by selects
    _, small = _loops('attention_small.hip', 'attn16_kernel<128, 1, 4, 0, 3>')
    for name, loops in small.items():
        c = [c for a, b, c in loops if c['mfma'] == 64][0]
        assert c['valu'] + c['trans'] <= 40, (name, dict(c))
