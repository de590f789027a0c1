"""CPU-side checks of the product: the C-ABI library loads and exports every declared symbol, the
nn.Module surface mirrors the reference (constructor kwargs, state_dict keys/shapes, mask
construction), and the product refuses -- loudly -- to run anywhere but on a HIP device."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

from conftest import ROOT, golden_names, load_golden

import lamp_amd
from lamp_amd import _native as N
from lamp_amd.Models import LAMP


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'lamp_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(lamp_[a-z0-9_]+)\s*\(', text)))


def test_library_loads_and_exports_every_declared_symbol():
    assert os.path.exists(N.LIB_PATH), 'run python -m lamp_amd.build'
    lib = ctypes.CDLL(N.LIB_PATH)
    names = header_functions()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(N.PROTOTYPES) == names  # the ctypes binding covers the whole header
    assert N.lib().lamp_version() == N.ABI_VERSION == 5
    # the product library exports no tuning / debug hook (those live in the -DLAMP_TUNING build only)
    exported = subprocess.run(['nm', '-D', '--defined-only', N.LIB_PATH], capture_output=True, text=True).stdout
    assert 'lamp_debug' not in exported and 'lamp_set_forward_streams' not in exported
    # ... and no function the header does not declare: the library is built with -fvisibility=hidden (what remains beside
    # the entry points are the HIP runtime's kernel handles and fat-binary markers: data objects, not functions)
    assert sorted(re.findall(r' [TtWw] (\S+)', exported)) == names
    tuned = subprocess.run(['nm', '-D', '--defined-only', N.TUNING_LIB_PATH], capture_output=True, text=True).stdout
    assert 'lamp_debug_force_gemm_tile' in tuned and 'lamp_debug_force_attn' in tuned
    # both libraries resolve every symbol at load time (a kernel whose launch stub the host pass dropped shows up here)
    for path in (N.LIB_PATH, N.TUNING_LIB_PATH):
        ctypes.CDLL(path, mode=getattr(os, 'RTLD_NOW', 2))
    undefined = subprocess.run(['nm', '-D', '--undefined-only', N.TUNING_LIB_PATH], capture_output=True, text=True).stdout
    assert 'gemm_nt_kernel' not in undefined and 'attn' not in undefined.lower().replace('pthread_attr', '')
    assert b'workspace' in N.lib().lamp_strerror(-3)


def test_struct_layouts_match_header_sizes():
    # sizes the C compiler produces for the same declarations (x86-64 SysV)
    assert ctypes.sizeof(N.Mask) == 56
    assert ctypes.sizeof(N.AttnLayout) == 96
    assert ctypes.sizeof(N.MhaWeights) == 56
    assert ctypes.sizeof(N.FfnWeights) == 48
    assert ctypes.sizeof(N.EncLayer) == 104
    assert ctypes.sizeof(N.DecLayer) == 208
    assert ctypes.sizeof(N.Model) == 152
    assert ctypes.sizeof(N.ChainPack) == 48
    assert ctypes.sizeof(N.Aux) == 40
    assert ctypes.sizeof(N.GemmDesc) == 160


def build_from_fixture(name):
    d, sd = load_golden(name)
    L, dm = sd['decoder.tgt_word_emb.weight'].shape
    V = sd['encoder.src_word_emb.weight'].size(0)
    h = d['n_head']
    n_enc = len({k.split('.')[2] for k in sd if k.startswith('encoder.layer_stack.')})
    n_dec = len({k.split('.')[2] for k in sd if k.startswith('decoder.layer_stack.')})
    pos = 'encoder.position_enc.weight' in sd
    n_max = sd['encoder.position_enc.weight'].size(0) - 1 if pos else 12
    dff = sd['encoder.layer_stack.0.pos_ffn.w_1.weight'].size(0)
    adj = d.get('label_adj_matrix')
    m = LAMP(V, L, n_max, L, n_layers_enc=n_enc, n_layers_dec=n_dec, n_head=h, n_head2=h, d_word_vec=dm,
             d_model=dm, d_inner_hid=dff, d_k=dm // h, d_v=dm // h, dropout=0.1, dec_dropout=0.1,
             dec_dropout2=False, proj_share_weight=True, embs_share_weight=True, encoder='graph',
             decoder='graph', enc_transform='', onehot=False, no_enc_pos_embedding=not pos,
             no_dec_self_att='decoder.layer_stack.0.slf_attn.w_qs.weight' not in sd, loss='ce',
             label_adj_matrix=adj.clone() if adj is not None else None, label_mask=d['label_mask'],
             matching_mlp=False, graph_conv=False, attn_type='softmax', int_preds=False)
    return m, d, sd


@pytest.mark.parametrize('name', golden_names('model_'))
def test_state_dict_layout_equals_reference(name):
    m, d, sd = build_from_fixture(name)
    own = m.state_dict()
    assert sorted(own) == sorted(sd)
    for k in sd:
        assert tuple(own[k].shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd)  # strict
    # the accidental non-tie of the read-out is reproduced
    assert m.tgt_word_proj.weight is m.decoder.tgt_word_emb.weight
    assert m.tgt_word_proj.linear.weight is not m.decoder.tgt_word_emb.weight
    # label mask: same tensor as the reference built, in the reference's own format
    if 'ref_label_mask' in d:
        assert torch.equal(m.decoder.label_mask, d['ref_label_mask'])
        L = own['decoder.tgt_word_emb.weight'].size(0)
        assert torch.equal(m.decoder.label_mask_u8, (d['ref_label_mask'].view(L, L) != 0).to(torch.uint8))
    elif d['label_mask'] == 'none':
        assert m.decoder.label_mask is None and m.decoder.label_mask_u8 is None
    assert not any(k.startswith('decoder.label_') for k in own)
    # the sinusoid table is frozen out of the optimiser's parameter list only
    n_train = sum(1 for _ in m.get_trainable_parameters())
    assert n_train == len(list(m.parameters())) - (1 if 'encoder.position_enc.weight' in sd else 0)


def test_position_table_bitwise():
    _, sd = load_golden('model_prior_pos1_h4')
    ref = sd['encoder.position_enc.weight']
    assert torch.equal(lamp_amd.utils.position_encoding_init(ref.size(0), ref.size(1)), ref)


def test_padding_mask_helper_and_swap():
    seq = torch.tensor([[5, 6, 0], [7, 0, 0]])
    m = lamp_amd.utils.get_attn_padding_mask(seq, seq)
    assert m.shape == (2, 3, 3) and m[0, :, 2].all() and not m[0, :, :2].any() and m[1, :, 1:].all()
    t = torch.tensor([[0., 2.], [3., 0.]])
    assert torch.equal(lamp_amd.utils.swap_0_1(t, 1, 0), torch.tensor([[1., 0.], [0., 1.]]))


def test_active_tile_list_of_a_clustered_label_graph():
    L = 100  # 4 query blocks x 4 key tiles of 32
    blocked = torch.ones(L, L, dtype=torch.uint8)
    blocked[:40, :40] = 0          # cluster A: labels 0..39  -> tiles 0,1 for query blocks 0,1
    blocked[40:, 40:] = 0          # cluster B: labels 40..99 -> tiles 1,2,3 for query blocks 1,2,3
    blocked[5, 99] = 0             # one stray edge: query block 0 also needs tile 3
    tl = N.active_tile_list(blocked)
    assert tl.shape == (4, 5) and tl.dtype == torch.int32
    assert tl[0].tolist() == [3, 0, 1, 3, 0]
    assert tl[1].tolist() == [4, 0, 1, 2, 3]
    assert tl[2].tolist() == [3, 1, 2, 3, 0]
    assert tl[3].tolist() == [3, 1, 2, 3, 0]


def test_tile_list_hint_is_dropped_on_graphs_it_cannot_help():
    """BASELINE configs[4]'s Bernoulli(0.05) prior graph has an edge in every 32x32 tile; walking the active-tile
    list there only costs.  GraphDecoder keeps the hint for block-structured graphs and drops it from 90 % active tiles on."""
    from lamp_amd.Decoders import GraphDecoder
    from lamp_amd import synthetic as S
    L = 256
    dense = GraphDecoder(L, L, n_layers=1, n_head=2, n_head2=2, d_k=8, d_v=8, d_word_vec=16, d_model=16, d_inner_hid=32,
                         label_adj_matrix=S.make_adjacency(L, 0.05, 0), label_mask='prior')
    assert dense.label_tile_density == 1.0 and dense.label_tiles is None and dense.label_mask_bits is not None
    adj = torch.eye(L)
    for c in range(0, L, 64):
        adj[c:c + 64, c:c + 64] = 1
    clustered = GraphDecoder(L, L, n_layers=1, n_head=2, n_head2=2, d_k=8, d_v=8, d_word_vec=16, d_model=16, d_inner_hid=32,
                             label_adj_matrix=adj, label_mask='prior')
    assert clustered.label_tile_density == 0.25 and clustered.label_tiles is not None
    assert GraphDecoder(L, L, n_layers=1, n_head=2, n_head2=2, d_k=8, d_v=8, d_word_vec=16, d_model=16, d_inner_hid=32,
                        label_mask='none').label_tile_density is None


def test_bench_counts_executed_flops_not_f_live():
    """bench.py's utilisation figures are on EXECUTED FLOPs (no fraction above 1): F_live minus the weights-only
    hoist / fold and, where the label self-attention runs the pair kernel, with 4 L^2 d replaced by 4 nnz d."""
    import bench

    class _Dec(object):
        label_rows_sparse, label_allowed_pairs = False, 0

    class _M(object):
        decoder = _Dec()
    w = dict(bench.WORKLOADS['reuters'])
    fx, sub = bench.executed_flops(w, _M())
    assert set(sub) == {'hoisted_dec0_query', 'folded_enc0_w1'}
    assert sub['hoisted_dec0_query'] == 2 * 90 * 512 * 512 and sub['folded_enc0_w1'] == 2 * 302 * 512 * 512
    assert fx == bench.f_live(w) - sum(sub.values()) and bench.f_live(w) == 2354997248
    w5 = dict(bench.WORKLOADS['synthetic4096'])
    _Dec.label_rows_sparse, _Dec.label_allowed_pairs = True, int(0.0978 * 4096 * 4096)
    fx5, sub5 = bench.executed_flops(w5, _M())
    skipped = sub5['blocked_label_pairs_skipped']
    assert abs(skipped - 2 * 4 * 1024 * (4096 * 4096 - _Dec.label_allowed_pairs)) < 1 and 0.25 < skipped / bench.f_live(w5) < 0.35
    assert 0 < fx5 < bench.f_live(w5)


def test_pack_mask_bits_layout():
    g = torch.Generator().manual_seed(0)
    blocked = (torch.rand(70, 100, generator=g) < 0.5).to(torch.uint8)
    blocked[3, 31] = 1
    blocked[3, 63] = 1
    bits = N.pack_mask_bits(blocked)
    assert bits.shape == (70, 4) and bits.dtype == torch.int32
    w = bits.to(torch.int64) & 0xFFFFFFFF
    for q, k in ((0, 0), (3, 31), (3, 63), (69, 99), (10, 64), (5, 33)):
        assert ((w[q, k >> 5] >> (k & 31)) & 1).item() == blocked[q, k].item()
    assert (w[:, 3] >> 4).max().item() == 0        # bits past lk = 100 (word 3 holds keys 96..99) are zero


def test_product_has_no_cpu_path():
    m, d, sd = build_from_fixture('model_prior_pos1_h4')
    m.load_state_dict(sd)
    m.eval()
    with pytest.raises(RuntimeError, match='HIP device only'):
        m((d['src_seq'], d['src_pos']), None, None, None)
    with pytest.raises(RuntimeError, match='HIP device only'):
        m.encoder.layer_stack[0].pos_ffn(torch.zeros(1, 2, 64))
    m.train()  # the training path is HIP-only as well
    with pytest.raises(RuntimeError, match='HIP device only'):
        m((d['src_seq'], d['src_pos']), None, None, None)
    with pytest.raises(RuntimeError, match='HIP device only'):
        m.encoder.layer_stack[0].pos_ffn(torch.zeros(1, 2, 64))
    with pytest.raises(RuntimeError, match='HIP device only'):   # the bare wrappers record autograd on the device too
        m.tgt_word_proj(torch.zeros(1, 2, 64))
    with pytest.raises(RuntimeError, match='HIP device only'):
        m.decoder.layer_stack[0].enc_attn.attention(torch.zeros(1, 2, 16), torch.zeros(1, 3, 16), torch.zeros(1, 3, 16))


def test_out_of_scope_branches_raise_at_construction():
    """What stays outside: the genomics one-hot / Conv1d input branch, enc_transform='max' (a NameError in the
    reference itself) and decoder names the reference's own LAMP rejects."""
    from lamp_amd.Encoders import GraphEncoder, pool_encoder_output
    with pytest.raises(NotImplementedError):
        GraphEncoder(10, 4, n_layers=1, n_head=1, d_k=8, d_v=8, d_word_vec=8, d_model=8, d_inner_hid=16, onehot=True)
    with pytest.raises(NotImplementedError):
        pool_encoder_output(torch.zeros(2, 3, 4), torch.ones(2, 3, dtype=torch.long), 'max')
    with pytest.raises(NotImplementedError):
        LAMP(10, 5, 4, 5, encoder='graph', decoder='sa_m', label_mask='none')


def test_argument_errors_come_back_as_status_codes_without_a_gpu():
    lib = N.lib()
    # validation happens before any launch, so these are safe on a GPU-less host
    assert lib.lamp_linear_fwd(None, 4, 8, 8, None, 4, 8, None, None, 0, 0, None, 4, None) == -5
    assert lib.lamp_linear_fwd(16, 4, 6, 6, 16, 4, 6, None, None, 0, 0, 16, 4, None) == -2   # K % 4
    assert lib.lamp_linear_fwd(16, 0, 8, 8, 16, 4, 8, None, None, 0, 0, 16, 4, None) == -1
    assert lib.lamp_layernorm_fwd(16, 4, 6, 16, 16, 1e-5, 16, None) == -4
    lay = N.AttnLayout(*([4] * 12))
    assert lib.lamp_sdpa_fwd(16, 16, 16, 16, None, 1, 1, 4, 4, 130, 130, 1.0, None, ctypes.byref(lay), None) == -4
    # wide heads need the map buffer (or scratch) for their scores: reported before any launch
    assert lib.lamp_sdpa_fwd(16, 16, 16, 16, None, 1, 1, 4, 4, 256, 256, 1.0, None, ctypes.byref(lay), None) == -3
    assert lib.lamp_forward_workspace_bytes(None, 1, 4, 0) == 0


def test_dropin_package_aliases_reference_import_paths():
    import subprocess, sys
    code = ("import lamp.Constants as C, lamp.Models, lamp.Translator, lamp.Beam, lamp.Layers, lamp.SubLayers;"
            "from lamp.Models import LAMP; from lamp.Translator import translate;"
            "from lamp.Attention import ScaledDotProductAttention as S; import lamp_amd;"
            "assert LAMP is lamp_amd.Models.LAMP and S is lamp_amd.SubLayers.ScaledDotProductAttention;"
            "assert C.PAD == 0 and C.EOS == 3; print('ok')")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, 'dropin'), ROOT]))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, cwd='/tmp')
    assert r.returncode == 0 and 'ok' in r.stdout, r.stderr[-2000:]


def test_bench_refuses_to_run_without_a_gpu_and_its_spawner_does_not_hang():
    """bench.py has no CPU path; started with --gpus 2 on a GPU-less host the self-spawned ranks fail and the parent
    returns their error instead of waiting for a rendezvous (SURVEY.md 8e entry point)."""
    import sys
    if torch.cuda.is_available():
        pytest.skip('GPU box: covered by tests/test_gpu_multi.py')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    for extra in ([], ['--gpus', '2']):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1'] + extra, capture_output=True,
                           text=True, env=env, timeout=300)
        assert r.returncode != 0 and 'needs an MI355X' in r.stderr and r.stdout.strip() == ''


def test_bench_kernel_trace_parsing_and_roofline_split(monkeypatch, tmp_path):
    """bench.py's kernel-only figures come from a rocprofv3 --kernel-trace --stats sub-run: the reading of its
    p_kernel_stats.csv (per-forward normalisation by the gather launches, GEMM class = gemm_nt_kernel + the chain launch in any of its forms) and
    the roofline split are host logic -- checked here against a hand-written trace, with the profiler call replaced."""
    import argparse
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    rows = [('void lamp::chain_rows4_kernel<2, 3>(lamp::ChainParams)', 40, 60000.0),
            ('void lamp::gemm_nt_kernel<64, 64, 16, 2, 2, false, 16, true, true, 0>(lamp::GemmParams, int)', 40, 45000.0),
            ('void lamp::gemm_nt_kernel<128, 64, 16, 2, 2, false, 16, false, true, 0>(lamp::GemmParams, int)', 10, 158000.0),
            ('void lamp::attn16_kernel<128, 1, 4, 0, 3>(lamp::AttnParams)', 20, 25000.0),
            ('lamp::embed_plan_kernel(long const*, long const*)', 10, 12000.0),
            ('__amd_rocclr_copyBuffer', 3, 4000.0)]

    def fake_run(cmd, **kw):
        assert cmd[1:3] == ['--kernel-trace', '--stats'] and '--pmc' not in cmd and '--no-kernel-trace' in cmd
        out = cmd[cmd.index('-d') + 1]
        with open(os.path.join(out, 'p_kernel_stats.csv'), 'w') as f:
            f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n')
            for name, calls, avg in rows:
                f.write('"%s",%d,%d,%f,1.0,1,1,0.0\n' % (name, calls, int(calls * avg), avg))
        return subprocess.CompletedProcess(cmd, 0, b'', b'')

    monkeypatch.setattr(bench.subprocess, 'run', fake_run)
    monkeypatch.setattr('shutil.which', lambda exe: '/opt/rocm/bin/rocprofv3')
    for k in [k for k in os.environ if k.startswith(('ROCPROF', 'ROCP_TOOL'))]:
        monkeypatch.delenv(k)
    args = argparse.Namespace(workload='reuters', batch=32)
    live = bench.live_kernel_trace(args)
    assert live['forwards_traced'] == 10
    assert abs(live['gemm_class_us_per_forward'] - (4 * 60.0 + 4 * 45.0 + 158.0)) < 1e-9
    assert abs(live['all_kernels_us_per_forward'] - (578.0 + 2 * 25.0 + 12.0)) < 1e-9   # the runtime's copy kernel is not ours
    assert list(live['by_kernel'])[0].startswith('chain_rows4_kernel') and bench.is_chain_kernel('lamp::chain_packed_kernel<2, 16, 32, 4>') and\
        not bench.is_chain_kernel('lamp::pack_weight_kernel<1>') and live['by_kernel']['embed_plan_kernel']['launches_per_forward'] == 1.0
    # 10 steps of 69.2 GFLOP in the GEMM class, 18.1 of them in the chain launches
    prof = {'gemm': {'flops': 69.2e9 * 10, 'ms': 7.0, 'launches': 120, 'bytes': 1e9}}
    roof = bench.roofline_of(prof, 10, None, live, 18.1)
    assert abs(roof['achieved_kernel_only'] - 69.2 / 578.0 * 1e3) < 1e-9 and roof['kernel_only_source'].startswith('live:')
    sp = roof['kernel_only_split']
    assert abs(sp['chain_kernel']['tflops'] - 18.1 / 240.0 * 1e3) < 1e-9
    assert abs(sp['gemm_nt_kernel']['tflops'] - 51.1 / 338.0 * 1e3) < 1e-6
    # ---- the counter passes (roofline.traffic, roofline.attention): three sub-runs, counters never beside another trace domain
    calls = []

    def fake_pmc(cmd, **kw):
        assert cmd[1] == '--kernel-trace' and cmd[2] == '--pmc' and '--stats' not in cmd and '--sys-trace' not in cmd and '--no-pmc' in cmd
        counters = cmd[3:cmd.index('-d')]
        calls.append(counters)
        out = cmd[cmd.index('-d') + 1]
        kernels = [('void lamp::gemm_nt_kernel<64, 64, 16, 2, 2, false, 16, true, true, 0>(lamp::GemmParams, int)', 45.0, 30000.0, 20000.0),
                   ('void lamp::chain_rows4_kernel<2, 3>(lamp::ChainParams)', 50.0, 4000.0, 6000.0),
                   ('void lamp::attn16_kernel<128, 1, 4, 0, 3>(lamp::AttnParams)', 25.0, 9000.0, 1500.0)]
        with open(os.path.join(out, 'p_kernel_trace.csv'), 'w') as f:
            f.write('"Dispatch_Id","Start_Timestamp","End_Timestamp"\n')
            for i, (_, us, _, _) in enumerate(kernels * 2):
                f.write('%d,%d,%d\n' % (i, 1000000 * i, 1000000 * i + int(us * 1000)))
        with open(os.path.join(out, 'p_counter_collection.csv'), 'w') as f:
            f.write('"Dispatch_Id","Kernel_Name","Grid_Size","Workgroup_Size","Counter_Name","Counter_Value"\n')
            for i, (name, us, fetch_kib, write_kib) in enumerate(kernels * 2):
                for c in counters:
                    v = {'FETCH_SIZE': fetch_kib, 'WRITE_SIZE': write_kib, 'SQ_BUSY_CYCLES': 32 * us * 1000 * 2.0,
                         'SQ_VALU_MFMA_BUSY_CYCLES': 1024 * us * 1000 * 2.0 * 0.5}[c]
                    f.write('%d,"%s",1024,256,"%s",%f\n' % (i, name, c, v))
        return subprocess.CompletedProcess(cmd, 0, b'', b'')

    monkeypatch.setattr(bench.subprocess, 'run', fake_pmc)
    pmc = bench.live_pmc(args)
    assert calls == [['FETCH_SIZE'], ['WRITE_SIZE'], ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES']]
    g = pmc['by_kernel']['gemm_nt_kernel<64, 64, 16, 2, 2, false, 16, true, true, 0>']
    assert g['launches'] == 2 and abs(g['fetch_bytes'] - 30000.0 * 1024 * 2) < 1e-3 and abs(g['write_bytes'] - 20000.0 * 1024) < 1e-3
    assert abs(g['mfma_busy'] - 0.5) < 1e-9 and abs(g['clock_ghz_under_profiler'] - 2.0) < 1e-9
    roof = bench.roofline_of(prof, 10, 'reuters', live, 18.1, pmc, {'tflops': 60.0, 'algorithmic_gbs': 500.0})
    want = ((30000.0 * 2 + 20000.0) + (4000.0 * 2 + 6000.0)) * 1024 / 2          # GEMM class: tile kernel + chain, per launch
    assert roof['traffic_source'] == 'live' and abs(roof['traffic'] - want) < 1e-3 and roof['traffic_stale'] is False
    att = roof['attention']
    assert abs(att['hbm_gbs'] - (9000.0 * 2 + 1500.0) * 1024 / 25e-6 / 1e9) < 1e-6 and abs(att['mfma_busy'] - 0.5) < 1e-9
    assert att['hbm_peak_gbs'] == 8000.0 and att['frac_of_fp32_mfma_peak'] == 60.0 / 157.3 and att['source'].startswith('live:')
    # without the passes the committed figure is used and says so
    assert bench.roofline_of(prof, 10, 'reuters', live, 18.1, {'skipped': 'x'}).get('traffic_source') in ('committed', None)
    monkeypatch.setattr(bench.subprocess, 'run', fake_run)
    # a profiled parent never starts a nested profiler
    monkeypatch.setenv('ROCPROFILER_TOOL', '1')
    assert 'skipped' in bench.live_kernel_trace(args)


def test_intra_op_threads_follow_the_cgroup_quota_not_nproc():
    """lamp_amd/hostcpu.py: a container's CPU quota (cgroup cpu.max), not the host's core count, sizes torch's OpenMP pool
    for the evaluation epoch's host work (profiles/r06_eval_epoch_threads.txt: 20 throttled periods vs 0)."""
    from lamp_amd import hostcpu as H
    assert H.parse_cpu_max('1600000 100000') == 16.0
    assert H.parse_cpu_max('max 100000') is None
    assert H.parse_cpu_max('-1 100000') is None            # cgroup v1 spelling of "no limit"
    assert H.parse_cpu_max('garbage') is None and H.parse_cpu_max('') is None
    assert H.parse_cpu_max('50000 100000') == 0.5
    assert H.fitted_threads(16, 128) == 8                  # half the quota; the rest is for the issuing / producer / HIP threads
    assert H.fitted_threads(16, 128, share=8) == 1         # eight ranks in one cgroup
    assert H.fitted_threads(256, 4) == 4                   # never raises what the user set
    assert H.fitted_threads(1, 128) == 1
    assert H.usable_cores() >= 1
    before = torch.get_num_threads()
    n = H.fit_intra_op_threads()
    assert 1 <= n <= before and torch.get_num_threads() == n
    assert H.fit_intra_op_threads() == n                   # once per process
