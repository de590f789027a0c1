"""-m gpu: the N > 1 and threaded entry points of the HIP path on ONE box (SURVEY.md 8e, 8b "Threading").

* bench.py --gpus 2 started plainly (no torchrun): the script spawns the ranks itself; with LAMP_BENCH_BACKEND=gloo the
  two ranks share the only GPU of the test box.
* run_eval -gpus 2: the batches of a test split sharded over two ranks running the HIP forward, against one rank.
* nn.DataParallel wrapping (main.py:106-108), DataParallel-style replicas driven from two host threads, and two
  threads on two streams: bitwise equal to the plain call.
"""
import argparse
import json
import os
import subprocess
import sys
import threading

import pytest
import torch

from conftest import ROOT, load_golden
from test_gpu_parity import CONFIGS, make_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from lamp_amd import _native as N
    N.lib()
    return torch.device('cuda:0')


def _bench(args, env_extra=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True,
                       env=env, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_spawns_its_own_ranks(dev):
    common = ['--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--no-pipelined', '--no-extra-workloads', '--no-kernel-trace']
    r1, one = _bench(['--gpus', '1'] + common)
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert one['n_gpus'] == 1 and one['ranks_seen'] == [0] and len(one['per_rank']) == 1
    assert one['metric'] == 'forward samples/sec, reuters d512 2+2L 4h' and one['steps'] == 5 and one['unit'] == 'samples/s'
    r2, two = _bench(['--gpus', '2'] + common, {'LAMP_BENCH_BACKEND': 'gloo'})
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert two['n_gpus'] == 2 and two['ranks_seen'] == [0, 1] and len(two['per_rank']) == 2
    assert two['config']['launcher'] == 'self-spawned ranks' and two['physical_devices'] == 1
    # same weights on both ranks, different batches: rank 0 recomputed rank 1's samples bit for bit
    assert two['backend'] == 'gloo' and two['control_plane_ranks'] == 2
    assert two['cross_rank_check']['bitwise_equal'] is True and two['cross_rank_check']['ranks_checked'] == [1]
    assert two['per_rank'][0]['device_identity'] == two['per_rank'][1]['device_identity']
    # ragged batches: the two ranks pad to DIFFERENT lengths; the samples still come out bit-identical on rank 0
    r2r, rag = _bench(['--gpus', '2', '--ragged'] + common, {'LAMP_BENCH_BACKEND': 'gloo'})
    assert r2r.returncode == 0, r2r.stderr[-2000:]
    assert rag['per_rank'][0]['padded_length'] != rag['per_rank'][1]['padded_length']
    assert rag['cross_rank_check']['bitwise_equal'] is True
    assert all(p['value'] > 0 for p in two['per_rank'])
    # aggregate = all samples / slowest rank's time
    slowest = max(p['ms_per_step'] for p in two['per_rank'])
    assert abs(two['value'] - 2 * 32 / (slowest * 1e-3)) < 1e-6 * two['value']
    assert 'cpu_baseline' not in two
    # the same under the launcher the driver uses
    env = dict(os.environ, LAMP_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r3 = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                         '--master-addr', '127.0.0.1', '--master-port', '29517', os.path.join(ROOT, 'bench.py'),
                         '--gpus', '2'] + common, capture_output=True, text=True, env=env, timeout=900)
    assert r3.returncode == 0, r3.stderr[-2000:]
    three = json.loads([l for l in r3.stdout.splitlines() if l.startswith('{')][-1])
    assert three['n_gpus'] == 2 and three['config']['launcher'] == 'torch.distributed.run'
    # a rank count that does not match --gpus is an error, not a silent n_gpus: 1
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r4 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + common, capture_output=True,
                        text=True, env=env, timeout=900)
    assert r4.returncode != 0


def test_bench_kernel_only_fraction_is_measured_in_the_run(dev):
    """roofline.frac_kernel_only comes from a rocprofv3 --kernel-trace --stats sub-run of the same invocation (N = 1), not
    from a committed file: the trace must be there, count every forward, and its GEMM-class time must be below the
    instrumented HIP-event time the plain `frac` uses."""
    r, out = _bench(['--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-pipelined', '--no-extra-workloads'])
    assert r.returncode == 0, r.stderr[-2000:]
    kt, roof = out['kernel_trace'], out['roofline']
    assert kt and 'skipped' not in kt, kt
    assert kt['forwards_traced'] >= 60 and 100.0 < kt['gemm_class_us_per_forward'] < 2000.0
    assert any('gemm_nt_kernel' in k for k in kt['by_kernel']) and any('attn' in k for k in kt['by_kernel'])
    assert roof['kernel_only_source'].startswith('live:') and 'kernel_only_stale' not in roof
    assert 0.90 * roof['frac'] <= roof['frac_kernel_only'] < 1.0   # separate process: allow for clock differences
    assert abs(roof['achieved_kernel_only'] - roof['algorithmic_gflop_per_step'] / kt['gemm_class_us_per_forward'] * 1e3) \
        < 1e-6 * roof['achieved_kernel_only']
    split = roof['kernel_only_split']
    assert split['chain_kernel']['frac'] < roof['frac_kernel_only'] < split['gemm_nt_kernel']['frac'] < 1.0
    # roofline.traffic and the attention block come from counter passes of the same invocation (three short rocprofv3 --pmc runs)
    assert out['pmc'] and 'skipped' not in out['pmc'], out['pmc']
    assert roof['traffic_source'] == 'live' and roof['traffic_stale'] is False and 1e6 < roof['traffic'] < 1e9
    att = roof['attention']
    assert att['source'].startswith('live:') and 0.0 < att['hbm_frac'] < 1.0 and 0.05 < att['mfma_busy'] < 1.0
    assert any('attn16_kernel' in k for k in att['by_kernel'])


def test_bench_line_carries_the_training_step(dev):
    """bench.py's default line reports the train-loop body beside the forward metric, measured in a child
    process of the same invocation: the leg itself."""
    import bench
    line = bench.training_step_line()
    assert 'skipped' not in line, line
    assert line['batch'] == 32 and line['dropout'] == 0.1 and line['optimizer'] == 'torch.optim.Adam'
    assert line['deferred_weight_gradients'] is True and line['composite_calls'] is True
    assert 1.0 < line['ms_per_step'] < 20.0 and abs(line['value'] - 32 / (line['ms_per_step'] * 1e-3)) < 1e-6 * line['value']
    assert 0.0 < line['host_issue_ms_per_step'] <= line['ms_per_step'] * 1.02


def test_bench_and_eval_with_eight_ranks_on_one_device(dev, tmp_path):
    """The rank count the driver's scaling run uses (batch shards over 8 GPUs), as far as one GPU allows:
    eight self-spawned ranks over gloo sharing this box's device -- every rank reports, rank 0 recomputes the first samples
    of ranks 1..7 bit for bit (fixed and ragged batches), and the aggregate is all samples over the slowest rank."""
    common = ['--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--no-pipelined', '--no-extra-workloads', '--no-kernel-trace']
    for extra in ([], ['--ragged']):
        r, out = _bench(['--gpus', '8'] + extra + common, {'LAMP_BENCH_BACKEND': 'gloo'})
        assert r.returncode == 0, r.stderr[-2000:]
        assert out['n_gpus'] == 8 and out['ranks_seen'] == list(range(8)) and len(out['per_rank']) == 8
        assert out['backend'] == 'gloo' and out['control_plane_ranks'] == 8 and out['physical_devices'] == 1
        assert out['cross_rank_check']['ranks_checked'] == list(range(1, 8))
        assert out['cross_rank_check']['bitwise_equal'] is True
        slowest = max(p['ms_per_step'] for p in out['per_rank'])
        assert abs(out['value'] - 8 * 32 / (slowest * 1e-3)) < 1e-6 * out['value']
        if extra:
            assert len({p['padded_length'] for p in out['per_rank']}) > 1
        # every rank says which physical device it ran on, which CPUs it was pinned to and how long its Python side takes to
        # enqueue one forward (the host side of the 8-GPU run: eight issue loops on one host)
        assert len({p['device_identity'] for p in out['per_rank']}) == 1
        for p in out['per_rank']:
            assert p['cpu_affinity']['pinned'] and p['cpu_affinity']['cpus'] >= 1, p['cpu_affinity']
            assert 0.0 < p['host_issue_us_per_forward'] < 5000.0 and p['host_issue_frac_of_step'] > 0.0
        assert out['host_issue_us_per_forward'] == max(p['host_issue_us_per_forward'] for p in out['per_rank'])


CONTRACT_KEYS = {'metric': str, 'value': float, 'unit': str, 'n_gpus': int, 'steps': int, 'warmup': int, 'ms_per_step': float,
                 'higher_is_better': bool, 'scaling': str, 'dtype': str, 'data': str, 'config': dict}


def _assert_contract(line, n, steps, warmup, workload='reuters'):
    """The fields the round-end driver parses from bench.py's one JSON line (and computes its scaling efficiency from)."""
    for k, t in CONTRACT_KEYS.items():
        assert k in line and isinstance(line[k], t), (k, line.get(k))
    assert 'vs_baseline' in line and line['vs_baseline'] is None        # BASELINE.md publishes no number for this metric
    assert line['n_gpus'] == n and line['steps'] == steps and line['warmup'] == warmup
    assert line['unit'] == 'samples/s' and line['higher_is_better'] is True and line['scaling'] == 'weak'
    assert line['dtype'] == 'f32' and line['data'] == 'synthetic' and workload in line['config']['workload']
    assert 'model' not in line['config'] and line['value'] > 0 and line['ms_per_step'] > 0
    assert line['ranks_seen'] == list(range(n)) and len(line['per_rank']) == n
    batch = line['config']['batch_per_gpu']
    assert abs(line['value'] - n * batch / (line['ms_per_step'] * 1e-3)) < 1e-6 * line['value']   # whole job / slowest rank


def test_the_drivers_scaling_commands_rehearsed_on_one_device(dev):
    """the exact launches of the round-end scaling run -- `python bench.py --gpus 1 ...` and, for N = 2, 4, 8,
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
    --steps K --warmup W` -- with the N ranks sharing this box's one device over gloo (RCCL refuses two ranks on one device),
    plus BASELINE configs[4]'s command at a batch that fits eight ranks on one device.  Every line must parse and satisfy the
    contract; the N = 1 line carries roofline and cpu_baseline; with no RCCL the reason is in config.backend_note."""
    r1, one = _bench(['--gpus', '1', '--steps', '5', '--warmup', '2', '--no-pipelined', '--cpu-budget', '4'])
    assert r1.returncode == 0, r1.stderr[-2000:]
    _assert_contract(one, 1, 5, 2)
    roof, cpu = one['roofline'], one['cpu_baseline']
    assert roof['bound'] == 'mfma' and roof['unit'] == 'TFLOP/s' and 0.0 < roof['frac'] < 1.0 and roof['peak'] == 157.3
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-9 and roof['traffic'] is not None
    assert cpu['kind'] == 'port' and cpu['cores'] >= 1 and cpu['value'] > 0 and cpu['unit'] == 'samples/s' and cpu['sample']
    assert one['config']['weights_only_precomputation']['embed_fold'] is True
    assert one['forward']['executed_gflop_per_sample'] < one['forward']['f_live_gflop_per_sample']
    env = dict(os.environ, LAMP_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)

    def launched(n, extra, port):
        r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
                            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
                            '--gpus', str(n)] + extra, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1, lines        # ONE line, from rank 0
        return json.loads(lines[0])
    values = {1: one['value']}
    for n, port in ((2, 29531), (4, 29532), (8, 29533)):
        line = launched(n, ['--steps', '5', '--warmup', '2'], port)
        _assert_contract(line, n, 5, 2)
        assert line['config']['launcher'] == 'torch.distributed.run' and line['backend'] == 'gloo'
        assert 'backend_note' in line['config']                    # the reason when an RCCL group was wanted and could not start
        assert line['cross_rank_check']['bitwise_equal'] is True and line['cross_rank_check']['ranks_checked'] == list(range(1, n))
        assert 'cpu_baseline' not in line
        values[n] = line['value']
    # one device shared by N ranks: the aggregate must not collapse (the ranks' forwards interleave on the device)
    assert all(values[n] > 0.5 * values[1] for n in (2, 4, 8)), values
    c5 = launched(8, ['--workload', 'synthetic4096', '--batch', '4', '--steps', '2', '--warmup', '1'], 29534)
    _assert_contract(c5, 8, 2, 1, workload='synthetic4096')
    assert c5['metric'] == 'forward samples/sec, synthetic4096 d1024 2+2L 8h' and c5['config']['batch_per_gpu'] == 4
    assert c5['cross_rank_check']['bitwise_equal'] is True


def test_rccl_control_plane_calls_work(dev):
    """bench.py's control plane class with its "nccl" (= RCCL) group -- probe all_reduce, barrier(device_ids), gathers of
    device tensors -- as a one-rank group (RCCL refuses two ranks on one device, so the 2-rank tests above run on gloo)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_rccl_control_plane.py')], capture_output=True,
                       text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert r.returncode == 0 and 'rccl control plane ok' in r.stdout, r.stderr[-2000:]


def test_run_eval_sharded_over_two_ranks_equals_one(dev, tmp_path):
    """Two processes (gloo, sharing this box's GPU) each run the HIP forward on their share of the test batches;
    the gathered result must equal the one-process run: same BCE, same metrics."""
    d, sd = load_golden('harness')

    def unflatten(flat, off):
        return [flat[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]
    splits = {name: {part: unflatten(d['%s_%s_flat' % (name, part)], d['%s_%s_off' % (name, part)])
                     for part in ('src', 'tgt')} for name in ('train', 'valid', 'test')}
    src = {('w%d' % i): i for i in range(d['n_src_dict'])}
    tgt = {('l%d' % i): i for i in range(d['n_tgt_dict'])}
    data = {'settings': argparse.Namespace(max_seq_len=d['max_seq_len']), 'dict': {'src': src, 'tgt': tgt}, **splits}
    torch.save(data, tmp_path / 'train_valid_test.pt')
    torch.save({'model': {'module.' + k: v for k, v in sd.items()}}, tmp_path / 'model.chkpt')  # DataParallel-style keys
    dm = sd['decoder.tgt_word_emb.weight'].size(1)
    args = ['-data', str(tmp_path / 'train_valid_test.pt'), '-checkpoint', str(tmp_path / 'model.chkpt'), '-d_model', str(dm),
            '-d_inner_hid', str(2 * dm), '-n_layers_enc', '2', '-n_head', str(d['n_head']), '-label_mask', 'prior',
            '-batch_size', str(d['batch_size'])]
    env = dict(os.environ, LAMP_EVAL_BACKEND='gloo', PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    env.pop('WORLD_SIZE', None)
    outs = []
    for gpus in (1, 2, 8):   # 8: more ranks than batches for some of them -- empty shares must combine cleanly
        r = subprocess.run([sys.executable, '-m', 'lamp_amd.run_eval'] + args + ['-gpus', str(gpus)], capture_output=True,
                           text=True, env=env, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1]))
    one, two, eight = outs
    assert one['n_gpus'] == 1 and two['n_gpus'] == 2 and eight['n_gpus'] == 8 and d['n_batches'] >= 2
    assert abs(one['bce_total'] - d['bce_total']) < 2e-5 * d['n_batches']
    assert abs(two['bce_total'] - one['bce_total']) < 1e-9 and abs(eight['bce_total'] - one['bce_total']) < 1e-9
    for k in ('subset_accuracy', 'hamming_accuracy', 'example_f1', 'micro_f1', 'macro_f1', 'n_samples'):
        assert two[k] == one[k] and eight[k] == one[k], k


def test_dataparallel_wrapper_is_bitwise_the_plain_call(dev):
    m, sd, blocked, seq, spos, h = make_case(CONFIGS['bibtex'], dev)
    src = (seq.to(dev), spos.to(dev))
    plain, enc_plain, _ = m(src, None, None, None)
    dp = torch.nn.DataParallel(m, device_ids=[0])
    out, enc, third = dp(src, None, None, None)
    assert third is None and torch.equal(out, plain) and torch.equal(enc, enc_plain)


def _replicas(m, n):
    """What nn.DataParallel builds on every forward (torch/nn/parallel/replicate.py), on ONE device: shallow module
    copies with empty _parameters and fresh tensor copies of every weight as plain attributes."""
    try:
        from torch.nn.parallel import replicate
        return replicate(m, [0] * n)
    except Exception:
        reps = []
        for _ in range(n):
            memo = {}
            for mod in m.modules():
                memo[mod] = mod._replicate_for_data_parallel()
            for mod, rep in memo.items():
                for k, child in mod._modules.items():
                    rep._modules[k] = memo[child] if child is not None else None
                for k, p in mod._parameters.items():
                    if p is not None:
                        setattr(rep, k, p.detach().clone())
                for k, b in mod._buffers.items():
                    if b is not None:
                        setattr(rep, k, b.clone())
            reps.append(memo[m])
        return reps


def test_replicas_in_threads_match_the_original(dev):
    """DataParallel-style: replicas (no Parameters, fresh tensors, a stale copied cache) run in one host thread each."""
    m, sd, blocked, seq, spos, h = make_case(CONFIGS['reuters_ragged'], dev)
    seq, spos = seq.to(dev), spos.to(dev)
    m.fold_embedding = False    # replicas keep no weights-only tables: they run the unfolded route (same bits as this)
    plain, enc_plain, _ = m((seq, spos), None, None, None)      # also fills the original's descriptor cache
    for _ in range(2):                                          # replicas are rebuilt per forward, as DataParallel does
        reps = _replicas(m, 2)
        assert all(getattr(r, '_is_replica', False) and not list(r.parameters()) for r in reps)
        chunks = [(seq[:3], spos[:3]), (seq[3:], spos[3:])]
        results, errors = [None, None], []

        def work(i):
            try:
                with torch.cuda.device(0), torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    results[i] = reps[i](chunks[i], None, None, None)[0]
                    torch.cuda.current_stream().synchronize()
            except Exception as e:  # noqa: BLE001
                errors.append(e)
        ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errors, errors
        assert torch.equal(torch.cat(results), plain)


def test_two_threads_two_streams_are_bitwise_the_plain_call(dev):
    """The library is re-entrant: two host threads drive the SAME model on their own HIP streams concurrently."""
    m, sd, blocked, seq, spos, h = make_case(CONFIGS['reuters_ragged'], dev)
    m2, *_ = make_case(CONFIGS['bibtex'], dev)
    seq, spos = seq.to(dev), spos.to(dev)
    seq2, spos2 = (t.to(dev) for t in make_case(CONFIGS['bibtex'], dev)[3:5])
    want = m((seq, spos), None, None, None)[0]
    want2 = m2((seq2, spos2), None, None, None)[0]
    torch.cuda.synchronize()
    errors = []

    def work(model, s, p, ref):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for _ in range(25):
                    out = model((s, p), None, None, None)[0]
                st.synchronize()
            assert torch.equal(out, ref)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
    ts = [threading.Thread(target=work, args=a) for a in ((m, seq, spos, want), (m2, seq2, spos2, want2), (m, seq, spos, want))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors


def test_data_written_through_dot_data_needs_explicit_invalidation(dev):
    """Parameter._version does not move on `.data` writes: invalidate_native_cache() (also called by load_state_dict,
    train() / eval(), .to()) refreshes the hoisted layer-0 query."""
    m, sd, blocked, seq, spos, h = make_case(CONFIGS['bibtex'], dev)
    src = (seq.to(dev), spos.to(dev))
    before, _, _ = m(src, None, None, None)
    m.decoder.layer_stack[0].enc_attn.w_qs.weight.data.mul_(1.5)
    m.invalidate_native_cache()
    after, _, _ = m(src, None, None, None)
    m.cache_layer0_query = False
    m.invalidate_native_cache()
    ref, _, _ = m(src, None, None, None)
    assert torch.equal(after, ref) and not torch.equal(after, before)
    # load_state_dict and eval() invalidate on their own
    m.cache_layer0_query = True
    m.load_state_dict(sd)
    again, _, _ = m.to(dev).eval()(src, None, None, None)
    assert torch.equal(again, before)
