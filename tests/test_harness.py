"""Eval harness either side of the hot path against golden/harness.npz, which holds what
the reference's process_data / DataLoader / test_epoch produced on a synthetic dataset."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, max_abs_diff
from oracle import lamp_ref as oracle
from lamp_amd import data as D


def unflatten(flat, off):
    return [flat[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]


@pytest.fixture(scope='module')
def fx():
    d, sd = load_golden('harness')
    splits = {}
    for name in ('train', 'valid', 'test'):
        splits[name] = {part: unflatten(d['%s_%s_flat' % (name, part)], d['%s_%s_off' % (name, part)])
                        for part in ('src', 'tgt')}
    return d, sd, splits


def test_prior_adjacency_and_vocab_sizes(fx):
    d, sd, splits = fx
    adj = D.prior_adjacency(splits['train']['tgt'], d['n_tgt_dict'])
    assert torch.equal(adj, d['label_adj_matrix'])
    data = {'dict': {'src': range(d['n_src_dict']), 'tgt': range(d['n_tgt_dict'])}}
    assert D.vocabulary_sizes(data) == (d['src_vocab_size'], d['tgt_vocab_size'])


def test_batcher_emits_the_reference_batches_bitwise(fx):
    d, sd, splits = fx
    b = D.EvalBatcher(splits['test']['src'], splits['test']['tgt'], d['batch_size'])
    assert len(b) == d['n_batches'] and b.n_insts == len(splits['test']['src'])
    for i, ((src_seq, src_pos), adj, tgt) in enumerate(b):
        assert adj is None
        for got, key in ((src_seq, 'src_seq'), (src_pos, 'src_pos'), (tgt, 'tgt')):
            ref = d['batch%d_%s' % (i, key)]
            assert got.dtype == torch.int64 and torch.equal(got, ref), (i, key)
    with pytest.raises(ValueError):
        D.EvalBatcher(splits['test']['src'][:3], None, 8)


def test_gold_binary_matches_reference_targets(fx):
    d, sd, splits = fx
    L = d['tgt_vocab_size']
    rows = []
    for (_, _, tgt) in D.EvalBatcher(splits['test']['src'], splits['test']['tgt'], d['batch_size']):
        rows.append(D.get_gold_binary(tgt[:, 1:], L))
    assert torch.equal(torch.cat(rows), d['targets'])


def test_pad_to_longest_edge_cases():
    ids, pos = D.pad_to_longest([[2, 7, 3], [2, 3], [2, 9, 9, 9, 3]])
    assert ids.tolist() == [[2, 7, 3, 0, 0], [2, 3, 0, 0, 0], [2, 9, 9, 9, 3]]
    assert pos.tolist() == [[1, 2, 3, 0, 0], [1, 2, 0, 0, 0], [1, 2, 3, 4, 5]]


def test_eval_producer_stages_hold_exactly_the_batches(fx):
    """Host side of evaluate.test_epoch (round 6: a producer thread packs `prefetch` batches per stage into one id buffer
    and one target block): every stage, unpacked, is the batcher's own tensors and get_gold_binary's rows."""
    import queue
    import threading
    from lamp_amd import evaluate as E
    d, sd, splits = fx
    L = sd['decoder.tgt_word_emb.weight'].size(0)
    bs = d['batch_size']
    b = D.EvalBatcher(splits['test']['src'], splits['test']['tgt'], bs)
    want = list(b)
    for prefetch in (1, 3, 64):
        q, stop = queue.Queue(), threading.Event()
        targets = torch.zeros(b.n_insts, L)
        E._produce(zip(range(len(b)), iter(b)), L, bs, prefetch, targets, q, False, torch.device('cpu'), stop)
        stages = []
        while True:
            st = q.get_nowait()
            if st is None:
                break
            assert not isinstance(st, BaseException), st
            stages.append(st)
        want_stages, left = 0, len(b)      # stages ramp 1, 2, 4, ... up to `prefetch` batches (evaluate.stage_batches)
        while left > 0:
            left -= E.stage_batches(want_stages, prefetch)
            want_stages += 1
        assert len(stages) == want_stages and q.empty()
        assert [len(st.items) for st in stages[:-1]] == [E.stage_batches(i, prefetch) for i in range(want_stages - 1)]
        seen = 0
        for st in stages:
            for bi, lo, real, T, off, row, adj in st.items:
                (seq, pos), _, tgt = want[bi]
                assert bi == seen and lo == bi * bs and (real, T) == tuple(seq.shape)
                assert torch.equal(st.ids[off:off + real * T].view(real, T), seq)
                assert torch.equal(st.ids[off + real * T:off + 2 * real * T].view(real, T), pos)
                assert torch.equal(st.gold[row:row + real], D.get_gold_binary(tgt[:, 1:], L))
                seen += 1
        assert seen == len(b) and torch.equal(targets, d['targets'])
    # merged stages: ONE (rows, longest T) token matrix and one position matrix per stage, zero-padded
    q, stop = queue.Queue(), threading.Event()
    E._produce(zip(range(len(b)), iter(b)), L, bs, 3, torch.zeros(b.n_insts, L), q, False, torch.device('cpu'), stop, True)
    while True:
        st = q.get_nowait()
        if st is None:
            break
        if len(st.items) == 1:
            assert st.merged is None
            continue
        rows, t_max = st.merged
        assert rows == sum(i[2] for i in st.items) and t_max == max(i[3] for i in st.items)
        seq_m = st.ids[:rows * t_max].view(rows, t_max)
        pos_m = st.ids[rows * t_max:].view(rows, t_max)
        for bi, lo, real, T, off, r0, adj in st.items:
            (seq, pos), _, _ = want[bi]
            assert torch.equal(seq_m[r0:r0 + real, :T], seq) and torch.equal(pos_m[r0:r0 + real, :T], pos)
            assert not seq_m[r0:r0 + real, T:].any() and not pos_m[r0:r0 + real, T:].any()
    # an exception inside the producer reaches the consumer instead of hanging it
    q, stop = queue.Queue(), threading.Event()
    E._produce(iter([(0, ((torch.zeros(2, 3, dtype=torch.long),) * 2, None, None))]), L, bs, 2, targets, q, False,
               torch.device('cpu'), stop)
    assert isinstance(q.get_nowait(), BaseException)


@pytest.mark.gpu
def test_test_epoch_matches_reference(fx):
    d, sd, splits = fx
    from lamp_amd.Models import LAMP
    from lamp_amd.evaluate import test_epoch
    dev = torch.device('cuda:0')
    L, dm = sd['decoder.tgt_word_emb.weight'].shape
    h = d['n_head']
    m = LAMP(d['src_vocab_size'], L, d['max_token_seq_len_e'], L, n_layers_enc=2, n_layers_dec=2, n_head=h,
             n_head2=h, d_word_vec=dm, d_model=dm, d_inner_hid=2 * dm, d_k=dm // h, d_v=dm // h, encoder='graph',
             decoder='graph', label_adj_matrix=d['label_adj_matrix'].clone(), label_mask='prior',
             dec_dropout2=False)
    m.load_state_dict(sd)
    m = m.to(dev)
    batches = D.EvalBatcher(splits['test']['src'], splits['test']['tgt'], d['batch_size'])
    preds, targets, bce = test_epoch(m, batches, L, d['batch_size'], dev)
    assert torch.equal(targets, d['targets'])
    assert not torch.isnan(preds).any()          # the zero-padded rows of the last batch were sliced off
    assert max_abs_diff(preds, d['predictions']) < 2e-5
    assert abs(bce - d['bce_total']) < 2e-5 * d['n_batches']
    # without the reference's last-batch padding the kept rows are bit-identical (samples are independent)
    preds2, _, bce2 = test_epoch(m, batches, L, d['batch_size'], dev, pad_last_batch=False)
    assert torch.equal(preds2, preds) and abs(bce2 - bce) < 1e-7
    # two batches in flight on two HIP streams: same numbers
    preds3, targets3, bce3 = test_epoch(m, batches, L, d['batch_size'], dev, streams=2)
    assert torch.equal(preds3, preds) and torch.equal(targets3, targets) and abs(bce3 - bce) < 1e-7
    # stages of several batches as ONE forward (padded to the stage's longest batch), small stages too: same numbers
    for prefetch in (2, 8):
        preds4, targets4, bce4 = test_epoch(m, batches, L, d['batch_size'], dev, prefetch=prefetch, merge_stage=True)
        assert torch.equal(preds4, preds) and torch.equal(targets4, targets) and abs(bce4 - bce) < 1e-7


@pytest.mark.gpu
def test_sigmoid_bce_kernel_vs_torch():
    from lamp_amd import _native as N
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 90, generator=g) * 6
    x[0, 0], x[0, 1] = 80.0, -80.0
    z = (torch.rand(37, 90, generator=g) < 0.2).float()
    probs, row_loss = N.sigmoid_bce(x.to(dev), z.to(dev))
    ref_loss = torch.nn.functional.binary_cross_entropy_with_logits(x.double(), z.double(), reduction='none').sum(1)
    assert max_abs_diff(probs, torch.sigmoid(x.double())) < 1e-6
    assert max_abs_diff(row_loss, ref_loss) < 1e-3 * 1e-1


def test_multilabel_metrics_known_values():
    from lamp_amd.run_eval import multilabel_metrics
    pred = torch.tensor([[0.9, 0.1, 0.8], [0.2, 0.7, 0.6], [float('nan')] * 3])
    tgt = torch.tensor([[1., 0., 1.], [0., 1., 0.], [1., 0., 0.]])
    m = multilabel_metrics(pred, tgt, 0.5)
    assert abs(m['subset_accuracy'] - 1 / 3) < 1e-6
    assert abs(m['hamming_accuracy'] - 7 / 9) < 1e-6
    assert abs(m['micro_f1'] - 2 * 3 / (2 * 3 + 1 + 1)) < 1e-6
    assert abs(m['example_f1'] - (1.0 + 2 / 3 + 0.0) / 3) < 1e-6


def test_multilabel_metrics_match_reference_evals():
    """run_eval.multilabel_metrics vs the reference's own utils/evals.py:compute_metrics (tests/golden/evals.npz, made by
    make_golden_evals.py): empty samples are dropped from example-F1, never-seen labels from macro-F1."""
    import numpy as np
    from lamp_amd.run_eval import multilabel_metrics
    z = np.load(os.path.join(GOLDEN, 'evals.npz'))
    for i in range(int(z['n_cases'])):
        m = multilabel_metrics(torch.from_numpy(z['pred_%d' % i]), torch.from_numpy(z['tgt_%d' % i]), 0.5)
        got = [m['subset_accuracy'], m['hamming_accuracy'], m['example_f1'], m['micro_f1'], m['macro_f1']]
        for a, b in zip(got, z['ref_%d' % i].tolist()):
            assert abs(a - b) < 1e-6, (i, got, z['ref_%d' % i])


def test_dataparallel_checkpoint_keys_load(fx, tmp_path):
    """main.py:106-108 wraps the model in nn.DataParallel before utils.save_model on multi-GPU hosts: the checkpoint's
    keys then carry a `module.` prefix.  run_eval's loader + LAMP.load_state_dict accept both forms."""
    from lamp_amd import run_eval
    from test_host_cpu import build_from_fixture
    m, d, sd = build_from_fixture('model_prior_pos1_h4')
    torch.save({'model': {'module.' + k: v for k, v in sd.items()}, 'epoch': 1}, tmp_path / 'dp.chkpt')
    state = run_eval.load_checkpoint_state(str(tmp_path / 'dp.chkpt'))
    assert all(k.startswith('module.') for k in state)
    res = m.load_state_dict(state)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k


@pytest.mark.gpu
def test_run_eval_end_to_end(fx, tmp_path):
    """The main.py -test_only flow on the MI355X path: reference-format .pt dataset + reference-format checkpoint
    in, the reference's own test_epoch numbers out (BCE from the golden harness fixture)."""
    import argparse
    from lamp_amd import run_eval
    d, sd, splits = fx
    src = {('w%d' % i): i for i in range(d['n_src_dict'])}
    tgt = {('l%d' % i): i for i in range(d['n_tgt_dict'])}
    data = {'settings': argparse.Namespace(max_seq_len=d['max_seq_len']), 'dict': {'src': src, 'tgt': tgt}, **splits}
    torch.save(data, tmp_path / 'train_valid_test.pt')
    torch.save({'model': sd, 'epoch': 3}, tmp_path / 'model.chkpt')
    dm = sd['decoder.tgt_word_emb.weight'].size(1)
    for streams in (1, 2):
        out = run_eval.main(['-data', str(tmp_path / 'train_valid_test.pt'), '-checkpoint', str(tmp_path / 'model.chkpt'),
                             '-d_model', str(dm), '-d_inner_hid', str(2 * dm), '-n_layers_enc', '2', '-n_head',
                             str(d['n_head']), '-label_mask', 'prior', '-batch_size', str(d['batch_size']),
                             '-streams', str(streams)])
        assert out['n_samples'] == len(splits['test']['src']) and out['n_batches'] == d['n_batches']
        assert abs(out['bce_total'] - d['bce_total']) < 2e-5 * d['n_batches']
        ref = run_eval.multilabel_metrics(d['predictions'], d['targets'], 0.5)
        for k, v in ref.items():
            assert abs(out[k] - v) < 1e-6, k


@pytest.mark.gpu
def test_prior_graph_kernel_matches_reference_adjacency(fx):
    """lamp_prior_graph_build vs the reference's own label_adj_matrix (golden, utils/data_loader.py:37-47): bit-exact,
    and the derived blocked mask equals Decoders.py:105-113's."""
    from lamp_amd import _native as N
    d, _, splits = fx
    dev = torch.device('cuda')
    adj = D.prior_adjacency_device(splits['train']['tgt'], d['n_tgt_dict'], dev)
    assert torch.equal(adj.cpu(), d['label_adj_matrix'])
    L = d['n_tgt_dict'] - 4
    flat = torch.cat([torch.as_tensor(s[1:-1]) - 4 for s in splits['train']['tgt']]).to(dev)
    off = torch.tensor([0] + [len(s) - 2 for s in splits['train']['tgt']]).cumsum(0).to(dev)
    adj2, blocked = N.prior_graph(flat, off, L, want_blocked=True)
    assert torch.equal(adj2, adj)
    assert torch.equal(blocked.cpu().bool(), oracle.label_block_mask(d['label_adj_matrix'], 'prior', L))


@pytest.mark.gpu
@pytest.mark.parametrize('L,n_samples,max_set', [(1, 3, 1), (7, 0, 0), (37, 50, 6), (983, 4000, 19), (4096, 3000, 12)])
def test_prior_graph_kernel_random_sets(L, n_samples, max_set):
    """Random label sets incl. empty samples, singletons and repeated labels, against the oracle's double loop
    (small) and a dense Y^T Y > 0 restatement (all sizes)."""
    from lamp_amd import _native as N
    g = torch.Generator().manual_seed(L * 31 + n_samples)
    sets = [torch.randint(0, L, (int(torch.randint(0, max_set + 1, (1,), generator=g)),), generator=g)
            for _ in range(n_samples)]
    if n_samples > 2 and max_set > 1:
        sets[1] = torch.cat([sets[1], sets[1]])  # repeated labels in one sample
    flat = torch.cat(sets) if n_samples and sum(len(s) for s in sets) else torch.zeros(0, dtype=torch.int64)
    off = torch.tensor([0] + [len(s) for s in sets], dtype=torch.int64).cumsum(0)
    adj, blocked = N.prior_graph(flat.cuda(), off.cuda(), L, want_blocked=True)
    Y = torch.zeros(max(n_samples, 1), L, dtype=torch.float64)
    for i, s in enumerate(sets):
        Y[i, s] = 1
    want = (((Y.t() @ Y) > 0) | torch.eye(L, dtype=torch.bool)).float()
    assert torch.equal(adj.cpu(), want)
    assert torch.equal(blocked.cpu(), (want == 0).to(torch.uint8))
    if L <= 64:
        assert torch.equal(adj.cpu(), oracle.prior_adjacency([(s + 4).tolist() for s in sets], L))


@pytest.mark.gpu
def test_prior_graph_rejects_bad_input():
    from lamp_amd import _native as N
    ids, off = torch.tensor([0, 5]).cuda(), torch.tensor([0, 2]).cuda()
    with pytest.raises(IndexError):
        N.prior_graph(ids, off, 5)
    with pytest.raises(IndexError):
        N.prior_graph(torch.tensor([-1, 2]).cuda(), off, 5)
    with pytest.raises(ValueError):
        N.prior_graph(torch.tensor([1, 2]).cuda(), torch.tensor([0, 3]).cuda(), 5)
    with pytest.raises(TypeError):
        N.prior_graph(torch.tensor([1, 2], dtype=torch.int32).cuda(), off, 5)
    with pytest.raises(RuntimeError):
        N.prior_graph(torch.tensor([1, 2]), torch.tensor([0, 2]), 5)
