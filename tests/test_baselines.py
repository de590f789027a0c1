"""SURVEY.md 8f n4, the remainder: the reference's baseline models (mlp / rnn encoders and decoders), beam search and
`translate` -- plain PyTorch restatements of lamp/Encoders.py:16-27,112-137, lamp/Decoders.py:16-93, lamp/Beam.py and
lamp/Translator.py -- against fixtures made by running the reference itself (tests/golden/make_golden_baselines.py).
The pure-PyTorch pieces are checked on the CPU; the combinations that contain graph layers run on the MI355X."""
import argparse

import pytest
import torch

from conftest import load_golden, max_abs_diff
from lamp_amd.Beam import Beam
from lamp_amd.Models import LAMP
from lamp_amd.Translator import translate


def _model(d, sd, encoder, decoder, enc_transform='', n_layers=2, label_mask='none'):
    h = int(d.get('n_head', 1))
    dm, L, V = int(d['d_model']), int(d['n_labels']), int(d['n_src'])
    T = int(d.get('n_max_seq', 1))
    adj = d.get('label_adj_matrix')
    m = LAMP(V, L, T, T, n_layers_enc=n_layers, n_layers_dec=n_layers, n_head=h, n_head2=h, d_word_vec=dm, d_model=dm,
             d_inner_hid=2 * dm, d_k=dm // h, d_v=dm // h, dropout=0.1, dec_dropout=0.1, dec_dropout2=False,
             proj_share_weight=(decoder != 'mlp'), encoder=encoder, decoder=decoder, enc_transform=enc_transform,
             label_adj_matrix=adj.clone() if adj is not None else None, label_mask=label_mask)
    assert sorted(m.state_dict()) == sorted(sd), (set(m.state_dict()) ^ set(sd))   # checkpoint compatibility
    m.load_state_dict(sd)
    return m.eval()


def test_every_encoder_decoder_choice_constructs():
    """main.py:57-88 builds LAMP for every -encoder / -decoder flag combination config_args.py offers."""
    for enc, dec, et in (('graph', 'graph', ''), ('mlp', 'mlp', ''), ('rnn', 'rnn_m', ''), ('graph', 'mlp', 'mean'),
                         ('graph', 'graph', 'sum'), ('mlp', 'graph', ''), ('rnn', 'graph', ''), ('graph', 'rnn_m', '')):
        m = LAMP(30, 9, 8, 8, n_layers_enc=1, n_layers_dec=1, n_head=1, n_head2=1, d_word_vec=16, d_model=16,
                 d_inner_hid=32, d_k=16, d_v=16, encoder=enc, decoder=dec, enc_transform=et,
                 proj_share_weight=(dec != 'mlp'), label_mask='none')
        assert len(list(m.get_trainable_parameters())) > 0
        assert hasattr(m, 'tgt_word_proj') == (dec != 'mlp')
    with pytest.raises(NotImplementedError):
        LAMP(30, 9, 8, 8, encoder='selfatt', decoder='graph', label_mask='none')   # the reference raises as well


def test_mlp_baseline_matches_reference():
    d, sd = load_golden('baseline_mlp')
    m = _model(d, sd, 'mlp', 'mlp')
    with torch.no_grad():
        logits, enc, third = m((d['src'], None), None, None, None)
    assert third is None and logits.shape == d['logits'].shape
    assert max_abs_diff(enc, d['enc_output']) < 1e-6 and max_abs_diff(logits, d['logits']) < 1e-6


def test_rnn_baseline_matches_reference():
    d, sd = load_golden('baseline_rnn')
    m = _model(d, sd, 'rnn', 'rnn_m')
    with torch.no_grad():
        logits, enc, _ = m((d['src_seq'], d['src_pos']), None, d['tgt_seq'], None)
    assert logits.shape == d['logits'].shape
    assert max_abs_diff(enc, d['enc_output']) < 1e-5 and max_abs_diff(logits, d['logits']) < 1e-5
    # training mode works too (plain autograd): the reference's train.py loop body
    m.train()
    out, _, _ = m((d['src_seq'], d['src_pos']), None, d['tgt_seq'], None)
    out.sum().backward()
    assert all(p.grad is not None for n, p in m.named_parameters() if 'tgt_word_proj' not in n)


def test_beam_matches_reference():
    d, _ = load_golden('baseline_beam')
    b = Beam(4, False)
    for lk in d['lk']:
        done = b.advance(lk)
    assert int(done) == int(d['done']) and b.done
    assert max_abs_diff(b.scores, d['scores']) < 1e-6
    assert torch.equal(torch.stack(b.prev_ks), d['prev_ks']) and torch.equal(torch.stack(b.next_ys), d['next_ys'])
    assert b.get_hypothesis(0) == d['hyp0'].tolist()
    assert torch.equal(b.get_tentative_hypothesis(), d['tentative'])
    s, i = b.get_the_best_score_and_idx()
    assert abs(s.item() - d['best'][0].item()) < 1e-6 and int(i) == int(d['best'][1])


def test_beam_of_full_width_finds_the_best_path():
    """Independent of the reference: with stationary per-step log-probabilities the score of a path is a sum, and a
    beam as wide as the vocabulary must return the exhaustive optimum after two steps."""
    g = torch.Generator().manual_seed(9)
    n = 6
    lk1 = torch.log_softmax(torch.randn(n, generator=g), 0)
    lk2 = torch.log_softmax(torch.randn(n, n, generator=g), 1)      # lk2[prev, next]
    b = Beam(n, False)
    b.advance(lk1.unsqueeze(0).expand(n, n))
    first = b.next_ys[-1]
    b.advance(lk2[first])
    best = max(((lk1[a] + lk2[a, c]).item(), a, c) for a in range(n) for c in range(n))
    assert abs(b.scores[0].item() - best[0]) < 1e-6 and b.get_hypothesis(0) == [best[1], best[2]]


def test_translate_matches_reference():
    d, sd = load_golden('baseline_rnn')
    t, _ = load_golden('baseline_translate')
    m = _model(d, sd, 'rnn', 'rnn_m')
    opt = argparse.Namespace(cuda=False, beam_size=int(t['beam_size']), n_best=int(t['n_best']), decoder='rnn_m',
                             max_token_seq_len_d=int(t['max_len']), d_model=int(d['d_model']))
    with torch.no_grad():
        hyp, scores = translate(m, opt, (d['src_seq'], d['src_pos']), None)
    assert len(hyp) == t['hyp'].size(0)
    for i, hs in enumerate(hyp):
        for j, h in enumerate(hs):
            want = [v for v in t['hyp'][i, j].tolist() if v >= 0]
            assert h == want, (i, j, h, want)
        want_s = t['hyp_scores'][i]
        want_s = want_s[~torch.isnan(want_s)]
        assert len(scores[i]) == want_s.numel()
        assert max_abs_diff(torch.stack([torch.as_tensor(s) for s in scores[i]]), want_s) < 1e-5


def test_translate_beam1_is_greedy_decoding():
    """Independent of the reference: beam size 1 must follow the arg-max label of every step (no label twice)."""
    d, sd = load_golden('baseline_rnn')
    m = _model(d, sd, 'rnn', 'rnn_m')
    opt = argparse.Namespace(cuda=False, beam_size=1, n_best=1, decoder='rnn_m', max_token_seq_len_d=5, d_model=int(d['d_model']))
    seq, pos = d['src_seq'][:1], d['src_pos'][:1]
    with torch.no_grad():
        hyp, _ = translate(m, opt, (seq, pos), None)
        enc, _ = m.encoder(seq, None, pos)
        hidden, prev, taken, greedy = enc.mean(1), torch.tensor([[2]]), [2], []
        for _ in range(5):
            out, hidden, _ = m.decoder.forward_step(prev, hidden.squeeze(0) if hidden.dim() == 3 else hidden, enc,
                                                    seq.eq(0).unsqueeze(1))
            out = out[-1].clone()
            out[0, taken] = float('-inf')
            nxt = int(out.argmax(1))
            greedy.append(nxt)
            if nxt == 3:
                break
            taken.append(nxt)
            prev = torch.tensor([[nxt]])
    assert hyp[0][0] == greedy


@pytest.mark.gpu
@pytest.mark.parametrize('name,dec,et,mask', [('baseline_graph_mean_mlp', 'mlp', 'mean', 'none'),
                                               ('baseline_graph_sum_mlp', 'mlp', 'sum', 'none'),
                                               ('baseline_graph_mean_graph', 'graph', 'mean', 'prior')])
def test_vector_encoder_variants_of_the_graph_model(name, dec, et, mask):
    """graph encoder (HIP kernels) pooled by enc_transform, feeding the mlp decoder (PyTorch) or the graph decoder (HIP
    kernels, one key per sample, no padding mask)."""
    d, sd = load_golden(name)
    dev = torch.device('cuda:0')
    m = _model(d, sd, 'graph', dec, enc_transform=et, label_mask=mask).to(dev)
    with torch.no_grad():
        logits, enc, _ = m((d['src_seq'].to(dev), d['src_pos'].to(dev)), None, None, None)
    assert logits.shape == d['logits'].shape
    assert max_abs_diff(enc, d['enc_output']) < 5e-5 and max_abs_diff(logits, d['logits']) < 1e-4


@pytest.mark.gpu
def test_per_sample_input_graphs_shape_only_the_encoder_maps():
    """`adj` (lamp/Encoders.py:81-85): the encoder's self-attention mask inside each sample's corner; its output is dead
    compute, so logits equal the adj-free call bit for bit and only the returned encoder maps change."""
    d, sd = load_golden('baseline_input_adj')
    dev = torch.device('cuda:0')
    m = _model(d, sd, 'graph', 'graph', label_mask='prior').to(dev)
    lengths = d['lengths'].tolist()
    adj, off = [], 0
    for n in lengths:
        adj.append(d['in_adj_flat'][off:off + n * n].view(n, n).to(dev))
        off += n * n
    src = (d['src_seq'].to(dev), d['src_pos'].to(dev))
    with torch.no_grad():
        logits, enc, enc_attns, dec2 = m(src, adj, None, None, return_attns=True)
        plain, enc_plain, _ = m(src, None, None, None)
        with_adj, _, _ = m(src, adj, None, None)
    assert max_abs_diff(logits, d['logits']) < 1e-4 and max_abs_diff(enc, d['enc_output']) < 5e-5
    for i in range(2):
        assert max_abs_diff(enc_attns[0][i], d['attn_enc_%d' % i]) < 1e-5
    assert max_abs_diff(dec2[0][1], d['attn_dec_slf_1']) < 1e-5 and max_abs_diff(dec2[1][1], d['attn_dec_enc_1']) < 1e-5
    assert torch.equal(with_adj, plain) and max_abs_diff(logits, plain) < 1e-6


@pytest.mark.gpu
def test_rnn_baseline_runs_on_the_device():
    d, sd = load_golden('baseline_rnn')
    dev = torch.device('cuda:0')
    m = _model(d, sd, 'rnn', 'rnn_m').to(dev)
    with torch.no_grad():
        logits, _, _ = m((d['src_seq'].to(dev), d['src_pos'].to(dev)), None, d['tgt_seq'].to(dev), None)
        opt = argparse.Namespace(cuda=True, beam_size=3, n_best=2, decoder='rnn_m', max_token_seq_len_d=6, d_model=int(d['d_model']))
        hyp, _ = translate(m, opt, (d['src_seq'].to(dev), d['src_pos'].to(dev)), None)
    t, _ = load_golden('baseline_translate')
    assert max_abs_diff(logits, d['logits']) < 1e-4
    assert hyp[0][0] == [v for v in t['hyp'][0, 0].tolist() if v >= 0]


@pytest.mark.gpu
@pytest.mark.parametrize('name,dec,et,mask', [('baseline_graph_mean_mlp', 'mlp', 'mean', 'none'),
                                               ('baseline_graph_mean_graph', 'graph', 'mean', 'prior')])
def test_composite_graph_models_train_through_the_hip_path(name, dec, et, mask):
    """The model combinations outside the fused launcher (graph encoder pooled by enc_transform, feeding the mlp or the
    graph decoder) record autograd module by module in train(): logits equal the eval forward, and the gradient agrees
    with a central finite difference of the EVAL-mode loss (a different code path: the inference kernels) along a
    random direction over all parameters."""
    d, sd = load_golden(name)
    dev = torch.device('cuda:0')
    m = _model(d, sd, 'graph', dec, enc_transform=et, label_mask=mask).to(dev)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    src = (d['src_seq'].to(dev), d['src_pos'].to(dev))
    tgt = (torch.rand(d['logits'].shape, generator=torch.Generator().manual_seed(1)) < 0.3).float().to(dev)
    loss_fn = torch.nn.functional.binary_cross_entropy_with_logits

    def eval_loss():
        m.eval()
        with torch.no_grad():
            return loss_fn(m(src, None, None, None)[0].double(), tgt.double()).item()

    base = eval_loss()
    m.train()
    logits = m(src, None, None, tgt)[0]
    assert logits.requires_grad and abs(loss_fn(logits.double(), tgt.double()).item() - base) < 1e-6
    loss_fn(logits, tgt).backward()
    params = [p for p in m.get_trainable_parameters() if p.grad is not None]
    assert len(params) > 8
    g = torch.Generator().manual_seed(2)
    dirs = [torch.randn(p.shape, generator=g).to(dev) * (p.grad.abs() > 0).float() for p in params]
    analytic = sum((p.grad.double() * u.double()).sum().item() for p, u in zip(params, dirs))
    eps = 1e-2 / max(1e-12, sum((u.double() ** 2).sum().item() for u in dirs) ** 0.5)
    with torch.no_grad():
        for p, u in zip(params, dirs):
            p.add_(eps * u)
        m.invalidate_native_cache()
        up = eval_loss()
        for p, u in zip(params, dirs):
            p.sub_(2 * eps * u)
        m.invalidate_native_cache()
        down = eval_loss()
    numeric = (up - down) / (2 * eps)
    assert abs(numeric - analytic) < 3e-2 * max(abs(analytic), 1e-3), (numeric, analytic)
