#!/usr/bin/env python3
"""Golden vectors for the model choices outside the label-graph hot path: the reference's
mlp / rnn baseline encoders and decoders, the vector-encoder variants of the graph model (enc_transform), the
beam-search bookkeeping (lamp/Beam.py) and `translate` (lamp/Translator.py).

Runs ONLY in the build container: imports the reference (read-only, /root/reference) with the oracle shims of
SURVEY.md 8c plus ONE more era shim: `LongTensor / int` is integer division on the PyTorch the reference was written
for and true division today; lamp/Beam.py:57 relies on the former (its back-pointers must index), so `/` on two
integer operands is restored to floor division while the reference runs.  Data only; no reference source is copied.
"""
import argparse
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get('LAMP_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path = [REF] + [p for p in sys.path if os.path.abspath(p or '.') not in
                    (os.path.abspath(os.path.join(HERE, '..', '..')),
                     os.path.abspath(os.path.join(HERE, '..', '..', 'dropin')), HERE,
                     os.path.abspath(os.path.join(HERE, '..')))]

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self
_mf = torch.Tensor.masked_fill
torch.Tensor.masked_fill = lambda self, m, v: _mf(self, m.bool() if m.dtype == torch.uint8 else m, v)
_td = torch.Tensor.__truediv__


def _era_div(a, b):
    if not a.is_floating_point() and not a.is_complex() and (isinstance(b, int) or
                                                             (torch.is_tensor(b) and not b.is_floating_point())):
        return torch.div(a, b, rounding_mode='floor')
    return _td(a, b)


torch.Tensor.__truediv__ = _era_div

from lamp.Models import LAMP  # noqa: E402
from lamp.Beam import Beam  # noqa: E402
from lamp.Translator import translate  # noqa: E402
import lamp  # noqa: E402

assert os.path.abspath(lamp.__file__).startswith(os.path.abspath(REF))


def save(name, sd, **arrays):
    out = {'sd__' + k: v.detach().numpy() for k, v in sd.items()}
    for k, v in arrays.items():
        out[k] = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, sorted(arrays))


def build(encoder, decoder, V, L, T, d, enc_transform='', n_layers=2, n_head=1, label_mask='none', adj=None):
    torch.manual_seed(3)
    m = LAMP(V, L, T, T, n_layers_enc=n_layers, n_layers_dec=n_layers, n_head=n_head, n_head2=n_head, d_word_vec=d, d_model=d,
             d_inner_hid=2 * d, d_k=d // n_head, d_v=d // n_head, dropout=0.1, dec_dropout=0.1, dec_dropout2=False,
             proj_share_weight=(decoder != 'mlp'), encoder=encoder, decoder=decoder, enc_transform=enc_transform,
             label_adj_matrix=adj, label_mask=label_mask)
    return m.eval()


def tokens(B, V, T, lengths, g):
    seq = torch.zeros(B, T, dtype=torch.int64)
    pos = torch.zeros(B, T, dtype=torch.int64)
    for b, n in enumerate(lengths):
        seq[b, :n] = torch.randint(4, V, (n,), generator=g)
        pos[b, :n] = torch.arange(1, n + 1)
    return seq, pos


def main():
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        # 1. mlp encoder + mlp decoder: src_seq is a (B, n_src_vocab) float feature matrix
        V, L, d = 23, 9, 16
        m = build('mlp', 'mlp', V, L, 1, d)
        x = (torch.rand(5, V, generator=g) < 0.3).float()
        logits, enc, third = m((x, None), None, None, None)
        assert third is None
        save('baseline_mlp', m.state_dict(), src=x, logits=logits, enc_output=enc, n_labels=L, d_model=d, n_src=V)

        # 2. rnn encoder + rnn_m decoder (autoregressive, arg-max feeding); tgt_seq = [BOS, labels..., EOS, PAD...]
        V, L, T, d = 31, 12, 7, 16       # L counts the 4 specials here (tgt vocabulary of the multi-label-as-sequence model)
        m = build('rnn', 'rnn_m', V, L, T, d)
        seq, pos = tokens(4, V, T, [7, 3, 5, 1], g)
        tgt = torch.tensor([[2, 5, 7, 3, 0], [2, 9, 3, 0, 0], [2, 4, 6, 11, 3], [2, 8, 3, 0, 0]])
        logits, enc, _ = m((seq, pos), None, tgt, None)
        save('baseline_rnn', m.state_dict(), src_seq=seq, src_pos=pos, tgt_seq=tgt, logits=logits, enc_output=enc,
             n_labels=L, d_model=d, n_src=V, n_max_seq=T)

        # 3. translate() on the same model: beam 3, n_best 2
        opt = argparse.Namespace(cuda=False, beam_size=3, n_best=2, decoder='rnn_m', max_token_seq_len_d=6, d_model=d)
        hyp, scores = translate(m, opt, (seq, pos), None)
        flat = np.full((len(hyp), opt.n_best, opt.max_token_seq_len_d), -1, dtype=np.int64)
        for i, hs in enumerate(hyp):
            for j, h in enumerate(hs):
                flat[i, j, :len(h)] = h
        sc = np.full((len(scores), opt.max_token_seq_len_d + 1), np.nan, dtype=np.float32)
        for i, s in enumerate(scores):
            sc[i, :len(s)] = [float(v) for v in s]
        save('baseline_translate', {}, hyp=flat, hyp_scores=sc, beam_size=3, n_best=2, max_len=6)

        # 4. Beam on its own: a fixed stream of log-probabilities
        b = Beam(4, False)
        steps = []
        for t in range(5):
            lk = torch.log_softmax(torch.randn(4, 9, generator=g) * 2, dim=1)
            if t == 3:
                lk[:, 3] += 4.0          # make EOS win
            steps.append(lk)
            done = b.advance(lk)
            if done:
                break
        save('baseline_beam', {}, lk=torch.stack(steps), scores=b.scores, done=int(b.done),
             prev_ks=torch.stack([p.long() for p in b.prev_ks]), next_ys=torch.stack([y.long() for y in b.next_ys]),
             hyp0=torch.tensor(b.get_hypothesis(0)), tentative=b.get_tentative_hypothesis(),
             best=torch.stack(b.get_the_best_score_and_idx()[:1] + (b.get_the_best_score_and_idx()[1].float(),)))

        # 5. graph encoder pooled to one vector (enc_transform) + mlp decoder
        V, L, T, d = 40, 11, 9, 32
        for et in ('mean', 'sum'):
            m = build('graph', 'mlp', V, L, T, d, enc_transform=et, n_head=4)
            seq, pos = tokens(5, V, T, [9, 2, 5, 9, 7], g)
            logits, enc, _ = m((seq, pos), None, None, None)
            save('baseline_graph_%s_mlp' % et, m.state_dict(), src_seq=seq, src_pos=pos, logits=logits, enc_output=enc,
                 n_labels=L, d_model=d, n_src=V, n_max_seq=T, n_head=4)

        # 6. graph encoder pooled to one vector + graph decoder (enc_vec: no key-padding mask, one key)
        adj = (torch.rand(L, L, generator=g) < 0.3).float()
        adj = ((adj + adj.t()) > 0).float()
        adj.fill_diagonal_(1)
        m = build('graph', 'graph', V, L, T, d, enc_transform='mean', n_head=4, label_mask='prior', adj=adj.clone())
        seq, pos = tokens(5, V, T, [9, 2, 5, 9, 7], g)
        logits, enc, _ = m((seq, pos), None, None, None)
        save('baseline_graph_mean_graph', m.state_dict(), src_seq=seq, src_pos=pos, logits=logits, enc_output=enc,
             n_labels=L, d_model=d, n_src=V, n_max_seq=T, n_head=4, label_adj_matrix=adj)

        # 7. per-sample input graphs (`adj`, lamp/Encoders.py:81-85) on the plain graph model: they only show in the
        #    encoder's attention maps
        m = build('graph', 'graph', V, L, T, d, n_head=4, label_mask='prior', adj=adj.clone())
        lengths = [9, 2, 5, 9, 7]
        seq, pos = tokens(5, V, T, lengths, g)
        in_adj = []
        for n in lengths:
            a = (torch.rand(n, n, generator=g) < 0.5).float()
            a.fill_diagonal_(1)
            in_adj.append(a)
        logits, enc, enc_attns, dec2 = m((seq, pos), in_adj, None, None, return_attns=True)
        flat = torch.cat([a.reshape(-1) for a in in_adj])
        save('baseline_input_adj', m.state_dict(), src_seq=seq, src_pos=pos, logits=logits, enc_output=enc,
             n_labels=L, d_model=d, n_src=V, n_max_seq=T, n_head=4, label_adj_matrix=adj, in_adj_flat=flat,
             lengths=torch.tensor(lengths), attn_enc_0=enc_attns[0][0], attn_enc_1=enc_attns[0][1],
             attn_dec_slf_1=dec2[0][1], attn_dec_enc_1=dec2[1][1])


if __name__ == '__main__':
    main()
