#!/usr/bin/env python3
"""Golden vectors for the eval harness either side of the hot path.

Runs ONLY in the build container: imports the reference's own `utils.data_loader.process_data`,
`DataLoader` and `test.test_epoch` (with the oracle shims of SURVEY.md 8c), drives them on a small
synthetic dataset in the reference's on-disk format (utils/preprocess.py:218-232) and records
  * the dataset itself (ragged id lists, flattened + offsets),
  * what process_data derives (prior label adjacency, vocabulary sizes),
  * every batch the test DataLoader emits (src_seq, src_pos, tgt),
  * test_epoch's outputs: sigmoid predictions, gold-binary targets, summed BCE.
Data only; no reference source is copied.
"""
import argparse
import os
import random
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

from utils.data_loader import process_data  # noqa: E402
import test as ref_test  # noqa: E402  (the reference's test.py)
from lamp.Models import LAMP  # noqa: E402

assert os.path.abspath(ref_test.__file__).startswith(os.path.abspath(REF))


def flatten(lists):
    off = np.cumsum([0] + [len(x) for x in lists]).astype(np.int64)
    flat = np.array([v for x in lists for v in x], dtype=np.int64)
    return flat, off


def main():
    rng = random.Random(7)
    n_words, n_labels, max_len = 60, 11, 14
    src_dict = {'<blank>': 0, '<unk>': 1, '<s>': 2, '</s>': 3}
    src_dict.update({'w%d' % i: 4 + i for i in range(n_words)})
    tgt_dict = {'<blank>': 0, '<unk>': 1, '<s>': 2, '</s>': 3}
    tgt_dict.update({'l%d' % i: 4 + i for i in range(n_labels)})

    def sample(force_label=None):
        n = rng.randint(1, max_len)
        src = [2] + [rng.randint(4, 4 + n_words - 1) for _ in range(n)] + [3]
        k = rng.randint(1, 4)
        labels = sorted(rng.sample(range(4, 4 + n_labels), k))
        if force_label is not None and force_label not in labels:
            labels = sorted(labels + [force_label])
        return src, [2] + labels + [3]

    splits = {}
    for name, n in (('train', 40), ('valid', 9), ('test', 19)):
        items = [sample(4 + (i % n_labels) if name == 'train' else None) for i in range(n)]
        splits[name] = {'src': [s for s, _ in items], 'tgt': [t for _, t in items]}
    settings = argparse.Namespace(max_seq_len=max_len + 2)
    data = {'settings': settings, 'dict': {'src': src_dict, 'tgt': tgt_dict}, **splits}

    batch_size = 8
    opt = argparse.Namespace(adj_matrix_lambda=0.0, label_mask='prior', dataset='synthetic', summarize_data=False,
                             batch_size=batch_size, test_batch_size=batch_size, binary_relevance=True, cuda=False,
                             max_ar_length=30, multi_gpu=True, int_preds=False)
    random.seed(0)
    train_data, valid_data, test_data, adj, opt = process_data(data, opt)

    d, h = 32, 2
    torch.manual_seed(5)
    model = LAMP(opt.src_vocab_size, opt.tgt_vocab_size, opt.max_token_seq_len_e, opt.max_token_seq_len_d,
                 proj_share_weight=True, embs_share_weight=True, d_k=d // h, d_v=d // h, d_model=d, d_word_vec=d,
                 d_inner_hid=2 * d, n_layers_enc=2, n_layers_dec=2, n_head=h, n_head2=h, dropout=0.1,
                 dec_dropout=0.1, dec_dropout2=False, encoder='graph', decoder='graph', enc_transform='',
                 onehot=False, no_enc_pos_embedding=False, no_dec_self_att=False, loss='ce',
                 label_adj_matrix=adj.clone(), attn_type='softmax', label_mask='prior', matching_mlp=False,
                 graph_conv=False, int_preds=False)

    out = {}
    for name in ('train', 'valid', 'test'):
        for part in ('src', 'tgt'):
            flat, off = flatten(splits[name][part])
            out['%s_%s_flat' % (name, part)] = flat
            out['%s_%s_off' % (name, part)] = off
    out['n_src_dict'] = np.int64(len(src_dict))
    out['n_tgt_dict'] = np.int64(len(tgt_dict))
    out['max_seq_len'] = np.int64(settings.max_seq_len)
    out['batch_size'] = np.int64(batch_size)
    out['label_adj_matrix'] = adj.numpy()
    out['src_vocab_size'] = np.int64(opt.src_vocab_size)
    out['tgt_vocab_size'] = np.int64(opt.tgt_vocab_size)
    out['max_token_seq_len_e'] = np.int64(opt.max_token_seq_len_e)
    out['n_head'] = np.int64(h)

    n_batches = 0
    for bi, batch in enumerate(test_data):
        (src_seq, src_pos), adj_insts, tgt = batch
        assert adj_insts is None
        out['batch%d_src_seq' % bi] = src_seq.numpy()
        out['batch%d_src_pos' % bi] = src_pos.numpy()
        out['batch%d_tgt' % bi] = tgt.numpy()
        n_batches += 1
    out['n_batches'] = np.int64(n_batches)

    with torch.no_grad():
        preds, targets, bce_total = ref_test.test_epoch(model, test_data, opt, data['dict'], 'golden')
    out['predictions'] = preds.numpy()
    out['targets'] = targets.numpy()
    out['bce_total'] = np.float64(bce_total)
    assert not np.isnan(out['predictions']).any()
    for k, v in model.state_dict().items():
        out['sd__' + k] = v.detach().numpy()
    path = os.path.join(HERE, 'harness.npz')
    np.savez_compressed(path, **out)
    print('harness.npz %.1f KB, %d batches, bce_total %.6f' % (os.path.getsize(path) / 1024.0, n_batches, bce_total))


if __name__ == '__main__':
    main()
