#!/usr/bin/env python3
"""An evaluation epoch end to end on one MI355X -- lamp_amd.evaluate.test_epoch, the reference's `main.py -test_only` flow
(test.py:16-58): batching + padding on the host, upload, forward, sigmoid + BCE on the device, copy back -- on a
synthetic test split in the reference's format: 3019 documents (reuters' test size), lengths U{20..300}, V = 23666, 90
labels, batch 32 -> 95 batches, model reuters d512 2+2 layers 4 heads label_mask = prior.

    python tools/bench_eval_epoch.py            # samples/s = documents / wall time of the whole call, 1 / 2 / 4 streams
"""
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lamp_amd import data as D  # noqa: E402
from lamp_amd import synthetic as S  # noqa: E402
from lamp_amd.evaluate import test_epoch  # noqa: E402
from lamp_amd.Models import LAMP  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    V, L, T, d, dff, h, n_docs, bs = 23666, 90, 302, 512, 512, 4, 3019, 32
    g = torch.Generator().manual_seed(0)
    lengths = torch.randint(20, 301, (n_docs,), generator=g).tolist()
    # instances as the reference stores them: [BOS, ids..., EOS] (utils/preprocess.py:218-232); labels = vocabulary ids >= 4
    src = [[2] + torch.randint(4, V, (n,), generator=g).tolist() + [3] for n in lengths]
    tgt = [[2] + sorted(set((torch.randint(0, L, (3,), generator=g) + 4).tolist())) + [3] for _ in range(n_docs)]
    sd = S.make_state_dict(V, L, T, d, dff, h, 2, 2, pos_emb=True, seed=0)
    adj = S.make_adjacency(L, 0.10, 0)
    m = LAMP(V, L, T, L, n_layers_enc=2, n_layers_dec=2, n_head=h, n_head2=h, d_word_vec=d, d_model=d, d_inner_hid=dff,
             d_k=d // h, d_v=d // h, encoder='graph', decoder='graph', label_adj_matrix=adj.clone(), label_mask='prior',
             dec_dropout2=False)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    out = {'documents': n_docs, 'batch': bs, 'mean_length': sum(lengths) / n_docs + 2}
    # LAMP_EVAL_GC: 'default' = CPython's collector as it comes; 'freeze' = gc.freeze() once the model and the split are loaded
    # (what lamp_amd/run_eval.py does: a full collection walks every object torch created at import, tens of milliseconds,
    # in the middle of a 60 ms epoch); 'off' = gc.disable() for the measurement
    gc_mode = os.environ.get('LAMP_EVAL_GC', 'freeze')
    out['gc'] = gc_mode
    if gc_mode == 'freeze':
        gc.collect()
        gc.freeze()
    elif gc_mode == 'off':
        gc.disable()
    prefetch = int(os.environ.get('LAMP_EVAL_PREFETCH', '8'))
    ref = None
    out['prefetch'] = prefetch
    if os.environ.get('LAMP_EVAL_NO_RAMP'):   # A/B leg: every stage `prefetch` batches from the start (the round's first version)
        from lamp_amd import evaluate as E
        E.stage_batches = lambda n_stage, prefetch: prefetch
        out['stage_ramp'] = False
    for streams, merge in ((1, False), (2, False), (4, False), (1, True)):
        rates, rates_all, lines = [], [], []
        for rep in range(6):   # first repetition warms the allocator and the clocks
            torch.cuda.synchronize()
            t_all = time.perf_counter()
            batches = D.EvalBatcher(src, tgt, bs)      # flattens the split once (the reference's DataLoader.__init__)
            t0 = time.perf_counter()
            tl = {}
            preds, targets, bce = test_epoch(m, batches, L, bs, dev, streams=streams, prefetch=prefetch, timeline=tl, merge_stage=merge)
            if ref is None:
                ref = (preds, bce)
            assert torch.equal(preds, ref[0]) and abs(bce - ref[1]) < 1e-9      # every mode: the same numbers
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            if rep:
                rates.append(n_docs / (t1 - t0))
                rates_all.append(n_docs / (t1 - t_all))
                lines.append({'ms': (t1 - t0) * 1e3, 'issued_ms': tl['issued'] * 1e3, 'device_done_ms': tl['done'] * 1e3})
        assert preds.shape == (n_docs, L) and not torch.isnan(preds).any()
        rates.sort()
        rates_all.sort()
        key = 'streams_%d' % streams + ('_merged_stages' if merge else '')
        out[key] = rates[len(rates) // 2]                 # median of five
        out[key + '_best'] = rates[-1]
        out[key + '_including_batcher_construction'] = rates_all[len(rates_all) // 2]
        out[key + '_repetitions'] = lines
    out['note'] = ('documents / wall time of one test_epoch call (padding, upload, forward, sigmoid + BCE, copy back); the second '
                   'figure also counts EvalBatcher.__init__, which flattens the split once; _merged_stages: the %d batches of a stage '
                   'as ONE forward padded to the longest (evaluate.test_epoch(merge_stage=True): same predictions, targets and '
                   'losses bit for bit -- checked in this run -- one host call per stage)' % prefetch)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
