#!/usr/bin/env python3
"""Golden vectors for the thresholded multi-label metrics the reference prints after an evaluation epoch
(utils/evals.py:316-372 compute_metrics with all_metrics=False).

Runs ONLY in the build container: imports the reference's own `utils.evals.compute_metrics` and records, for a few
random (prediction, target) matrices -- including samples with no gold and no predicted label, labels that never
occur and are never predicted, an all-correct and an all-wrong case -- the five numbers it returns.  Data only.
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

from utils import evals  # noqa: E402

assert os.path.abspath(evals.__file__).startswith(os.path.abspath(REF))


def main():
    g = torch.Generator().manual_seed(11)
    out = {}
    cases = []
    for i, (n, L, p_gold, p_pred) in enumerate(((3, 3, 0.4, 0.4), (40, 17, 0.15, 0.5), (64, 90, 0.03, 0.2), (25, 8, 0.3, 0.9),
                                               (12, 5, 0.5, 0.5), (9, 4, 0.5, 0.5))):
        tgt = (torch.rand(n, L, generator=g) < p_gold).float()
        pred = torch.rand(n, L, generator=g) * (torch.rand(n, L, generator=g) < p_pred).float()
        if i == 0:   # the advisor's case: one sample with neither gold nor predicted labels, others perfect
            tgt = torch.tensor([[1., 0., 0.], [0., 1., 0.], [0., 0., 0.]])
            pred = torch.tensor([[0.9, 0.1, 0.2], [0.1, 0.8, 0.3], [0.2, 0.1, 0.4]])
        if i in (1, 2):
            tgt[3] = 0
            pred[3] = 0          # empty sample
            tgt[:, 2] = 0
            pred[:, 2] = 0       # label never gold, never predicted
        if i == 4:
            pred = tgt * 0.9 + 0.05       # all correct
        if i == 5:
            pred = (1 - tgt) * 0.9 + 0.05  # all wrong
        args = argparse.Namespace(br_threshold=0.5, decoder='graph')
        m = evals.compute_metrics(pred.clone(), tgt.clone(), 0.0, args, 0.0, all_metrics=False, verbose=False)
        out['pred_%d' % i] = pred.numpy()
        out['tgt_%d' % i] = tgt.numpy()
        out['ref_%d' % i] = np.array([m['ACC'], m['HA'], m['ebF1'], m['miF1'], m['maF1']], dtype=np.float64)
        cases.append(i)
    out['n_cases'] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, 'evals.npz'), **out)
    print('wrote evals.npz:', {k: out[k] for k in out if k.startswith('ref_')})


if __name__ == '__main__':
    main()
