"""Evaluation runner: the `main.py -test_only` flow of the reference for encoder=graph / decoder=graph,
on the MI355X path.

    python -m lamp_amd.run_eval -data data/reuters/train_valid_test.pt -dataset reuters \
           -d_model 512 -d_inner_hid 512 -n_layers_enc 2 -n_head 4 -label_mask prior \
           [-checkpoint results/.../model.chkpt] [-split test] [-batch_size 32] [-streams 2]

Flag names and derived defaults follow the reference's config_args.py (single-dash flags; n_layers_dec =
n_layers_enc :87-88, d_k = d_v = d_model / n_head :96-99, d_inner_hid = 2 d_model :110-111, no position
embedding for bibtext / delicious / bookmarks / sider :104-105, n_head2 = n_head :135-136).  The model is
built from the dataset exactly as main.py:53-88 does (vocabulary sizes, max sequence length, prior label
adjacency from the train split).  Metrics are the thresholded multi-label basics the reference prints first
(utils/evals.py: subset accuracy, Hamming accuracy, example-/micro-/macro-F1 at -br_threshold); the
sklearn-based ranking metrics are CPU post-processing outside this path.
"""
import argparse
import json
import sys
import time

import torch

from . import data as D
from .evaluate import test_epoch
from .Models import LAMP


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('-data', required=True, help='train_valid_test.pt in the reference format')
    ap.add_argument('-dataset', default='', help='dataset name (only used for the no-position-embedding rule)')
    ap.add_argument('-checkpoint', default=None, help="reference checkpoint: {'model': state_dict, ...}")
    ap.add_argument('-split', default='test', choices=['train', 'valid', 'test'])
    ap.add_argument('-batch_size', type=int, default=32)
    ap.add_argument('-d_model', type=int, default=512)
    ap.add_argument('-d_inner_hid', type=int, default=-1)
    ap.add_argument('-n_layers_enc', type=int, default=5)
    ap.add_argument('-n_layers_dec', type=int, default=None)
    ap.add_argument('-n_head', type=int, default=4)
    ap.add_argument('-n_head2', type=int, default=0)
    ap.add_argument('-label_mask', default='none', choices=['none', 'inveye', 'prior'])
    ap.add_argument('-no_dec_self_att', action='store_true')
    ap.add_argument('-no_enc_pos_embedding', action='store_true')
    ap.add_argument('-br_threshold', type=float, default=0.5)
    ap.add_argument('-streams', type=int, default=2, choices=[1, 2, 3, 4], help='batches in flight (HIP streams)')
    ap.add_argument('-seed', type=int, default=0, help='weight init seed when no checkpoint is given')
    opt = ap.parse_args(argv)
    if opt.n_layers_dec is None:
        opt.n_layers_dec = opt.n_layers_enc
    if opt.d_inner_hid == -1:
        opt.d_inner_hid = 2 * opt.d_model
    if opt.n_head2 == 0:
        opt.n_head2 = opt.n_head
    if opt.dataset in ('bibtext', 'delicious', 'bookmarks', 'sider'):
        opt.no_enc_pos_embedding = True
    if opt.d_model % opt.n_head:
        ap.error('d_model must be divisible by n_head')
    return opt


def multilabel_metrics(pred, target, threshold):
    """Thresholded metrics on (n, L) cpu tensors; rows with NaN predictions are counted as all-negative."""
    p = (torch.nan_to_num(pred, nan=0.0) >= threshold).float()
    t = target.float()
    tp = (p * t).sum(0)
    fp = (p * (1 - t)).sum(0)
    fn = ((1 - p) * t).sum(0)
    f1 = lambda a, b, c: (2 * a / (2 * a + b + c).clamp_min(1e-12))  # noqa: E731
    ex_tp = (p * t).sum(1)
    ex_den = (p.sum(1) + t.sum(1)).clamp_min(1e-12)
    return {
        'subset_accuracy': (p == t).all(dim=1).float().mean().item(),
        'hamming_accuracy': (p == t).float().mean().item(),
        'example_f1': (2 * ex_tp / ex_den).mean().item(),
        'micro_f1': f1(tp.sum(), fp.sum(), fn.sum()).item(),
        'macro_f1': f1(tp, fp, fn).mean().item(),
    }


def main(argv=None):
    opt = parse(argv)
    if not torch.cuda.is_available():
        raise SystemExit('lamp_amd.run_eval needs an MI355X: no HIP device visible (there is no CPU path)')
    device = torch.device('cuda', torch.cuda.current_device())
    data = D.load_dataset(opt.data)
    n_src, n_labels = D.vocabulary_sizes(data)
    adj = (D.prior_adjacency_device(data['train']['tgt'], len(data['dict']['tgt']), device).cpu()
           if opt.label_mask == 'prior' else None)
    d, h = opt.d_model, opt.n_head
    torch.manual_seed(opt.seed)
    model = LAMP(n_src, n_labels, data['settings'].max_seq_len, n_labels, n_layers_enc=opt.n_layers_enc,
                 n_layers_dec=opt.n_layers_dec, n_head=h, n_head2=opt.n_head2, d_word_vec=d, d_model=d,
                 d_inner_hid=opt.d_inner_hid, d_k=d // h, d_v=d // h, encoder='graph', decoder='graph',
                 no_enc_pos_embedding=opt.no_enc_pos_embedding, no_dec_self_att=opt.no_dec_self_att,
                 label_adj_matrix=adj, label_mask=opt.label_mask, dec_dropout2=False)
    if opt.checkpoint:
        ckpt = torch.load(opt.checkpoint, map_location='cpu', weights_only=False)
        model.load_state_dict(ckpt['model'] if 'model' in ckpt else ckpt)
    model = model.to(device).eval()
    split = data[opt.split]
    batches = D.EvalBatcher(split['src'], split['tgt'], opt.batch_size)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    preds, targets, bce_total = test_epoch(model, batches, n_labels, opt.batch_size, device, streams=opt.streams)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {'split': opt.split, 'n_samples': batches.n_insts, 'n_labels': n_labels, 'n_batches': len(batches),
           'bce_total': bce_total, 'seconds': dt, 'samples_per_s': batches.n_insts / dt,
           'checkpoint': opt.checkpoint}
    out.update(multilabel_metrics(preds, targets, opt.br_threshold))
    print(json.dumps(out))
    return out


if __name__ == '__main__':
    main(sys.argv[1:])
