"""Evaluation runner: the `main.py -test_only` flow of the reference for encoder=graph / decoder=graph,
on the MI355X path.

    python -m lamp_amd.run_eval -data data/reuters/train_valid_test.pt -dataset reuters \
           -d_model 512 -d_inner_hid 512 -n_layers_enc 2 -n_head 4 -label_mask prior \
           [-checkpoint results/.../model.chkpt] [-split test] [-batch_size 32] [-streams 2] [-gpus N]

-gpus N starts one process per GPU (rendezvous on 127.0.0.1; "nccl" = RCCL for the final gather only); every rank
evaluates its contiguous share of the batches -- the forward path needs no collective (lamp_amd/sharding.py).

Flag names and derived defaults follow the reference's config_args.py (single-dash flags; n_layers_dec =
n_layers_enc :87-88, d_k = d_v = d_model / n_head :96-99, d_inner_hid = 2 d_model :110-111, no position
embedding for bibtext / delicious / bookmarks / sider :104-105, n_head2 = n_head :135-136).  The model is
built from the dataset exactly as main.py:53-88 does (vocabulary sizes, max sequence length, prior label
adjacency from the train split).  Metrics are the thresholded multi-label basics the reference prints first
(utils/evals.py:316-372: subset accuracy, Hamming accuracy, example-/micro-/macro-F1 at -br_threshold, with the
reference's conventions for empty samples / labels); the sklearn-based ranking metrics are CPU post-processing
outside this path.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

from . import data as D
from . import sharding
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
    ap.add_argument('-streams', type=int, default=4, choices=[1, 2, 3, 4],
                    help='batches in flight (HIP streams); 4 measured best: 36.6 k / 43.3 k / 46.2 k samples/s with 1 / 2 / 4 on a '
                         'reuters-sized split (tools/bench_eval_epoch.py)')
    ap.add_argument('-prefetch', type=int, default=8, help='batches per stage of the evaluation epoch (padded by the producer thread while the device runs the previous stage)')
    ap.add_argument('-merge_stages', action='store_true',
                    help='one forward per stage instead of one per batch (same predictions, targets and losses bit for bit)')
    ap.add_argument('-seed', type=int, default=0, help='weight init seed when no checkpoint is given')
    ap.add_argument('-gpus', type=int, default=1, help='processes (one per GPU) the batches are sharded over')
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
    """Thresholded metrics on (n, L) cpu tensors, with the conventions of the reference's utils/evals.py:
    example-based F1 averages only over samples with at least one gold or predicted label (example_f1_score
    :105-123 deletes zero denominators), macro-F1 only over labels with tp + fp + fn > 0 (f1_score_from_stats
    :141-147 drops the non-finite ratios).  Rows with NaN predictions are counted as all-negative."""
    p = (torch.nan_to_num(pred, nan=0.0) >= threshold).double()
    t = target.double()
    tp = (p * t).sum(0)
    fp = (p * (1 - t)).sum(0)
    fn = ((1 - p) * t).sum(0)
    ex_tp = (p * t).sum(1)
    ex_den = p.sum(1) + t.sum(1)
    ex_ok = ex_den > 0
    lab_den = 2 * tp + fp + fn
    lab_ok = lab_den > 0
    nan = float('nan')
    return {
        'subset_accuracy': (p == t).all(dim=1).double().mean().item(),
        'hamming_accuracy': (p == t).double().mean().item(),
        'example_f1': (2 * ex_tp[ex_ok] / ex_den[ex_ok]).mean().item() if ex_ok.any() else nan,
        'micro_f1': (2 * tp.sum() / lab_den.sum()).item() if lab_den.sum() > 0 else nan,
        'macro_f1': (2 * tp[lab_ok] / lab_den[lab_ok]).mean().item() if lab_ok.any() else nan,
    }


def load_checkpoint_state(path):
    """state_dict of a reference checkpoint ({'model': state_dict, ...} or a bare state_dict).  Hosts with more than
    one GPU save from inside nn.DataParallel (main.py:106-108 before utils.save_model): `module.`-prefixed keys, which
    LAMP.load_state_dict strips."""
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    return ckpt['model'] if isinstance(ckpt, dict) and 'model' in ckpt else ckpt


def spawn_ranks(n, argv):
    """One process per GPU, each re-running this module with RANK / WORLD_SIZE set; rank 0 prints the result."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, '-m', 'lamp_amd.run_eval'] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0:
                rc = rc or code
                for q in alive:   # a dead rank leaves the others in a collective: stop exactly the ones we started
                    q.terminate()
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    opt = parse(argv)
    if not torch.cuda.is_available():
        raise SystemExit('lamp_amd.run_eval needs an MI355X: no HIP device visible (there is no CPU path)')
    if opt.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        rc = spawn_ranks(opt.gpus, argv)
        if rc:
            raise SystemExit(rc)
        return None
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    dev_index = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count() if world > 1 else torch.cuda.current_device()
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if world > 1:   # this rank's issue loop and host threads on the cores next to its device (sharding.pin_rank_to_device_cpus)
        sharding.pin_rank_to_device_cpus(int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    # gloo rendezvous + an RCCL group for the final gather when RCCL comes up (probed; falls back to gloo and says so);
    # LAMP_EVAL_BACKEND=gloo lets several ranks share one GPU (tests)
    plane = sharding.ControlPlane(rank, world, device, os.environ.get('LAMP_EVAL_BACKEND', 'nccl'))
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
        model.load_state_dict(load_checkpoint_state(opt.checkpoint))
    model = model.to(device).eval()
    split = data[opt.split]
    batches = D.EvalBatcher(split['src'], split['tgt'], opt.batch_size)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    preds, targets, bce_total = test_epoch(model, batches, n_labels, opt.batch_size, device, streams=opt.streams,
                                           prefetch=opt.prefetch, merge_stage=opt.merge_stages,
                                           world_size=world, rank=rank, group=plane.group)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {'split': opt.split, 'n_samples': batches.n_insts, 'n_labels': n_labels, 'n_batches': len(batches),
           'bce_total': bce_total, 'seconds': dt, 'samples_per_s': batches.n_insts / dt,
           'checkpoint': opt.checkpoint, 'n_gpus': world, 'backend': plane.backend, 'backend_note': plane.note}
    out.update(multilabel_metrics(preds, targets, opt.br_threshold))
    if rank == 0:
        print(json.dumps(out), flush=True)
    plane.close()
    return out


if __name__ == '__main__':
    main()
