"""The evaluation epoch around the hot path (reference: test.py:16-58, binary-relevance branch).

Per batch: zero-pad the last batch up to `batch_size` (the reference does so because it forces
multi_gpu, test.py:35-39; the all-PAD rows come out NaN and are sliced off again),
run ``model(src, adj, None, None)`` on the MI355X, sigmoid + BCE-with-logits on the device
(lamp_sigmoid_bce_fwd), gold-binary targets on the host.

Unlike the reference, which pulls every batch's predictions to the CPU before starting the next forward
(test.py:49-56), results stay on the device until the end of the epoch; with ``streams=n`` consecutive
batches are issued round-robin on n HIP streams, so one batch's kernel ramps / tails are filled by the
other batches' kernels (the batch-32 forward is a chain of ~35 short kernels: 38k -> 46k / 50k samples/s with
2 / 4 batches in flight on reuters).
Each sample's numbers are identical in every mode.
"""
import torch

from . import _native as N
from .data import get_gold_binary


def test_epoch(model, batches, n_labels, batch_size, device, pad_last_batch=True, int_preds=False, streams=1):
    """-> (all_predictions (n, L) cpu, all_targets (n, L) cpu, bce_total float), as test.py:16-78 returns
    them.  `batches` yields ((src_seq, src_pos), adj, tgt) like lamp_amd.data.EvalBatcher."""
    model.eval()
    n = batches.n_insts
    all_targets = torch.zeros(n, n_labels)
    lanes = [torch.cuda.Stream(device=device) for _ in range(streams)] if streams > 1 else [None]
    done = []  # (row offset, real rows, probs (device), row_loss (device))
    for bi, ((src_seq, src_pos), adj, tgt) in enumerate(batches):
        real = src_seq.size(0)
        gold_binary = get_gold_binary(tgt[:, 1:], n_labels)
        lo = bi * batch_size
        all_targets[lo:lo + real] = gold_binary
        lane = lanes[bi % len(lanes)]
        with torch.cuda.stream(lane) if lane is not None else _null():
            src_seq, src_pos = src_seq.to(device), src_pos.to(device)
            if pad_last_batch and real < batch_size:
                pad = torch.zeros((batch_size - real, src_seq.size(1)), dtype=src_seq.dtype, device=device)
                src_seq = torch.cat((src_seq, pad), 0)
                src_pos = torch.cat((src_pos, pad), 0)
            pred = model((src_seq, src_pos), adj, None, None, int_preds=int_preds)[0]
            probs, row_loss = N.sigmoid_bce(pred[:real], gold_binary.to(device))
        done.append((lo, real, probs, row_loss))
    torch.cuda.synchronize(device)
    all_predictions = torch.zeros(n, n_labels)
    bce_total = 0.0
    for lo, real, probs, row_loss in done:
        all_predictions[lo:lo + real] = probs.cpu()
        bce_total += float(row_loss.cpu().double().sum()) / (real * n_labels)  # reduction='mean' per batch
    return all_predictions, all_targets, bce_total


class _null(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False
