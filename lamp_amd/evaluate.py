"""The evaluation epoch around the hot path (reference: test.py:16-58, binary-relevance branch).

Per batch: zero-pad the last batch up to `batch_size` (the reference does so because it forces
multi_gpu, test.py:35-39; the all-PAD rows come out NaN and are sliced off again),
run ``model(src, adj, None, None)`` on the MI355X, sigmoid + BCE-with-logits on the device
(lamp_sigmoid_bce_fwd), gold-binary targets on the host.

The reference pulls every batch's predictions to the CPU before it starts the next forward (test.py:49-56): host and GPU
alternate.  Here they overlap (round 6):
  * a PRODUCER thread pads the next `prefetch` batches (utils/data_loader.py:261-279), builds their gold-binary rows and
    packs token ids, positions and targets into pinned host buffers while the device runs the previous stage;
  * the issuing thread uploads a stage with ONE asynchronous copy per buffer (ids, targets) in front of the stage's first
    forward, and issues forward + sigmoid / BCE per batch;
  * sigmoid + BCE write every batch's probabilities and per-row losses straight into ONE device matrix / vector for the whole
    split (one launch per batch, nothing else); both come back with one copy each after the last batch, and the per-batch
    mean losses (reduction='mean', test.py:51) are taken on the host in float64.
The only host-side wait is the final synchronize.  With ``streams=n`` consecutive batches are issued round-robin on n HIP
streams, so one batch's kernel ramps / tails are filled by the other batches' kernels (tools/bench_eval_epoch.py).
Each sample's numbers are identical in every mode.

The epoch's host work is small-tensor work; torch's intra-op pool is fitted to the container's CPU quota first (hostcpu.py).
"""
import queue
import threading

import numpy as np
import torch

from . import _native as N
from . import hostcpu
from . import sharding
from .data import get_gold_binary


class _Stage(object):
    """`prefetch` consecutive batches, ready to upload: ids = every batch's tokens then positions (int64, pinned), gold = their
    target rows (float32, pinned), items = (batch index, first row, rows, T, offset into ids, first gold row, adj)."""
    __slots__ = ('ids', 'gold', 'items', 'device_batches', 'slot', 'merged')


class _Slot(object):
    """One of the producer's pinned staging buffers (a ring of four, grown on demand, allocated once and kept across epochs: hipHostMalloc
    is not something to call per stage next to a busy device).  `uploaded` = the event behind the stage's host-to-device copies:
    the producer waits for it before it overwrites the buffers -- four slots, because stage k - 3 may have been taken off the
    queue without its copies being issued yet, while stage k - 4 certainly has been."""
    __slots__ = ('ids', 'gold', 'uploaded')

    def __init__(self):
        self.ids = self.gold = self.uploaded = None

    def take(self, n_ids, n_rows, n_labels, pin):
        if self.uploaded is not None:
            self.uploaded.synchronize()
            self.uploaded = None
        if self.ids is None or self.ids.numel() < n_ids:
            self.ids = torch.empty(max(n_ids, 1) * 3 // 2, dtype=torch.int64, pin_memory=pin)
        if self.gold is None or self.gold.size(0) < n_rows or self.gold.size(1) != n_labels:
            self.gold = torch.empty((max(n_rows, 1) * 3 // 2, n_labels), dtype=torch.float32, pin_memory=pin)
        return self.ids[:n_ids], self.gold[:n_rows]


# Pinned staging rings outlive an epoch: hipHostMalloc page-locks memory (tens of milliseconds now and then, next to a busy
# device), so an epoch borrows a ring of this process and gives it back; concurrent epochs each get their own.
_RINGS = []
_RINGS_LOCK = threading.Lock()


def _borrow_ring():
    with _RINGS_LOCK:
        return _RINGS.pop() if _RINGS else [_Slot() for _ in range(4)]


def _return_ring(ring):
    for slot in ring:
        slot.uploaded = None      # the epoch has synchronised: nothing of it is in flight
    with _RINGS_LOCK:
        if len(_RINGS) < 4:
            _RINGS.append(ring)


def _hand_over(out_q, item, stop):
    while not stop.is_set():
        try:
            out_q.put(item, timeout=0.05)
            return True
        except queue.Full:
            pass
    return False


def stage_batches(n_stage, prefetch):
    """Batches in stage number `n_stage`: 1, 2, 4, ... up to `prefetch` -- the first forward is issued after ONE batch has been
    padded instead of after `prefetch` of them (the device idles while the first stage is prepared)."""
    return min(prefetch, 1 << min(n_stage, 30))


def _produce(it, n_labels, batch_size, prefetch, all_targets, out_q, pin, device, stop, merge=False, ring=None):
    """Producer thread: the host side of utils/data_loader.py:242-312 + utils/utils.py:205-216 for stage after stage."""
    try:
        if pin:
            torch.cuda.set_device(device)    # pinned allocations belong to THIS rank's device context, not to device 0's
        ring, n_stage = ring if ring is not None else [_Slot() for _ in range(4)], 0
        while not stop.is_set():
            host = []
            for _ in range(stage_batches(n_stage, prefetch)):
                nxt = next(it, None)
                if nxt is None:
                    break
                host.append(nxt)
            if not host:
                break
            st = _Stage()
            st.items, st.device_batches, st.merged = [], None, None
            golds, total_ids, row = [], 0, 0
            on_device = all(b[1][0][0].is_cuda for b in host)
            merged = merge and not on_device and len(host) > 1 and all(b[1][1] is None for b in host)
            for bi, ((src_seq, src_pos), adj, tgt) in host:
                real = src_seq.size(0)
                gold = get_gold_binary(tgt[:, 1:], n_labels)
                lo = bi * batch_size
                all_targets[lo:lo + real] = gold
                golds.append(gold)
                st.items.append((bi, lo, real, src_seq.size(1), total_ids, row, adj))
                total_ids += 2 * src_seq.numel()
                row += real
            st.slot = ring[n_stage % len(ring)]
            n_stage += 1
            if merged:   # ONE token / position matrix for the whole stage, padded to its longest batch
                t_max = max(item[3] for item in st.items)
                total_ids = 2 * row * t_max
                st.merged = (row, t_max)
            st.ids, st.gold = st.slot.take(0 if on_device else total_ids, row, n_labels, pin)
            torch.cat(golds, out=st.gold)
            if on_device:    # the batcher already put the tokens on the device (EvalBatcher(device=...))
                st.ids = None
                st.device_batches = [(b[1][0][0], b[1][0][1]) for b in host]
            elif merged:
                seq_m = st.ids.numpy()[:row * t_max].reshape(row, t_max)
                pos_m = st.ids.numpy()[row * t_max:].reshape(row, t_max)
                seq_m[:] = 0
                pos_m[:] = 0
                for (bi, lo, real, T, off, r0, _), (_, ((src_seq, src_pos), _, _)) in zip(st.items, host):
                    seq_m[r0:r0 + real, :T] = src_seq.numpy()
                    pos_m[r0:r0 + real, :T] = src_pos.numpy()
            else:
                flat = st.ids.numpy()
                for (bi, lo, real, T, off, _, _), (_, ((src_seq, src_pos), _, _)) in zip(st.items, host):
                    cnt = real * T
                    flat[off:off + cnt] = src_seq.numpy().reshape(-1)
                    flat[off + cnt:off + 2 * cnt] = src_pos.numpy().reshape(-1)
            if not _hand_over(out_q, st, stop):
                return
        _hand_over(out_q, None, stop)
    except BaseException as e:  # noqa: BLE001  -- handed to the issuing thread, which re-raises it
        _hand_over(out_q, e, stop)


def test_epoch(model, batches, n_labels, batch_size, device, pad_last_batch=True, int_preds=False, streams=1,
               prefetch=8, world_size=1, rank=0, group=None, timeline=None, merge_stage=False):
    """-> (all_predictions (n, L) cpu, all_targets (n, L) cpu, bce_total float), as test.py:16-78 returns
    them.  `batches` yields ((src_seq, src_pos), adj, tgt) like lamp_amd.data.EvalBatcher.

    Multi-GPU (replaces nn.DataParallel's per-forward scatter, main.py:106-108): with world_size > 1
    this process -- one per GPU, torch.distributed initialised by the caller -- runs only its contiguous share of the
    BATCHES (sharding.shard_bounds); no collective touches the forward path.  After the last batch the per-rank
    prediction / target rows and BCE sums are combined once (all_reduce over disjoint rows), so every rank returns the
    full matrices.  A sample's numbers do not depend on world_size.

    `prefetch` = batches per stage (module docstring; the first stages ramp 1, 2, 4, ... up to it: stage_batches): the first
    forward is issued after ONE batch has been padded, and at most two further stages wait in the producer's queue.  `timeline` (a dict, optional) receives host timestamps in
    seconds from the call's start: 'issued' = the last batch was enqueued, 'done' = the device finished
    (tools/bench_eval_epoch.py: issued ~ done means the issuing thread, not the GPU, bounds the epoch).

    `merge_stage=True`: the `prefetch` batches of a stage go through the model as ONE forward, padded to the stage's longest
    batch (no all-PAD filler rows).  A sample's outputs do not depend on the batch it travels in nor on the padded length, bit
    for bit, so predictions, targets and the per-batch mean losses are the ones of the batch-by-batch loop;
    the issuing thread makes one call per stage instead of one per batch and the kernels see `prefetch` times the rows."""
    import time
    t_start = time.perf_counter()
    hostcpu.fit_intra_op_threads(world_size)   # a 256-thread OpenMP pool under a 16-core cgroup quota stalls the whole process (hostcpu.py)
    model.eval()
    n = batches.n_insts
    pin = torch.cuda.is_available()
    all_targets = torch.zeros(n, n_labels)
    b_lo, b_hi = sharding.shard_bounds(len(batches), world_size, rank) if world_size > 1 else (0, len(batches))
    if hasattr(batches, 'iter_range'):   # EvalBatcher: only this rank's batches are padded and uploaded at all
        it = zip(range(b_lo, b_hi), batches.iter_range(b_lo, b_hi))
    else:
        it = iter((bi, b) for bi, b in enumerate(batches) if b_lo <= bi < b_hi)
    main = torch.cuda.current_stream(device)
    lanes = [torch.cuda.Stream(device=device) for _ in range(streams)] if streams > 1 else [main]
    # this rank's rows of the result: probabilities and summed BCE per row, filled batch by batch on the device
    r_lo, r_hi = min(b_lo * batch_size, n), min(b_hi * batch_size, n)
    probs_d = torch.empty((max(r_hi - r_lo, 1), n_labels), dtype=torch.float32, device=device)
    row_loss_d = torch.empty((max(r_hi - r_lo, 1),), dtype=torch.float32, device=device)
    for lane in lanes:
        lane.wait_stream(main)    # the buffers (and the model's weights) are ready on every lane
    stages, stop = queue.Queue(maxsize=2), threading.Event()
    ring = _borrow_ring()
    producer = threading.Thread(target=_produce, name='lamp-eval-producer', daemon=True,
                                args=(it, n_labels, batch_size, max(int(prefetch), 1), all_targets, stages, pin, device, stop, bool(merge_stage), ring))
    producer.start()
    try:
        _issue(model, stages, lanes, device, batch_size, pad_last_batch, int_preds, probs_d, row_loss_d, r_lo)
    finally:
        stop.set()          # an exception on this side must not leave the producer blocked on a full queue
        producer.join()
    t_issued = time.perf_counter()
    for lane in lanes:
        main.wait_stream(lane)
    if timeline is not None:
        torch.cuda.synchronize(device)
        timeline.update(issued=t_issued - t_start, done=time.perf_counter() - t_start)
    all_predictions = torch.zeros(n, n_labels)
    bce_total = 0.0
    if r_hi > r_lo:
        all_predictions[r_lo:r_hi] = probs_d[:r_hi - r_lo].cpu()     # (synchronises with the main stream, which waited for the lanes)
        row_loss = row_loss_d[:r_hi - r_lo].cpu().numpy().astype(np.float64)
        # the reference adds one python float per batch, the batch's MEAN loss (test.py:51-52): same order, float64
        for lo in range(r_lo, r_hi, batch_size):
            real = min(lo + batch_size, r_hi) - lo
            bce_total += float(row_loss[lo - r_lo:lo - r_lo + real].sum()) / (real * n_labels)
    _return_ring(ring)    # (after the copies above: the device has consumed every upload of this epoch)
    return _combine_ranks(all_predictions, all_targets, bce_total, n, n_labels, world_size, device, group)


def _issue(model, stages, lanes, device, batch_size, pad_last_batch, int_preds, probs_d, row_loss_d, r_lo):
    """The issuing side: upload a stage, then forward + sigmoid / BCE per batch into the epoch's result buffers.  Buffers that
    cross streams are handed to the caching allocators' own bookkeeping (record_stream for device blocks; pinned blocks are
    not reused before the copies that read them have run), so nothing here waits for the device."""
    while True:
        st = stages.get()
        if st is None:
            return
        if isinstance(st, BaseException):
            raise st
        # the stage's two uploads go out on the FIRST lane, in order with its forwards: asynchronous for the host (pinned
        # source), ~0.1 ms of copy per stage for the device.  (A separate copy stream + an event wait per batch measured the same
        # median and a long tail -- epochs of 90-150 ms instead of 60 -- on some boxes: profiles/r06_eval_epoch_end_to_end.json.)
        with torch.cuda.stream(lanes[0]):
            ids_d = st.ids.to(device, non_blocking=True) if st.ids is not None else None
            gold_d = st.gold.to(device, non_blocking=True)
            uploaded = lanes[0].record_event()
        st.slot.uploaded = uploaded      # the producer may refill this slot's pinned buffers once these copies have run
        if st.merged is not None:
            rows, t_max = st.merged
            lo0 = st.items[0][1]
            with (torch.cuda.stream(lanes[0]) if len(lanes) > 1 else _SAME_STREAM):
                src_seq = ids_d[:rows * t_max].view(rows, t_max)
                src_pos = ids_d[rows * t_max:2 * rows * t_max].view(rows, t_max)
                pred = model((src_seq, src_pos), None, None, None, int_preds=int_preds)[0]
                N.sigmoid_bce(pred, gold_d, probs_out=probs_d[lo0 - r_lo:lo0 - r_lo + rows],
                              row_loss_out=row_loss_d[lo0 - r_lo:lo0 - r_lo + rows])
            continue
        for k, (bi, lo, real, T, off, row, adj) in enumerate(st.items):
            lane = lanes[bi % len(lanes)]
            if lane is not lanes[0]:
                lane.wait_event(uploaded)
            with (torch.cuda.stream(lane) if len(lanes) > 1 else _SAME_STREAM):
                if ids_d is not None:
                    cnt = real * T
                    src_seq = ids_d[off:off + cnt].view(real, T)
                    src_pos = ids_d[off + cnt:off + 2 * cnt].view(real, T)
                    ids_d.record_stream(lane)
                else:
                    src_seq, src_pos = st.device_batches[k]
                gold_d.record_stream(lane)
                if pad_last_batch and real < batch_size:
                    pad = torch.zeros((batch_size - real, T), dtype=src_seq.dtype, device=device)
                    src_seq = torch.cat((src_seq, pad), 0)
                    src_pos = torch.cat((src_pos, pad), 0)
                pred = model((src_seq, src_pos), adj, None, None, int_preds=int_preds)[0]
                N.sigmoid_bce(pred[:real], gold_d[row:row + real], probs_out=probs_d[lo - r_lo:lo - r_lo + real],
                              row_loss_out=row_loss_d[lo - r_lo:lo - r_lo + real])


class _SameStream(object):
    """Stand-in for torch.cuda.stream(lane) when there is one lane and it IS the current stream (the context manager costs
    the issuing thread ~15 us per batch)."""

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_SAME_STREAM = _SameStream()


def _combine_ranks(all_predictions, all_targets, bce_total, n, n_labels, world_size, device, group):
    if world_size > 1:
        import torch.distributed as dist
        on_gpu = dist.get_backend(group) == 'nccl'
        packed = torch.cat((all_predictions.reshape(-1), all_targets.reshape(-1),
                            torch.tensor([bce_total], dtype=torch.float32)))
        # rows are disjoint across ranks and zero elsewhere: the sum IS the concatenation (NaN rows cannot occur here --
        # padded rows were sliced off above); bce in float64 to keep the reference's python-float accumulation exact enough
        bce = torch.tensor([bce_total], dtype=torch.float64)
        if on_gpu:
            packed, bce = packed.to(device), bce.to(device)
        dist.all_reduce(packed, group=group)
        dist.all_reduce(bce, group=group)
        packed = packed.cpu()
        k = n * n_labels
        all_predictions = packed[:k].view(n, n_labels)
        all_targets = packed[k:2 * k].view(n, n_labels)
        bce_total = float(bce.cpu())
    return all_predictions, all_targets, bce_total
