"""Data side of the evaluation harness: what reaches the hot path and in which shape
(reference: utils/data_loader.py; on-disk format: utils/preprocess.py:218-232).

Host-side Python on purpose: this is index bookkeeping done once per dataset / per batch on the CPU in
the reference too.  The tensors it emits are exactly the ones ``LAMP.forward`` receives.
"""
import numpy as np
import torch

from . import Constants


def load_dataset(path):
    """The ``train_valid_test.pt`` dict: settings / dict{src,tgt} / train,valid,test{src,tgt} (main.py:23).
    It pickles an argparse.Namespace, hence weights_only=False."""
    return torch.load(path, weights_only=False)


def vocabulary_sizes(data, binary_relevance=True):
    """(src_vocab_size, tgt_vocab_size) as process_data derives them: the label count excludes the four
    special tokens for the graph decoder (utils/data_loader.py:119-124)."""
    n_src = len(data['dict']['src'])
    n_tgt = len(data['dict']['tgt'])
    return n_src, (n_tgt - 4 if binary_relevance else n_tgt)


def prior_adjacency(train_tgt, n_tgt_dict):
    """Label co-occurrence graph of the train split (utils/data_loader.py:37-44): identity, plus an edge
    between every two distinct labels that share a sample.  Targets are [BOS, label ids..., EOS] with ids
    offset by the four special tokens."""
    L = n_tgt_dict - 4
    adj = torch.eye(L)
    for sample in train_tgt:
        labels = [int(v) - 4 for v in sample[1:-1]]
        for i, a in enumerate(labels):
            for b in labels[i + 1:]:
                if a != b:
                    adj[a, b] = 1
                    adj[b, a] = 1
    return adj


def prior_adjacency_device(train_tgt, n_tgt_dict, device):
    """Same matrix as prior_adjacency, built by the HIP kernel (lamp_prior_graph_build): the host only flattens
    the label sets into CSR.  For delicious-sized splits the Python double loop above takes seconds; this does
    not."""
    from . import _native as N
    L = n_tgt_dict - 4
    lens = np.fromiter((max(len(s) - 2, 0) for s in train_tgt), dtype=np.int64, count=len(train_tgt))
    offsets = np.zeros(len(train_tgt) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    ids = np.empty(int(offsets[-1]), dtype=np.int64)
    for s, lo in zip(train_tgt, offsets[:-1]):
        n = len(s) - 2
        if n > 0:
            ids[lo:lo + n] = np.asarray(s[1:-1], dtype=np.int64) - 4
    return N.prior_graph(torch.from_numpy(ids).to(device), torch.from_numpy(offsets).to(device), L)


def pad_to_longest(insts):
    """-> (ids int64 (B, T), positions int64 (B, T)); T = longest instance of the batch, PAD = 0, position
    = 1-based index on non-PAD tokens and 0 on PAD (utils/data_loader.py:261-279)."""
    T = max(len(x) for x in insts)
    ids = np.full((len(insts), T), Constants.PAD, dtype=np.int64)
    for i, x in enumerate(insts):
        ids[i, :len(x)] = x
    pos = np.where(ids != Constants.PAD, np.arange(1, T + 1, dtype=np.int64)[None, :], 0)
    return torch.from_numpy(ids), torch.from_numpy(pos)


class _Flat(object):
    """A list of variable-length id lists stored once as ONE int64 array + offsets (CSR), so that padding a batch is a single
    masked store instead of a Python loop over its instances (the reference converts list by list on every batch,
    utils/data_loader.py:261-279; at reuters' size that loop costs more host time than the batch costs the GPU)."""

    def __init__(self, insts):
        import itertools
        self.lens = np.fromiter((len(x) for x in insts), dtype=np.int64, count=len(insts))
        self.offs = np.zeros(len(insts) + 1, dtype=np.int64)
        np.cumsum(self.lens, out=self.offs[1:])
        self.ids = np.fromiter(itertools.chain.from_iterable(insts), dtype=np.int64, count=int(self.offs[-1]))

    def pad(self, lo, hi):
        """pad_to_longest(insts[lo:hi]) -> (ids, positions) as torch int64 (B, T)."""
        lens = self.lens[lo:hi]
        T = int(lens.max())
        live = np.arange(T, dtype=np.int64)[None, :] < lens[:, None]
        ids = np.full((hi - lo, T), Constants.PAD, dtype=np.int64)
        ids[live] = self.ids[self.offs[lo]:self.offs[hi]]     # row-major order of the mask == concatenation order
        pos = np.where(ids != Constants.PAD, np.arange(1, T + 1, dtype=np.int64)[None, :], 0)
        return torch.from_numpy(ids), torch.from_numpy(pos)


class EvalBatcher(object):
    """Sequential (unshuffled) batches in the reference DataLoader's format
    ``((src_seq, src_pos), None, tgt)`` -- utils/data_loader.py:129-312 with shuffle=False, drop_last=False,
    as process_data builds the valid/test loaders.  Tensors are moved to `device` if given.  The instances are flattened
    once at construction (_Flat); every batch is then padded by vectorised numpy, same tensors as pad_to_longest."""

    def __init__(self, src_insts, tgt_insts, batch_size, device=None):
        if not src_insts or len(src_insts) < batch_size:
            raise ValueError('need at least batch_size instances (reference: data_loader.py:139)')
        if tgt_insts is not None and len(tgt_insts) != len(src_insts):
            raise ValueError('src / tgt instance counts differ')
        self._n = len(src_insts)
        self._src = _Flat(src_insts)
        self._tgt = _Flat(tgt_insts) if tgt_insts is not None else None
        self._batch_size = batch_size
        self._n_batch = (len(src_insts) + batch_size - 1) // batch_size
        self.device = device

    def __len__(self):
        return self._n_batch

    @property
    def n_insts(self):
        return self._n

    def __iter__(self):
        return self.iter_range(0, self._n_batch)

    def iter_range(self, b_lo, b_hi):
        """Batches b_lo .. b_hi - 1 only: a rank of a sharded evaluation materialises (pads, uploads) just its own share."""
        for b in range(max(b_lo, 0), min(b_hi, self._n_batch)):
            lo, hi = b * self._batch_size, min((b + 1) * self._batch_size, self._n)
            src_seq, src_pos = self._src.pad(lo, hi)
            tgt = None
            if self._tgt is not None:
                tgt, _ = self._tgt.pad(lo, hi)
            if self.device is not None:
                src_seq, src_pos = src_seq.to(self.device), src_pos.to(self.device)
            yield (src_seq, src_pos), None, tgt


def get_gold_binary(gold, n_labels):
    """(B, n_labels) float 0/1 matrix from padded target rows WITHOUT their leading BOS: drop PADs, drop the
    trailing EOS (the last remaining id of the row), shift ids by the four specials (utils/utils.py:205-216).
    numpy, one fancy-index store instead of the reference's per-row index_fill_ (tiny torch CPU ops pay the
    thread-pool wake-up of a 128-core host: 2 ms per call measured)."""
    g = gold.numpy() if isinstance(gold, torch.Tensor) else np.asarray(gold)
    B, W = g.shape
    out = np.zeros((B, n_labels + 4), dtype=np.float32)
    if W:
        keep = g > 0
        cols = np.arange(W)[None, :]
        last = np.where(keep, cols, -1).max(axis=1, keepdims=True)
        keep &= cols != last
        rows = np.broadcast_to(np.arange(B)[:, None], g.shape)
        out[rows[keep], g[keep]] = 1.0
    return torch.from_numpy(out[:, 4:].copy())
