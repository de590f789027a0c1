"""Beam-search bookkeeping for one sample (reference: lamp/Beam.py, after OpenNMT-py's Beam).

State: the running log-probability of each of `size` hypotheses, and per step the back-pointer (which hypothesis was
extended) and the token appended.  ``advance`` takes the (size, n_words) log-probabilities of the next token, keeps the
`size` best continuations and reports whether the best one just emitted EOS.

The reference computes the back-pointer as ``best_scores_id / num_words`` (lamp/Beam.py:57): integer division on the
PyTorch it was written for, TRUE division -- float back-pointers that cannot index -- on current PyTorch.  This
implementation uses the intended floor division; the golden fixtures were produced with that one semantic shim.
SURVEY.md 8f n4; plain PyTorch, runs wherever the log-probabilities live.
"""
import torch

from . import Constants


class Beam(object):
    """Attributes kept under the reference's names (test.py / Translator read them): size, done, scores,
    all_scores, prev_ks (back-pointers per step), next_ys (tokens per step; step 0 = [BOS, PAD, ...])."""

    def __init__(self, size, cuda=False):
        dev = torch.device('cuda' if cuda else 'cpu')
        start = torch.full((size,), Constants.PAD, dtype=torch.int64, device=dev)
        start[0] = Constants.BOS
        self.size, self.done, self.device = size, False, dev
        self.next_ys, self.prev_ks, self.all_scores = [start], [], []
        self.scores = torch.zeros(size, dtype=torch.float32, device=dev)

    # ---- one decoding step ------------------------------------------------------------------------------------
    def advance(self, word_lk):
        """word_lk (size, n_words): log-probabilities of the next token for every live hypothesis.  Keeps the `size`
        best (hypothesis, token) continuations; returns True once the best hypothesis ends in EOS."""
        n_words = word_lk.size(1)
        # before the first step all hypotheses are the same BOS prefix: only row 0 competes
        totals = word_lk[0] if not self.prev_ks else word_lk + self.scores.unsqueeze(1)
        top, flat = totals.reshape(-1).topk(self.size, 0, True, True)
        origin = torch.div(flat, n_words, rounding_mode='floor')
        self.all_scores.append(self.scores)
        self.scores = top
        self.prev_ks.append(origin)
        self.next_ys.append(flat - origin * n_words)
        if self.next_ys[-1][0].item() == Constants.EOS:
            self.done = True
            self.all_scores.append(self.scores)
        return self.done

    # ---- read-outs ----------------------------------------------------------------------------------------------
    def sort_scores(self):
        return self.scores.sort(0, descending=True)

    def get_the_best_score_and_idx(self):
        ranked, order = self.sort_scores()
        return ranked[1], order[1]        # sic: the reference hands back the runner-up (lamp/Beam.py:79)

    def get_current_origin(self):
        return self.prev_ks[-1]

    def get_hypothesis(self, k):
        """Tokens of hypothesis k, oldest first: follow the back-pointers from the newest step to the first."""
        tokens = []
        for step in reversed(range(len(self.prev_ks))):
            tokens.append(int(self.next_ys[step + 1][k]))
            k = self.prev_ks[step][k]
        tokens.reverse()
        return tokens

    def get_tentative_hypothesis(self):
        """(size, steps + 1) token matrix of the live hypotheses, best first, each starting with BOS."""
        if not self.prev_ks:
            return self.next_ys[0].unsqueeze(1)
        order = self.sort_scores()[1]
        return torch.tensor([[Constants.BOS] + self.get_hypothesis(k) for k in order], dtype=torch.int64)

    get_current_state = get_tentative_hypothesis
