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
    def __init__(self, size, cuda=False):
        self.size = size
        self.done = False
        self.device = torch.device('cuda') if cuda else torch.device('cpu')
        self.scores = torch.zeros(size, dtype=torch.float32, device=self.device)
        self.all_scores = []
        self.prev_ks = []                                    # back-pointers per step
        first = torch.full((size,), Constants.PAD, dtype=torch.int64, device=self.device)
        first[0] = Constants.BOS
        self.next_ys = [first]                               # tokens per step

    def get_current_state(self):
        return self.get_tentative_hypothesis()

    def get_current_origin(self):
        return self.prev_ks[-1]

    def advance(self, word_lk):
        """word_lk (size, n_words): log-probabilities of the next token for every live hypothesis."""
        num_words = word_lk.size(1)
        if len(self.prev_ks) > 0:
            beam_lk = word_lk + self.scores.unsqueeze(1).expand_as(word_lk)
        else:
            beam_lk = word_lk[0]                             # all hypotheses are still the same BOS prefix
        best_scores, best_ids = beam_lk.reshape(-1).topk(self.size, 0, True, True)
        self.all_scores.append(self.scores)
        self.scores = best_scores
        prev_k = torch.div(best_ids, num_words, rounding_mode='floor')
        self.prev_ks.append(prev_k)
        self.next_ys.append(best_ids - prev_k * num_words)
        if self.next_ys[-1][0] == Constants.EOS:             # end condition: top of the beam is EOS
            self.done = True
            self.all_scores.append(self.scores)
        return self.done

    def sort_scores(self):
        return torch.sort(self.scores, 0, True)

    def get_the_best_score_and_idx(self):
        scores, ids = self.sort_scores()
        return scores[1], ids[1]                             # sic: the reference returns the runner-up (lamp/Beam.py:79)

    def get_tentative_hypothesis(self):
        if len(self.next_ys) == 1:
            return self.next_ys[0].unsqueeze(1)
        _, keys = self.sort_scores()
        hyps = [[Constants.BOS] + self.get_hypothesis(k) for k in keys]
        return torch.tensor(hyps, dtype=torch.int64)

    def get_hypothesis(self, k):
        """Tokens of hypothesis k, oldest first, by walking the back-pointers."""
        hyp = []
        for j in range(len(self.prev_ks) - 1, -1, -1):
            hyp.append(self.next_ys[j + 1][k].item())
            k = self.prev_ks[j][k]
        return hyp[::-1]
