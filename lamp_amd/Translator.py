"""Beam-search label decoding for the autoregressive baseline decoders (reference: lamp/Translator.py:14-171; called by
test.py:60 when ``opt.binary_relevance`` is off, i.e. decoder 'rnn_m').  SURVEY.md 8f n4; plain PyTorch.

Per step: gather the live beams' partial label sequences, run one decoder step, forbid labels a hypothesis already
holds (-inf before the log-softmax), let every beam keep its best continuations, and drop finished samples from the
batch.  Returns, per sample, the n_best label sequences and the per-step probabilities of the top hypothesis.
"""
import torch
import torch.nn.functional as F

from . import Constants
from .Beam import Beam


def translate(model, opt, src_batch, adj):
    src_seq, src_pos = src_batch
    device = src_seq.device
    batch_size, beam_size = src_seq.size(0), opt.beam_size
    enc_output, *_ = model.encoder(src_seq, adj, src_pos)

    # every sample appears beam_size times, hypotheses of one sample adjacent
    src_seq = src_seq.repeat(1, beam_size).view(batch_size * beam_size, src_seq.size(1))
    enc_output = enc_output.detach().repeat(1, beam_size, 1).view(batch_size * beam_size, enc_output.size(1),
                                                                  enc_output.size(2))
    beams = [Beam(beam_size, device.type == 'cuda') for _ in range(batch_size)]
    slot_of = {b: b for b in range(batch_size)}           # beam index -> row block among the still-active samples
    n_active = batch_size
    decoder_hidden = enc_output.mean(1) if opt.decoder == 'rnn_m' else None

    def keep_active(t, idx, width=None):
        """Rows of t belonging to the samples in idx (t is (n_active * beam, ...))."""
        rest = t.shape[1:]
        t = t.reshape(n_active, -1).index_select(0, idx)
        return t.reshape((len(idx) * beam_size,) + tuple(rest))

    for i in range(opt.max_token_seq_len_d):
        partial = torch.stack([b.get_current_state() for b in beams if not b.done]).view(-1, i + 1).to(device)
        if opt.decoder == 'rnn_m':
            dec_output, decoder_hidden, _ = model.decoder.step(partial[:, -1], decoder_hidden, enc_output,
                                                               src_seq.eq(Constants.PAD))
        else:
            dec_output, *_ = model.decoder(partial, src_seq, enc_output)
            dec_output = model.tgt_word_proj(dec_output[:, -1, :])
        dec_output = dec_output.detach().clone()
        dec_output.scatter_(1, partial, float('-inf'))      # a label set holds every label at most once
        word_lk = F.log_softmax(dec_output, dim=1).view(n_active, beam_size, -1)

        still = [b for b in range(batch_size) if not beams[b].done and not beams[b].advance(word_lk[slot_of[b]])]
        if not still:
            break
        idx = torch.tensor([slot_of[b] for b in still], dtype=torch.int64, device=device)
        src_seq = keep_active(src_seq, idx)
        enc_output = keep_active(enc_output, idx)
        if decoder_hidden is not None:
            decoder_hidden = keep_active(decoder_hidden, idx)
        slot_of = {b: s for s, b in enumerate(still)}
        n_active = len(still)

    all_hyp, all_hyp_scores = [], []
    for b in beams:
        _, order = b.sort_scores()
        all_hyp.append([b.get_hypothesis(k) for k in order[:opt.n_best]])
        all_hyp_scores.append([torch.exp(s)[0] for s in b.all_scores])
    return all_hyp, all_hyp_scores
