"""Beam search belongs to the reference's autoregressive decoders ('sa_m', 'rnn_m'); the label-graph
decoder never reaches it (reference: test.py:30 vs :59, lamp/Beam.py).  Importable for drop-in
compatibility of ``import lamp.Beam``; constructing a Beam raises."""


class Beam(object):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError('beam search is outside the label-graph hot path')
