"""``translate`` drives the reference's autoregressive decoders only (reference: lamp/Translator.py,
called from test.py:59-60 when ``binary_relevance`` is false).  With ``-decoder graph`` it is never
reached; kept importable because main.py imports it unconditionally (main.py:8)."""


def translate(model, opt, src_batch, adj):
    raise NotImplementedError('translate() serves the sa_m / rnn_m decoders, which are outside the '
                              'label-graph hot path')
