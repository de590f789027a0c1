"""Special-token vocabulary shared by the source and label dictionaries.

The four specials occupy ids 0..3 of both vocabularies (the reference's preprocessing writes them first),
which is why label ids are offset by 4 wherever targets are turned into label indices
(lamp_amd/data.py).  Names and values match what the reference's callers import from ``lamp.Constants``.
"""
SPECIAL_TOKENS = ('<blank>', '<unk>', '<s>', '</s>')   # padding, unknown, begin-of-sequence, end-of-sequence
N_SPECIAL = len(SPECIAL_TOKENS)

PAD, UNK, BOS, EOS = range(N_SPECIAL)
PAD_WORD, UNK_WORD, BOS_WORD, EOS_WORD = SPECIAL_TOKENS

assert (PAD, UNK, BOS, EOS) == (0, 1, 2, 3)
