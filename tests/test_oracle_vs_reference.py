"""In-container only: the oracle against the *live* reference at the BASELINE.json sizes.

Skipped wherever /root/reference is absent (the GPU box).  Runs in a subprocess so the reference's
top-level package name `lamp` never collides with anything imported by the test session.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

SCRIPT = r'''
import sys, json
sys.dont_write_bytecode = True
sys.path.insert(0, %(ref)r)
import torch
torch.Tensor.cuda = lambda self, *a, **k: self
_mf = torch.Tensor.masked_fill
torch.Tensor.masked_fill = lambda self, m, v: _mf(self, m.bool() if m.dtype == torch.uint8 else m, v)
from lamp.Models import LAMP
sys.path.insert(0, %(root)r)
from oracle import lamp_ref as R

def run(V, L, T, d, dff, h, mask, pos_emb, B, p, lengths=None):
    adj = R.make_adjacency(L, p, seed=0) if mask == 'prior' else None
    torch.manual_seed(0)
    m = LAMP(V, L, T, L, proj_share_weight=True, embs_share_weight=True, d_k=d // h, d_v=d // h,
             d_model=d, d_word_vec=d, d_inner_hid=dff, n_layers_enc=2, n_layers_dec=2, n_head=h,
             n_head2=h, dropout=0.1, dec_dropout=0.1, dec_dropout2=False, encoder='graph',
             decoder='graph', enc_transform='', onehot=False, no_enc_pos_embedding=not pos_emb,
             no_dec_self_att=False, loss='ce',
             label_adj_matrix=adj.clone() if adj is not None else None, attn_type='softmax',
             label_mask=mask, matching_mlp=False, graph_conv=False, int_preds=False).eval()
    seq, pos = R.make_batch(B, V, T, lengths=lengths, seed=0)
    with torch.no_grad():
        lg, enc, _ = m((seq, pos), None, None, None)
    sd = m.state_dict()
    blocked = R.label_block_mask(adj, mask, L)
    out = {}
    for aw in (False, True):
        with torch.no_grad():
            lg2, enc2, _ = R.forward(sd, seq, pos, h, blocked, as_written=aw)
        out['logits_%%d' %% aw] = (lg - lg2).abs().max().item()
        out['enc_%%d' %% aw] = (enc - enc2).abs().max().item()
    return out

cases = {
  'reuters_fixed': dict(V=23666, L=90, T=302, d=512, dff=512, h=4, mask='prior', pos_emb=True, B=8, p=0.10),
  'reuters_ragged': dict(V=23666, L=90, T=302, d=512, dff=512, h=4, mask='prior', pos_emb=True, B=6, p=0.10,
                         lengths=[302, 20, 150, 77, 201, 33]),
  'bibtex': dict(V=1840, L=159, T=100, d=512, dff=1024, h=4, mask='prior', pos_emb=False, B=4, p=0.05),
  'delicious': dict(V=504, L=983, T=40, d=1024, dff=2048, h=8, mask='none', pos_emb=False, B=2, p=0.0),
  'inveye_h1': dict(V=300, L=50, T=30, d=128, dff=256, h=1, mask='inveye', pos_emb=True, B=3, p=0.0),
}
print(json.dumps({k: run(**v) for k, v in cases.items()}))
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'lamp')), reason='reference not present')
def test_oracle_matches_live_reference_at_baseline_sizes():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', PYTHONPATH='')
    r = subprocess.run([sys.executable, '-c', SCRIPT % dict(ref=REF, root=ROOT)], capture_output=True,
                       text=True, cwd='/tmp', env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    for case, errs in res.items():
        for k, v in errs.items():
            assert v <= 1e-5, (case, k, v)
