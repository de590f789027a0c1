"""Pin the CPU oracle (oracle/lamp_ref.py) against every golden vector captured from the reference."""
import pytest
import torch

from conftest import golden_names, load_golden, max_abs_diff
from oracle import lamp_ref as R

TOL = 2e-5  # oracle and reference run the same ATen CPU kernels; only op grouping differs


def test_sdpa_all_masks():
    d, _ = load_golden('sdpa')
    for name in ('none', 'keypad', 'shared', 'fullrow'):
        m = d.get('mask_' + name)
        out, attn = R.sdpa(d['q'], d['k'], d['v'], m)
        assert max_abs_diff(out, d['out_' + name]) < TOL, name
        assert max_abs_diff(attn, d['attn_' + name]) < TOL, name
    # the fully masked row is NaN in the reference, and only that row
    assert torch.isnan(d['out_fullrow'][2, 3]).all()
    assert torch.isnan(d['out_fullrow']).sum() == d['out_fullrow'].size(-1)


@pytest.mark.parametrize('h', [1, 4])
def test_mha(h):
    d, sd = load_golden('mha_h%d' % h)
    p = (sd['w_qs.weight'], sd['w_ks.weight'], sd['w_vs.weight'], sd.get('fc.weight'),
         sd['layer_norm.weight'], sd['layer_norm.bias'])
    assert ('fc.weight' in sd) == (h > 1)
    for as_written in (False, True):
        o, a = R.mha(d['xq'], d['xkv'], d['pad'], *p, n_head=h, as_written=as_written)
        assert max_abs_diff(o, d['out_cross']) < TOL and max_abs_diff(a, d['attn_cross']) < TOL
        o, a = R.mha(d['xq'], d['xq'], d['slf'], *p, n_head=h, as_written=as_written)
        assert max_abs_diff(o, d['out_self']) < TOL and max_abs_diff(a, d['attn_self']) < TOL
        o, a = R.mha(d['xq'], d['xkv'], None, *p, n_head=h, as_written=as_written)
        assert max_abs_diff(o, d['out_nomask']) < TOL and max_abs_diff(a, d['attn_nomask']) < TOL


def test_ffn():
    d, sd = load_golden('ffn')
    p = (sd['w_1.weight'], sd['w_1.bias'], sd['w_2.weight'], sd['w_2.bias'],
         sd['layer_norm.weight'], sd['layer_norm.bias'])
    for as_written in (False, True):
        assert max_abs_diff(R.ffn(d['x'], *p, as_written=as_written), d['out']) < TOL


def _blocked(d, sd):
    L = sd['decoder.tgt_word_emb.weight'].size(0)
    return R.label_block_mask(d.get('label_adj_matrix'), d['label_mask'], L)


@pytest.mark.parametrize('name', golden_names('model_'))
def test_model(name):
    d, sd = load_golden(name)
    h = d['n_head']
    blocked = _blocked(d, sd)
    if 'ref_label_mask' in d:  # mask construction matches Decoders.py:109-116 (1 = blocked)
        assert torch.equal(blocked, d['ref_label_mask'].view(blocked.shape) != 0)
    for as_written in (False, True):
        logits, enc, _ = R.forward(sd, d['src_seq'], d['src_pos'], h, blocked, as_written=as_written)
        assert logits.shape == d['logits'].shape
        assert max_abs_diff(enc, d['enc_output']) < TOL
        assert max_abs_diff(logits, d['logits']) < 5e-5 * max(1, d.get('ref_gap', 0) / 2.5e-6)
    if any(k.startswith('attn_') for k in d):
        _, _, enc_attns, dec2 = R.forward(sd, d['src_seq'], d['src_pos'], h, blocked, return_attns=True)
        for i, a in enumerate(enc_attns[0]):
            assert max_abs_diff(a, d['attn_enc_%d' % i]) < TOL
        for i, a in enumerate(dec2[0]):
            assert max_abs_diff(a, d['attn_dec_slf_%d' % i]) < TOL
        for i, a in enumerate(dec2[1]):
            assert max_abs_diff(a, d['attn_dec_enc_%d' % i]) < TOL
    if 'int_pred_0' in d:
        _, _, ips = R.forward(sd, d['src_seq'], d['src_pos'], h, blocked, int_preds=True)
        n = len([k for k in d if k.startswith('int_pred_')])
        assert len(ips) == n == 3
        for i, p in enumerate(ips):
            assert max_abs_diff(p, d['int_pred_%d' % i]) < TOL


def test_allpad_row_is_nan_only_there():
    d, sd = load_golden('model_allpad_row')
    logits, _, _ = R.forward(sd, d['src_seq'], d['src_pos'], d['n_head'], _blocked(d, sd))
    assert torch.isnan(logits[1]).all() and not torch.isnan(logits[[0, 2, 3]]).any()


@pytest.mark.parametrize('scale', [1, 3, 10])
def test_conditioning_sweep_fp64(scale):
    """compare against the reference's fp64 evaluation with a tolerance scaled by the
    reference's own fp32-vs-fp64 gap."""
    d, sd = load_golden('model_qkscale_%d' % scale)
    blocked = _blocked(d, sd)
    lg32, _, _ = R.forward(sd, d['src_seq'], d['src_pos'], d['n_head'], blocked)
    lg64, _, _ = R.forward(R.to_dtype(sd, torch.float64), d['src_seq'], d['src_pos'], d['n_head'], blocked)
    assert max_abs_diff(lg64, d['logits_fp64']) < 1e-9
    assert max_abs_diff(lg32, d['logits_fp64']) <= max(1e-4, 3 * d['ref_gap'])


def test_sinusoid_table_matches_reference_table():
    d, sd = load_golden('model_prior_pos1_h4')
    ref = sd['encoder.position_enc.weight']
    assert torch.equal(R.sinusoid_table(ref.size(0), ref.size(1)), ref)
