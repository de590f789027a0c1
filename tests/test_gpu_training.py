"""Training through the HIP path (reference train.py:36-48): loss.backward() on lamp_amd's LAMP in
train() mode against torch.autograd on the fp64 CPU oracle -- every parameter's gradient -- and dropout checked
against a plain-torch restatement that uses the library's (counter-based, reproducible) keep masks."""
import pytest
import torch
import torch.nn.functional as F

from conftest import max_abs_diff
from oracle import lamp_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


CASES = {
    # name: V, L, T, d, dff, h, mask, pos_emb, B, p_adj, lengths
    'tiny_prior_h4': (50, 37, 23, 64, 96, 4, 'prior', True, 3, 0.2, [23, 5, 14]),
    'tiny_none_h1': (40, 20, 17, 32, 48, 1, 'none', False, 2, 0.0, [17, 9]),
    'inveye_h8': (60, 70, 33, 128, 256, 8, 'inveye', True, 3, 0.0, [33, 1, 20]),
    'reuters_like': (300, 90, 60, 512, 512, 4, 'prior', True, 2, 0.1, [60, 41]),
    'heads8_d512': (200, 50, 40, 512, 512, 8, 'prior', True, 2, 0.15, [40, 33]),     # d_k = 64 (bibtex-like heads)
}


def build(cfg, dev, dropout=0.0, seed=0):
    from lamp_amd.Models import LAMP
    V, L, T, d, dff, h, mask, pos, B, p, lengths = cfg
    sd = R.make_state_dict(V, L, T, d, dff, h, 2, 2, pos_emb=pos, seed=seed)
    adj = R.make_adjacency(L, p, seed) if mask == 'prior' else None
    seq, spos = R.make_batch(B, V, T, lengths=lengths, seed=seed)
    m = LAMP(V, L, T, L, n_layers_enc=2, n_layers_dec=2, n_head=h, n_head2=h, d_word_vec=d, d_model=d,
             d_inner_hid=dff, d_k=d // h, d_v=d // h, encoder='graph', decoder='graph', dropout=dropout,
             dec_dropout=dropout, no_enc_pos_embedding=not pos,
             label_adj_matrix=adj.clone() if adj is not None else None, label_mask=mask, dec_dropout2=False)
    m.load_state_dict(sd)
    blocked = R.label_block_mask(adj, mask, L)
    tgt = (torch.rand(B, L, generator=torch.Generator().manual_seed(seed + 1)) < 0.2).float()
    return m.to(dev), sd, blocked, seq, spos, h, tgt


@pytest.mark.parametrize('name', sorted(CASES))
def test_every_parameter_gradient_matches_oracle_autograd(dev, name):
    m, sd, blocked, seq, spos, h, tgt = build(CASES[name], dev)
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ref_logits, ref_enc, _ = R.forward(sd64, seq, spos, h, blocked)
    ref_loss = F.binary_cross_entropy_with_logits(ref_logits, tgt.double())
    ref_loss.backward()

    m.train()
    logits, enc, extra = m((seq.to(dev), spos.to(dev)), None, None, tgt.to(dev))
    assert extra is None and logits.requires_grad and enc.requires_grad
    assert max_abs_diff(logits, ref_logits.detach()) < 1e-4
    assert max_abs_diff(enc, ref_enc.detach()) < 5e-5
    loss = F.binary_cross_entropy_with_logits(logits, tgt.to(dev))     # train.py:38
    loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-5

    checked = 0
    for pname, p in m.named_parameters():
        ref = sd64[pname].grad
        if pname == 'encoder.position_enc.weight':
            assert p.grad is None   # frozen sinusoid table: never handed to the optimizer (lamp/Models.py:97-107)
            continue
        if 'encoder.layer_stack' in pname and 'slf_attn' in pname:
            assert p.grad is None and ref is None, pname      # dead code in the reference: no gradient there either
            continue
        if pname == 'decoder.tgt_word_emb.weight' and 'tgt_word_proj.weight' in sd64 and sd64['tgt_word_proj.weight'].grad is not None:
            ref = ref + sd64['tgt_word_proj.weight'].grad
        assert p.grad is not None and ref is not None, pname
        scale = ref.abs().max().item()
        assert max_abs_diff(p.grad, ref) <= 3e-4 * scale + 1e-9, (pname, max_abs_diff(p.grad, ref), scale)
        checked += 1
    assert checked >= 40

    # eval-mode forward of the same weights agrees with the train-mode forward at dropout 0
    m.eval()
    with torch.no_grad():
        ev, _, _ = m((seq.to(dev), spos.to(dev)), None, None, None)
    assert max_abs_diff(ev, logits.detach()) < 2e-5


@pytest.mark.parametrize('case', ['reuters_like', 'tiny_none_h1', 'inveye_h8'])
@pytest.mark.parametrize('dropout', [0.0, 0.1])
def test_deferred_weight_gradients_equal_the_autograd_route(dev, dropout, case):
    """lamp_amd/training.py queues dW = dY^T.X of every projection and computes them in one grouped launch when the
    backward pass has run (train.py:40): same gradients as the per-layer route (up to the K-split's summation order),
    identical data gradients, `.grad` accumulation across two backward passes, and nothing deferred for non-leaf
    weights."""
    from lamp_amd import training
    m, sd, blocked, seq, spos, h, tgt = build(CASES[case], dev, dropout=dropout)
    m.train()

    def grads(defer, passes=1, composite=True, stale=0):
        training.DEFER_WEIGHT_GRADS, training.COMPOSITE_CALLS = defer, composite
        try:
            m.zero_grad(set_to_none=True)
            for _ in range(passes):
                torch.manual_seed(3)
                logits, enc, _ = m((seq.to(dev), spos.to(dev)), None, None, tgt.to(dev))
                F.binary_cross_entropy_with_logits(logits, tgt.to(dev)).backward()
                assert training._weight_grads.pending() == stale
            return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            training.DEFER_WEIGHT_GRADS = training.COMPOSITE_CALLS = True

    now, later, twice = grads(False), grads(True), grads(True, passes=2)
    # one library call per sub-layer and direction (lamp_ffn_bwd, lamp_mha_bwd, ...) issues the launches of the per-launch
    # route: every gradient bit for bit, with and without the queue
    for defer in (False, True):
        per_launch = grads(defer, composite=False)
        want = later if defer else now
        assert per_launch.keys() == want.keys()
        for n in want:
            if n == 'encoder.src_word_emb.weight':   # scatter-add of repeated tokens through atomics: order not fixed
                assert max_abs_diff(per_launch[n], want[n]) <= 1e-6 * want[n].abs().max().item()
                continue
            assert torch.equal(per_launch[n], want[n]), (n, defer)
    assert now.keys() == later.keys() == twice.keys() and len(now) >= 40
    n_weights = 0
    for n in now:
        scale = now[n].abs().max().item()
        assert max_abs_diff(now[n], later[n]) <= 2e-5 * scale + 1e-9, n
        assert max_abs_diff(twice[n], 2 * later[n]) <= 4e-6 * scale + 1e-9, n
        if n.endswith(('w_qs.weight', 'w_ks.weight', 'w_vs.weight', 'w_1.weight')) and 'encoder.layer_stack.0.slf' not in n:
            n_weights += 1
        else:
            continue
    assert n_weights >= 18
    # biases, LayerNorm parameters, embeddings never go through the queue: bitwise the same either way
    for n in now:
        if n.endswith('bias') or 'layer_norm' in n or 'tgt_word' in n:
            assert torch.equal(now[n], later[n]), n

    # a backward pass that raised never ran its callback and leaves its queue behind: later passes are not disturbed by it,
    # and it is dropped once MAX_TASKS newer passes have queued
    want = grads(True)
    training._weight_grads.tasks[-7] = ([('left over by a failed pass',)], [])
    again = grads(True, stale=1)
    for n in want:
        if n != 'encoder.src_word_emb.weight':
            assert torch.equal(want[n], again[n]), n
    for k in range(training._WeightGrads.MAX_TASKS):
        training._weight_grads.tasks[-6 + k] = ([('another',)], [])
    grads(True, stale=training._WeightGrads.MAX_TASKS - 1)
    assert -7 not in training._weight_grads.tasks
    training._weight_grads.tasks.clear()

    # a non-leaf weight (what nn.DataParallel's replicas hold) keeps the autograd route
    ffn = m.decoder.layer_stack[0].pos_ffn1
    assert training._deferrable(ffn.w_1.weight, ffn.w_2.weight) is not None
    assert training._deferrable(ffn.w_1.weight * 1.0, ffn.w_2.weight) is None


def test_optimizer_step_reduces_the_loss_and_invalidates_cached_query(dev):
    """A few Adam steps of the reference's train loop (train.py:34-48) on one batch."""
    m, sd, blocked, seq, spos, h, tgt = build(CASES['tiny_prior_h4'], dev)
    opt = torch.optim.Adam(m.get_trainable_parameters(), lr=2e-3)
    losses = []
    for _ in range(8):
        m.train()
        opt.zero_grad()
        pred, enc_output, *results = m((seq.to(dev), spos.to(dev)), None, None, tgt.to(dev))
        loss = F.binary_cross_entropy_with_logits(pred, tgt.to(dev), reduction='mean')
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.8 * losses[0], losses
    # eval after training sees the updated weights (the cached layer-0 query is keyed on parameter versions)
    m.eval()
    with torch.no_grad():
        ev, _, _ = m((seq.to(dev), spos.to(dev)), None, None, None)
    sd_now = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref, _, _ = R.forward(R.to_dtype(sd_now, torch.float64), seq, spos, h, blocked)
    assert max_abs_diff(ev, ref) < 1e-4


def test_dropout_forward_is_reproducible_under_manual_seed(dev):
    m, sd, blocked, seq, spos, h, tgt = build(CASES['tiny_prior_h4'], dev, dropout=0.1)
    m.train()
    outs = []
    for s in (7, 7, 8):
        torch.manual_seed(s)
        outs.append(m((seq.to(dev), spos.to(dev)), None, None, None)[0].detach())
    assert torch.equal(outs[0], outs[1])
    assert not torch.equal(outs[0], outs[2])
    m.eval()
    with torch.no_grad():
        ev = m((seq.to(dev), spos.to(dev)), None, None, None)[0]
    assert not torch.equal(ev, outs[0])


def test_ffn_with_dropout_matches_torch_restatement_with_the_same_mask(dev):
    from lamp_amd import _native as N
    from lamp_amd import training
    g = torch.Generator().manual_seed(1)
    M, d, dff, p, seed = 77, 64, 96, 0.3, 4242
    x = torch.randn(M, d, generator=g)
    w1, b1 = torch.randn(dff, d, 1, generator=g) * 0.1, torch.randn(dff, generator=g) * 0.1
    w2, b2 = torch.randn(d, dff, 1, generator=g) * 0.1, torch.randn(d, generator=g) * 0.1
    lg, lb = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    dy = torch.randn(M, d, generator=g)
    leaves = [t.double().requires_grad_() for t in (x, w1, b1, w2, b2, lg, lb)]
    X, W1, B1, W2, B2, LG, LB = leaves
    keep = N.dropout_keep_mask(M * d, p, seed).view(M, d)
    o = torch.relu(X @ W1[:, :, 0].t() + B1) @ W2[:, :, 0].t() + B2
    ref = F.layer_norm(o * keep / (1 - p) + X, (d,), LG, LB, 1e-5)
    ref.backward(dy.double())
    dl = [t.to(dev).requires_grad_() for t in (x, w1, b1, w2, b2, lg, lb)]
    y = training._FFNFn.apply(*dl, p, seed)
    y.backward(dy.to(dev))
    assert max_abs_diff(y.detach(), ref.detach()) < 2e-5
    for a, b in zip(dl, leaves):
        assert max_abs_diff(a.grad, b.grad) <= 2e-4 * b.grad.abs().max().item() + 1e-9


@pytest.mark.parametrize('H,p_attn,p_out', [(4, 0.0, 0.0), (4, 0.25, 0.2), (1, 0.25, 0.0)])
def test_mha_with_dropout_matches_torch_restatement_with_the_same_masks(dev, H, p_attn, p_out):
    from lamp_amd import _native as N
    from lamp_amd import training
    g = torch.Generator().manual_seed(2 + H)
    B, lq, lk, d, dk = 3, 21, 34, 64, 16
    xq, xkv = torch.randn(B, lq, d, generator=g), torch.randn(B, lk, d, generator=g)
    wq, wk, wv = (torch.randn(H * dk, d, generator=g) * 0.2 for _ in range(3))
    fc = torch.randn(d, H * dk, generator=g) * 0.2 if H > 1 else None
    lg, lb = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    blocked = torch.rand(lq, lk, generator=g) < 0.3
    blocked[:, 0] = False
    dy = torch.randn(B, lq, d, generator=g)
    s_attn, s_out = 99, 100
    if H == 1:
        d_out = dk  # without fc the attention output IS the sub-layer output: d_model = d_v
        xq, dy = torch.randn(B, lq, dk, generator=g), torch.randn(B, lq, dk, generator=g)
        xkv = torch.randn(B, lk, dk, generator=g)
        wq, wk, wv = (torch.randn(dk, dk, generator=g) * 0.2 for _ in range(3))
        lg, lb = 1 + 0.1 * torch.randn(dk, generator=g), 0.1 * torch.randn(dk, generator=g)
        d = dk
    tens = [xq, xkv, wq, wk, wv] + ([fc] if fc is not None else []) + [lg, lb]
    leaves = [t.double().requires_grad_() for t in tens]
    if fc is not None:
        XQ, XKV, WQ, WK, WV, FC, LG, LB = leaves
    else:
        (XQ, XKV, WQ, WK, WV, LG, LB), FC = leaves, None
    split = lambda t, l: t.view(B, l, H, dk).permute(2, 0, 1, 3)  # noqa: E731
    q, k, v = split(XQ @ WQ.t(), lq), split(XKV @ WK.t(), lk), split(XKV @ WV.t(), lk)
    s = (q @ k.transpose(-1, -2)) / dk ** 0.5
    P = torch.softmax(s.masked_fill(blocked, float('-inf')), -1)                     # (H, B, lq, lk)
    keep_a = N.dropout_keep_mask(H * B * lq * lk, p_attn, s_attn).view(H, B, lq, lk)
    a = ((P * keep_a / (1 - p_attn)) @ v).permute(1, 2, 0, 3).reshape(B, lq, H * dk)
    o = a @ FC.t() if FC is not None else a
    keep_o = N.dropout_keep_mask(B * lq * d, p_out, s_out).view(B, lq, d)
    ref = F.layer_norm(o * keep_o / (1 - p_out) + XQ, (d,), LG, LB, 1e-5)
    ref.backward(dy.double())

    dl = [t.to(dev).requires_grad_() for t in tens]
    mask, keepalive = N.make_mask(blocked.to(dev), B, lq, lk)
    args = dl[:5] + ([dl[5]] if fc is not None else [None]) + dl[-2:]
    y, Pm = training._MHAFn.apply(*args, H, mask, keepalive, p_attn, p_out, s_attn, s_out)
    y.backward(dy.to(dev))
    assert max_abs_diff(y.detach(), ref.detach()) < 3e-5
    assert max_abs_diff(Pm.view(H, B, lq, lk), (P * keep_a / (1 - p_attn)).detach()) < 1e-5   # the dropped map
    for a_, b_ in zip(dl, leaves):
        assert max_abs_diff(a_.grad, b_.grad) <= 3e-4 * b_.grad.abs().max().item() + 1e-9


def test_int_preds_training_matches_oracle_autograd(dev):
    """train.py:40-43 (-int_preds): BCE on every intermediate read-out, through a DETACHED copy of the projection."""
    m, sd, blocked, seq, spos, h, tgt = build(CASES['tiny_prior_h4'], dev)
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    w = sd64['tgt_word_proj.linear.weight']
    enc_o, _ = R.encoder_forward(sd64, seq, spos, h)
    y, _, _, int_outs = R.decoder_forward(sd64, seq, enc_o, blocked, h, None, False, True)
    ref_logits = R.readout(y, w)
    ref_int = [R.readout(o, w.detach()) for o in int_outs[:-1]]   # detached copy, lamp/Models.py:129
    loss_ref = F.binary_cross_entropy_with_logits(ref_logits, tgt.double())
    for ip in ref_int:
        loss_ref = loss_ref + 0.2 * F.binary_cross_entropy_with_logits(ip.reshape(tgt.shape), tgt.double())
    loss_ref.backward()
    m.train()
    pred, enc, ipreds = m((seq.to(dev), spos.to(dev)), None, None, tgt.to(dev), int_preds=True)
    assert len(ipreds) == len(ref_int) == 3
    loss = F.binary_cross_entropy_with_logits(pred, tgt.to(dev))
    for ip in ipreds:
        loss = loss + 0.2 * F.binary_cross_entropy_with_logits(ip, tgt.to(dev))
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-5
    for pname, p in m.named_parameters():
        ref = sd64[pname].grad
        if p.grad is None:
            continue
        scale = ref.abs().max().item()
        assert max_abs_diff(p.grad, ref) <= 3e-4 * scale + 1e-9, pname


def test_return_attns_in_training_mode(dev):
    """train.py:36 with -attns_loss: the maps come back in the reference's structure; with dropout they are the
    DROPPED maps the reference returns (lamp/SubLayers.py:40-43)."""
    m, sd, blocked, seq, spos, h, tgt = build(CASES['tiny_prior_h4'], dev, dropout=0.0)
    m.train()
    pred, enc, enc_attns, dec_attns = m((seq.to(dev), spos.to(dev)), None, None, None, return_attns=True)
    ref = R.forward(sd, seq, spos, h, blocked, return_attns=True)
    assert max_abs_diff(pred.detach(), ref[0]) < 1e-4
    for got, want in zip(enc_attns[0], ref[2][0]):
        assert max_abs_diff(got, want) < 1e-5
    for k in (0, 1):
        for got, want in zip(dec_attns[k], ref[3][k]):
            assert max_abs_diff(got, want) < 1e-5
    pred.sum().backward()   # graph is intact with the maps attached
    m2 = build(CASES['tiny_prior_h4'], dev, dropout=0.5)[0]
    m2.train()
    _, _, _, dec_attns2 = m2((seq.to(dev), spos.to(dev)), None, None, None, return_attns=True)
    frac_zero = (dec_attns2[1][0] == 0).float().mean().item()
    assert 0.3 < frac_zero < 0.9     # about half the unblocked entries are dropped, the rest scaled by 2


def test_full_size_reuters_step_with_dropout(dev):
    """BASELINE's reuters shape (V=23666, L=90, T=302, d=512, 4 heads, prior mask), dropout 0.1: every trainable
    parameter but the dead encoder self-attention gets a finite gradient, the step is reproducible under
    torch.manual_seed, and bias / LayerNorm gradients are deterministic (fixed-order reductions)."""
    cfg = (23666, 90, 302, 512, 512, 4, 'prior', True, 8, 0.1, None)
    m, sd, blocked, seq, spos, h, tgt = build(cfg, dev, dropout=0.1)
    m.train()

    def run(seed):
        torch.manual_seed(seed)
        m.zero_grad(set_to_none=True)
        pred, enc, _ = m((seq.to(dev), spos.to(dev)), None, None, tgt.to(dev))
        loss = F.binary_cross_entropy_with_logits(pred, tgt.to(dev))
        loss.backward()
        return loss.item(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    l1, g1 = run(3)
    l2, g2 = run(3)
    l3, _ = run(4)
    assert l1 == l2 and l1 != l3
    trainable = {n for n, p in m.named_parameters()
                 if not ('encoder.layer_stack' in n and 'slf_attn' in n) and n != 'encoder.position_enc.weight'}
    assert set(g1) == trainable
    for n in trainable:
        assert torch.isfinite(g1[n]).all(), n
        if n != 'encoder.src_word_emb.weight':      # the embedding scatter-add uses atomics (order may vary)
            assert torch.equal(g1[n], g2[n]), n
        else:
            assert max_abs_diff(g1[n], g2[n]) < 1e-6
    assert (g1['encoder.src_word_emb.weight'][0] == 0).all()   # PAD row


def test_gradients_with_wide_heads(dev):
    """d_k = 160 > 128: the training path through the general attention (scores through memory)."""
    cfg = (60, 23, 19, 320, 256, 2, 'prior', True, 2, 0.2, [19, 8])
    m, sd, blocked, seq, spos, h, tgt = build(cfg, dev)
    sd64 = {k: v.double().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ref_logits, _, _ = R.forward(sd64, seq, spos, h, blocked)
    F.binary_cross_entropy_with_logits(ref_logits, tgt.double()).backward()
    m.train()
    logits, _, _ = m((seq.to(dev), spos.to(dev)), None, None, tgt.to(dev))
    F.binary_cross_entropy_with_logits(logits, tgt.to(dev)).backward()
    assert max_abs_diff(logits, ref_logits.detach()) < 1e-4
    for pname, p in m.named_parameters():
        if p.grad is None:
            continue
        ref = sd64[pname].grad
        assert max_abs_diff(p.grad, ref) <= 3e-4 * ref.abs().max().item() + 1e-9, pname


@pytest.mark.parametrize('H', [1, 4])
def test_mha_with_distinct_key_and_value_sources(dev, H):
    """lamp/SubLayers.py:77-93 projects k and v independently; no layer of the reference passes different tensors, the
    module surface allows it.  Eval forward and every training gradient against the fp64 oracle."""
    from lamp_amd.SubLayers import MultiHeadAttention
    g = torch.Generator().manual_seed(7 + H)
    B, lq, lk, d = 3, 21, 34, 64
    mod = MultiHeadAttention(H, d, d // H, d // H, dropout=0.0).to(dev)
    xq, xk, xv = (torch.randn(B, l, d, generator=g) for l in (lq, lk, lk))
    mask = torch.rand(B, lq, lk, generator=g) < 0.3
    mask[:, :, 0] = False
    w = {k: v.detach().cpu().double().requires_grad_() for k, v in mod.state_dict().items()}
    xq64, xk64, xv64 = (t.double().requires_grad_() for t in (xq, xk, xv))
    ref, ref_attn = R.mha(xq64, xk64, mask, w['w_qs.weight'], w['w_ks.weight'], w['w_vs.weight'], w.get('fc.weight'),
                          w['layer_norm.weight'], w['layer_norm.bias'], H, xv=xv64)
    mod.eval()
    out, attn = mod(xq.to(dev), xk.to(dev), xv.to(dev), attn_mask=mask.to(dev))
    assert max_abs_diff(out, ref.detach()) < 2e-5 and max_abs_diff(attn, ref_attn.detach()) < 5e-6
    mod.train()
    xs = [t.to(dev).requires_grad_() for t in (xq, xk, xv)]
    out_t, _ = mod(*xs, attn_mask=mask.to(dev))
    assert max_abs_diff(out_t, ref.detach()) < 2e-5
    dy = torch.randn(B, lq, d, generator=g)
    out_t.backward(dy.to(dev))
    ref.backward(dy.double())
    for got, want in zip(xs, (xq64, xk64, xv64)):
        assert max_abs_diff(got.grad, want.grad) < 3e-4 * max(1.0, want.grad.abs().max().item())
    for n, p in mod.named_parameters():
        assert max_abs_diff(p.grad, w[n].grad) < 3e-4 * max(1.0, w[n].grad.abs().max().item()), n


def test_bare_wrappers_record_autograd_in_training_mode(dev):
    """XavierLinear and ScaledDotProductAttention called on their own in train() (lamp/SubLayers.py:7-43 are plain
    autograd modules in the reference): forward and gradients against fp64 torch."""
    from lamp_amd.SubLayers import ScaledDotProductAttention, XavierLinear
    g = torch.Generator().manual_seed(11)
    lin = XavierLinear(48, 30).to(dev).train()
    x = torch.randn(5, 9, 48, generator=g)
    xd = x.to(dev).requires_grad_()
    y = lin(xd)
    dy = torch.randn(5, 9, 30, generator=g)
    y.backward(dy.to(dev))
    w64, b64, x64 = (t.detach().cpu().double().requires_grad_() for t in (lin.linear.weight, lin.linear.bias, x))
    ref = F.linear(x64, w64, b64)
    ref.backward(dy.double())
    assert max_abs_diff(y, ref.detach()) < 2e-5 and max_abs_diff(xd.grad, x64.grad) < 1e-4
    assert max_abs_diff(lin.linear.weight.grad, w64.grad) < 1e-4 and max_abs_diff(lin.linear.bias.grad, b64.grad) < 1e-4

    att = ScaledDotProductAttention(temperature=8.0, dropout=0.0).to(dev).train()
    n, lq, lk, dk = 6, 17, 29, 64
    q, k, v = (torch.randn(n, l, dk, generator=g) for l in (lq, lk, lk))
    mask = torch.rand(n, lq, lk, generator=g) < 0.25
    mask[:, :, 0] = False
    qd, kd, vd = (t.to(dev).requires_grad_() for t in (q, k, v))
    out, attn = att(qd, kd, vd, attn_mask=mask.to(dev))
    do = torch.randn(n, lq, dk, generator=g)
    out.backward(do.to(dev))
    q64, k64, v64 = (t.double().requires_grad_() for t in (q, k, v))
    ref_o, ref_a = R.sdpa(q64, k64, v64, mask, 8.0)
    ref_o.backward(do.double())
    assert max_abs_diff(out, ref_o.detach()) < 2e-5 and max_abs_diff(attn, ref_a.detach()) < 5e-6
    for got, want in ((qd, q64), (kd, k64), (vd, v64)):
        assert max_abs_diff(got.grad, want.grad) < 1e-4
    # attention dropout: the returned map is the dropped one, and the output is its product with v
    att_p = ScaledDotProductAttention(temperature=8.0, dropout=0.3).to(dev).train()
    torch.manual_seed(5)
    out_p, attn_p = att_p(q.to(dev), k.to(dev), v.to(dev), attn_mask=mask.to(dev))
    assert max_abs_diff(out_p, attn_p.double().cpu() @ v.double()) < 2e-5
    kept = attn_p != 0
    assert 0.6 < kept.float().mean().item() / (ref_a > 0).float().mean().item() < 0.8
