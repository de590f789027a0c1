"""LAMP model facade (reference: lamp/Models.py:18-137) for encoder='graph', decoder='graph'.

``forward`` in eval mode on a HIP device is ONE call into liblamp_hip.so (``lamp_forward``): about
thirty kernel launches for a 2+2-layer model instead of the reference's ~140 ATen ops, no
per-forward mask materialisation, no head split/merge copies, and the discarded encoder
self-attention (lamp/Layers.py:16-18) is simply not computed unless its maps are requested.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _native as N
from .Decoders import GraphDecoder, MLPDecoder, RNNDecoder
from .Encoders import GraphEncoder, MLPEncoder, RNNEncoder
from .SubLayers import XavierLinear


class LAMP(nn.Module):
    def __init__(self, n_src_vocab, n_tgt_vocab, n_max_seq_e, n_max_seq_d, n_layers_enc=6, n_layers_dec=6,
                 n_head=8, n_head2=8, d_word_vec=512, d_model=512, d_inner_hid=1024, d_k=64, d_v=64,
                 dropout=0.1, dec_dropout=0.1, dec_dropout2=0.1, proj_share_weight=True,
                 embs_share_weight=True, encoder='selfatt', decoder='sa_m', enc_transform='', onehot=False,
                 no_enc_pos_embedding=False, no_dec_self_att=False, loss='ce', label_adj_matrix=None,
                 label_mask=None, matching_mlp=False, graph_conv=False, attn_type='softmax', int_preds=False):
        super().__init__()
        if d_model != d_word_vec:
            raise ValueError('d_model must equal d_word_vec (residual connections)')
        self.decoder_type = decoder
        self.onehot = onehot
        self.loss = loss
        self.enc_vec = encoder == 'mlp' or enc_transform != ''

        if encoder == 'graph':
            self.encoder = GraphEncoder(
                n_src_vocab, n_max_seq_e, n_layers=n_layers_enc, n_head=n_head, d_word_vec=d_word_vec,
                d_model=d_model, d_k=d_k, d_v=d_v, d_inner_hid=d_inner_hid, onehot=onehot, dropout=dropout,
                no_enc_pos_embedding=no_enc_pos_embedding, enc_transform=enc_transform)
        elif encoder == 'mlp':
            self.encoder = MLPEncoder()
        elif encoder == 'rnn':
            self.encoder = RNNEncoder()
        else:
            raise NotImplementedError(encoder)

        if decoder == 'graph':
            self.decoder = GraphDecoder(
                n_tgt_vocab, n_max_seq_d, n_layers=n_layers_dec, n_head=n_head, n_head2=n_head2,
                d_word_vec=d_word_vec, d_model=d_model, d_k=d_k, d_v=d_v, d_inner_hid=d_inner_hid,
                dropout=dec_dropout, dropout2=dec_dropout2, no_dec_self_att=no_dec_self_att,
                label_adj_matrix=label_adj_matrix, label_mask=label_mask, enc_vec=self.enc_vec,
                graph_conv=graph_conv, attn_type=attn_type)
        elif decoder == 'mlp':
            self.decoder = MLPDecoder()
        elif decoder == 'rnn_m':
            self.decoder = RNNDecoder()
        else:
            raise NotImplementedError(decoder)

        # Read-out.  Assigning the embedding Parameter to `tgt_word_proj.weight` registers a second
        # key on the XavierLinear wrapper but does NOT tie `tgt_word_proj.linear.weight`, which stays
        # the matrix the forward uses -- exactly the reference's (accidental) behaviour; both keys
        # are needed for checkpoint compatibility (lamp/Models.py:87-94).
        bias = self.decoder_type in ('mlp', 'graph', 'star') and not proj_share_weight
        if proj_share_weight:
            self.tgt_word_proj = XavierLinear(d_model, n_tgt_vocab, bias=bias)
            self.tgt_word_proj.weight = self.decoder.tgt_word_emb.weight
        else:
            self.tgt_word_proj = XavierLinear(d_model, 1, bias=bias)
        if int_preds:
            self.tgt_word_proj_copy = XavierLinear(d_model, n_tgt_vocab, bias=bias)

        self.d_model, self.d_inner, self.d_k, self.d_v = d_model, d_inner_hid, d_k, d_v
        self.n_labels = n_tgt_vocab
        self._native_cache = None
        self._param_list = None

    def get_trainable_parameters(self):
        """Everything but the frozen sinusoid table (reference: lamp/Models.py:97-107)."""
        frozen = set()
        if hasattr(self.encoder, 'position_enc'):
            frozen |= {id(p) for p in self.encoder.position_enc.parameters()}
        return (p for p in self.parameters() if id(p) not in frozen)

    # ------------------------------------------------------------------ native model descriptor
    def _native_model(self):
        """Build (and cache, keyed on every parameter's data_ptr) the lamp_model struct."""
        # the Parameter objects are stable (load_state_dict / .to() replace their data, not them); a DataParallel
        # replica is a shallow copy whose parameters ARE different tensors -- detected by the first one
        params = self._param_list
        if params is None or params[0] is not next(self.parameters()):
            params = self._param_list = [p for p in self.parameters()]
        mask = self.decoder.label_mask_u8
        tiles = self.decoder.label_tiles
        bits = self.decoder.label_mask_bits
        key = tuple(p.data_ptr() for p in params) + (N.ptr(mask), N.ptr(bits), N.ptr(tiles), self.use_label_tiles,
                                                      self.cache_layer0_query, self.use_mask_bits)
        if self.cache_layer0_query:  # the hoisted projection below is stale once either operand changes
            l0 = self.decoder.layer_stack[0].enc_attn
            key += (self.decoder.tgt_word_emb.weight._version, l0.w_qs.weight._version)
        fuse = bool(self.fuse_layernorm) and all(
            l.enc_attn.n_head > 1 and (not hasattr(l, 'slf_attn') or l.slf_attn.n_head > 1) for l in self.decoder.layer_stack)
        key += (fuse,)
        if fuse:  # folded weights depend on (almost) every parameter: any in-place update invalidates them
            key += tuple(p._version for p in params)
        if self._native_cache is not None and self._native_cache[0] == key:
            return self._native_cache[1]
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError('lamp_amd expects contiguous fp32 parameters')
        N.require_device(*params)
        enc, dec = self.encoder, self.decoder
        enc_arr = (N.EncLayer * max(1, len(enc.layer_stack)))()
        for i, l in enumerate(enc.layer_stack):
            enc_arr[i] = N.EncLayer(N.mha_weights(l.slf_attn), N.ffn_weights(l.pos_ffn))
        dec_arr = (N.DecLayer * max(1, len(dec.layer_stack)))()
        for i, l in enumerate(dec.layer_stack):
            slf = N.mha_weights(l.slf_attn) if hasattr(l, 'slf_attn') else N.MhaWeights()
            dec_arr[i] = N.DecLayer(N.mha_weights(l.enc_attn), N.ffn_weights(l.pos_ffn1), slf,
                                    N.ffn_weights(l.pos_ffn2))
        pos = enc.position_enc.weight if hasattr(enc, 'position_enc') else None
        w_out = self.tgt_word_proj.linear.weight
        if w_out.size(0) != self.n_labels:
            raise NotImplementedError('proj_share_weight=False read-out is not on the graph path')
        m = N.Model(enc.src_word_emb.weight.size(0), pos.size(0) if pos is not None else 0, self.n_labels,
                    self.d_model, self.d_inner, self.d_k, self.d_v, len(enc.layer_stack), len(dec.layer_stack),
                    0, N.ptr(enc.src_word_emb.weight), N.ptr(pos), N.ptr(dec.tgt_word_emb.weight),
                    N.ptr(w_out), N.ptr(mask), N.ptr(bits) if self.use_mask_bits else 0,
                    N.ptr(tiles) if self.use_label_tiles else 0, enc_arr, dec_arr, 0)
        folded = None
        if fuse:
            folded = self._fold_layernorms(enc, dec)
            m.fused_ln = C.pointer(folded[0])
        q0 = None
        if self.cache_layer0_query and len(dec.layer_stack) > 0:
            # decoder layer 0's query = label table x W_q: weights only, so it is projected here once per
            # weight version (lamp_linear_fwd) instead of on every forward
            q0 = N.linear(dec.tgt_word_emb.weight.detach(), dec.layer_stack[0].enc_attn.w_qs.weight.detach())
            m.dec0_query = q0.data_ptr()
            # forwards may be issued from several streams (evaluate.test_epoch(streams=2)); the cached
            # projection must be complete before any of them reads it -- a one-off sync per weight version
            torch.cuda.current_stream().synchronize()
        self._native_cache = (key, (m, enc_arr, dec_arr, (q0, folded)))
        return self._native_cache[1]

    def _fold_layernorms(self, enc, dec):
        """Deferred LayerNorm (include/lamp_hip.h: lamp_fused_ln): fold each elidable LayerNorm's gamma / beta into the
        linear maps that consume its output (lamp_layernorm_fold, weights only).  -> (FusedLn struct, keepalive)."""
        keep = []

        def fold(lin_w, ln, bias=None):
            w = lin_w.detach()
            w = w.view(w.size(0), w.size(1))
            wf, s, bf = N.layernorm_fold(w, ln.weight.detach(), ln.bias.detach(), bias.detach() if bias is not None else None)
            keep.extend((wf, s, bf))
            return N.FoldedLinear(wf.data_ptr(), s.data_ptr(), bf.data_ptr())
        enc_arr = (N.FusedLnEncLayer * max(1, len(enc.layer_stack)))()
        for i, l in enumerate(enc.layer_stack):
            if i > 0:
                prev = enc.layer_stack[i - 1].pos_ffn
                enc_arr[i] = N.FusedLnEncLayer(fold(l.pos_ffn.w_1.weight, prev.layer_norm, l.pos_ffn.w_1.bias))
        dec_arr = (N.FusedLnDecLayer * max(1, len(dec.layer_stack)))()
        for i, l in enumerate(dec.layer_stack):
            e = N.FusedLnDecLayer()
            if i > 0:
                e.enc_q = fold(l.enc_attn.w_qs.weight, dec.layer_stack[i - 1].pos_ffn2.layer_norm)
                e.ffn1_w1 = fold(l.pos_ffn1.w_1.weight, l.enc_attn.layer_norm, l.pos_ffn1.w_1.bias)
            if hasattr(l, 'slf_attn'):
                e.slf_q = fold(l.slf_attn.w_qs.weight, l.pos_ffn1.layer_norm)
                e.slf_k = fold(l.slf_attn.w_ks.weight, l.pos_ffn1.layer_norm)
                e.slf_v = fold(l.slf_attn.w_vs.weight, l.pos_ffn1.layer_norm)
                e.ffn2_w1 = fold(l.pos_ffn2.w_1.weight, l.slf_attn.layer_norm, l.pos_ffn2.w_1.bias)
            else:
                e.ffn2_w1 = fold(l.pos_ffn2.w_1.weight, l.pos_ffn1.layer_norm, l.pos_ffn2.w_1.bias)
            dec_arr[i] = e
        fused = N.FusedLn(enc_arr, dec_arr)
        keep.extend((enc_arr, dec_arr))
        return fused, keep

    def forward(self, src, adj, tgt_seq, binary_tgt, return_attns=False, int_preds=False):
        if self.decoder_type != 'graph':
            raise NotImplementedError(self.decoder_type)
        if adj:
            raise NotImplementedError('per-sample adjacency for the encoder is outside the hot path')
        src_seq, src_pos = src
        N.require_device(src_seq, src_pos)
        if self.training:
            # train.py:36: the autograd-recording path (HIP kernels forward and backward, lamp_amd/training.py)
            from . import training
            return training.forward_train(self, src_seq, src_pos, return_attns=bool(return_attns and not int_preds),
                                          int_preds=bool(int_preds))
        dev = src_seq.device
        seq = src_seq.long().contiguous()
        pos = src_pos.long().contiguous()
        B, T = seq.shape
        L, d = self.n_labels, self.d_model
        model, enc_arr, dec_arr, _q0 = self._native_model()
        Ne, Nd = model.n_layers_enc, model.n_layers_dec

        logits = torch.empty((B, L), dtype=torch.float32, device=dev)
        enc_output = torch.empty((B, T, d), dtype=torch.float32, device=dev)

        aux, keep = None, []
        want_attn = bool(return_attns and not int_preds)
        enc_attns = slf_attns = encdec_attns = ipreds = None
        if int_preds:
            n_int = sum(2 if hasattr(l, 'slf_attn') else 1 for l in self.decoder.layer_stack) - 1
            ipreds = [torch.empty((B, L), dtype=torch.float32, device=dev) for _ in range(n_int)]
            arr = (C.c_void_p * max(1, n_int))(*[t.data_ptr() for t in ipreds])
            keep.append(arr)
            aux = N.Aux(None, None, None, arr, n_int, 0)
        elif want_attn:
            def alloc(h, lq, lk):
                return torch.empty((h * B, lq, lk), dtype=torch.float32, device=dev)
            enc_attns = [alloc(l.slf_attn.n_head, T, T) for l in self.encoder.layer_stack]
            slf_attns = [alloc(l.slf_attn.n_head, L, L) if hasattr(l, 'slf_attn') else None
                         for l in self.decoder.layer_stack]
            encdec_attns = [alloc(l.enc_attn.n_head, L, T) for l in self.decoder.layer_stack]
            a0 = (C.c_void_p * max(1, Ne))(*[t.data_ptr() for t in enc_attns])
            a1 = (C.c_void_p * max(1, Nd))(*[N.ptr(t) for t in slf_attns])
            a2 = (C.c_void_p * max(1, Nd))(*[t.data_ptr() for t in encdec_attns])
            keep += [a0, a1, a2]
            aux = N.Aux(a0, a1, a2, None, 0, 0)

        lib = N.lib()
        per_sample = lib.lamp_forward_workspace_bytes(C.byref(model), 1, T, int(want_attn))
        fixed = 2 * per_sample - lib.lamp_forward_workspace_bytes(C.byref(model), 2, T, int(want_attn))
        # enough for the whole batch in one pass unless that exceeds the cap (then it micro-batches)
        whole = fixed + (per_sample - fixed) * B + 4096
        budget = max(per_sample + 4096, min(whole, self.workspace_limit_bytes))
        ws = N.workspace(budget, dev)
        N.check(lib.lamp_forward(C.byref(model), seq.data_ptr(), pos.data_ptr(), B, T, logits.data_ptr(),
                                 enc_output.data_ptr(), C.byref(aux) if aux is not None else None,
                                 ws.data_ptr(), ws.numel(), N.stream()), 'lamp_forward')
        del keep
        if int_preds:
            return logits, enc_output, ipreds
        if want_attn:
            return logits, enc_output, [enc_attns], [slf_attns, encdec_attns]
        return logits, enc_output, None

    # Upper bound on the scratch a forward may claim; larger batches are processed in micro-batches
    # inside lamp_forward.  8 GiB of 288 GB keeps even the 4096-label configuration at >= 64 samples.
    workspace_limit_bytes = 8 << 30
    # Hoist decoder layer 0's (weights-only) query projection out of the per-batch path.
    cache_layer0_query = True
    # Skip fully blocked 32x32 tiles of the label graph in the label->label attention.
    use_label_tiles = True
    # Read the label mask bit-packed (one 32-bit word per 32-key tile and row) instead of as bytes.
    use_mask_bits = True
    # Deferred LayerNorm: skip the LayerNorm launches whose output only feeds the next sub-layer's linear maps
    # (include/lamp_hip.h: lamp_fused_ln).  Logits agree with the plain path to fp32 rounding, not bitwise.
    fuse_layernorm = False
