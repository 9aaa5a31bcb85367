"""CPU oracle for the recogniser head that follows TPS++: NRTR encoder, NRTR decoder (teacher-forced
and greedy), the attention label convertor and the recogniser's test-time composition.

TEST INFRASTRUCTURE, NOT PRODUCT: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.  ``tps_pp_amd`` never imports it.

Parity pin: outputs of the reference itself (encoders/nrtr_encoder.py, decoders/nrtr_decoder.py,
convertors/attn.py executed from /root/reference by tests/golden/make_golden.py) on the inputs of
tests/golden/cases.py, committed as tests/golden/nrtr_{encoder,decoder,head_full}.npz and replayed
by tests/test_oracle_golden.py.

Functional restatement over a plain state_dict (reference key names), PyTorch CPU functional ops in
the reference's order of composition.  Paths below are relative to /root/reference/mmocr/models/.
"""
import math

import torch
import torch.nn.functional as F


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(a)


def multi_head_attention(sd, prefix, q, k, v, n_head, mask=None):
    """MultiHeadAttention.forward + ScaledDotProductAttention.forward
    (common/modules/transformer_module.py:24-33,75-99), dropout = identity (eval).
    mask: None | (N, Tk) key mask | (N, Tq, Tk); zeros are filled with -inf before the softmax."""
    n, lq, dm = q.shape
    lk = k.shape[1]
    dk = dm // n_head

    def lin(name, x):
        return F.linear(x, sd[f"{prefix}.{name}.weight"], sd.get(f"{prefix}.{name}.bias"))
    qh = lin("linear_q", q).view(n, lq, n_head, dk).transpose(1, 2)
    kh = lin("linear_k", k).view(n, lk, n_head, dk).transpose(1, 2)
    vh = lin("linear_v", v).view(n, lk, n_head, dk).transpose(1, 2)
    attn = torch.matmul(qh / (dk ** 0.5), kh.transpose(2, 3))
    if mask is not None:
        m = mask.unsqueeze(1) if mask.dim() == 3 else mask.unsqueeze(1).unsqueeze(1)
        attn = attn.masked_fill(m == 0, float("-inf"))
    out = torch.matmul(F.softmax(attn, dim=-1), vh)
    out = out.transpose(1, 2).contiguous().view(n, lq, dm)
    return lin("fc", out)


def feed_forward(sd, prefix, x):
    """PositionwiseFeedForward.forward (transformer_module.py:119-125), act = mmcv.GELU = erf GELU."""
    h = F.gelu(F.linear(x, sd[prefix + ".w_1.weight"], sd[prefix + ".w_1.bias"]))
    return F.linear(h, sd[prefix + ".w_2.weight"], sd[prefix + ".w_2.bias"])


def layer_norm(sd, prefix, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def valid_mask(n, t, valid_ratios):
    """NRTREncoder._get_mask / NRTRDecoder._get_mask (textrecog/encoders/nrtr_encoder.py:51-65,
    decoders/nrtr_decoder.py:115-129): the first ceil(T * ratio) tokens are valid."""
    if valid_ratios is None:
        return None
    mask = torch.zeros((n, t))
    for i, r in enumerate(valid_ratios):
        mask[i, :min(t, math.ceil(t * r))] = 1
    return mask


def n_layers_of(sd, prefix="layer_stack."):
    return 1 + max(int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix))


def encoder_forward(sd, feat, n_head=8, valid_ratios=None):
    """NRTREncoder.forward (textrecog/encoders/nrtr_encoder.py:67-87) with TFEncoderLayer in its
    default ('norm','self_attn','norm','ffn') order (common/layers/transformer_layers.py:67-75).
    feat (N, C, H, W) -> (N, H*W, C)."""
    feat = _t(feat)
    n, c, h, w = feat.shape
    x = feat.view(n, c, h * w).permute(0, 2, 1).contiguous()
    mask = valid_mask(n, h * w, valid_ratios)
    for i in range(n_layers_of(sd)):
        p = f"layer_stack.{i}"
        y = layer_norm(sd, p + ".norm1", x)
        x = x + multi_head_attention(sd, p + ".attn", y, y, y, n_head, mask)
        y = layer_norm(sd, p + ".norm2", x)
        x = x + feed_forward(sd, p + ".mlp", y)
    return layer_norm(sd, "layer_norm", x)


def decoder_attention(sd, trg_seq, src, n_head, padding_idx, src_mask=None):
    """NRTRDecoder._attention (textrecog/decoders/nrtr_decoder.py:95-113): embedding + sinusoid
    position table, target mask = pad mask & causal mask, TFDecoderLayer in its default
    ('norm','self_attn','norm','enc_dec_attn','norm','ffn') order (transformer_layers.py:150-163),
    final LayerNorm eps 1e-6."""
    emb = F.embedding(trg_seq, sd["trg_word_emb.weight"])
    x = emb + sd["position_enc.position_table"][:, :trg_seq.shape[1]]
    ls = trg_seq.shape[1]
    causal = (1 - torch.triu(torch.ones((ls, ls)), diagonal=1)).unsqueeze(0).bool()
    trg_mask = (trg_seq != padding_idx).unsqueeze(-2) & causal
    for i in range(n_layers_of(sd)):
        p = f"layer_stack.{i}"
        y = layer_norm(sd, p + ".norm1", x)
        x = x + multi_head_attention(sd, p + ".self_attn", y, y, y, n_head, trg_mask)
        y = layer_norm(sd, p + ".norm2", x)
        x = x + multi_head_attention(sd, p + ".enc_attn", y, src, src, n_head, src_mask)
        y = layer_norm(sd, p + ".norm3", x)
        x = x + feed_forward(sd, p + ".mlp", y)
    return layer_norm(sd, "layer_norm", x, eps=1e-6)


def decoder_forward_train(sd, out_enc, padded_targets, n_head=8, padding_idx=92, valid_ratios=None):
    """NRTRDecoder.forward_train (nrtr_decoder.py:131-152): raw logits (N, T, C-1)."""
    out_enc = _t(out_enc)
    src_mask = valid_mask(out_enc.shape[0], out_enc.shape[1], valid_ratios)
    x = decoder_attention(sd, _t(padded_targets).long(), out_enc, n_head, padding_idx, src_mask)
    return F.linear(x, sd["classifier.weight"], sd["classifier.bias"])


def decoder_forward_test(sd, out_enc, n_head=8, max_seq_len=40, start_idx=91, padding_idx=92,
                         valid_ratios=None):
    """NRTRDecoder.forward_test (nrtr_decoder.py:154-177): greedy decoding, the whole padded
    sequence is re-run every step (as the reference does); returns softmax scores (N, T, C-1)."""
    out_enc = _t(out_enc)
    n = out_enc.shape[0]
    src_mask = valid_mask(n, out_enc.shape[1], valid_ratios)
    seq = torch.full((n, max_seq_len + 1), padding_idx, dtype=torch.long)
    seq[:, 0] = start_idx
    outs = []
    for step in range(max_seq_len):
        x = decoder_attention(sd, seq, out_enc, n_head, padding_idx, src_mask)
        prob = F.softmax(F.linear(x[:, step, :], sd["classifier.weight"], sd["classifier.bias"]), dim=-1)
        outs.append(prob)
        seq[:, step + 1] = torch.max(prob, dim=-1)[1]
    return torch.stack(outs, dim=1)


def sinusoid_table(n_position=200, d_hid=512):
    """PositionalEncoding._get_sinusoid_encoding_table (transformer_module.py:141-153): float64
    powers rounded to fp32, fp32 product with the position, sin on even / cos on odd columns."""
    import numpy as np
    den = torch.Tensor([1.0 / np.power(10000, 2 * (j // 2) / d_hid) for j in range(d_hid)]).view(1, -1)
    tab = torch.arange(n_position).unsqueeze(-1).float() * den
    tab[:, 0::2] = torch.sin(tab[:, 0::2])
    tab[:, 1::2] = torch.cos(tab[:, 1::2])
    return tab.unsqueeze(0)


# ---- label convertor (textrecog/convertors/base.py:20-24,28-46 + attn.py:47-74) -------------------
DICT36 = tuple("0123456789abcdefghijklmnopqrstuvwxyz")
DICT90 = tuple("0123456789abcdefghijklmnopqrstuvwxyz"
               "ABCDEFGHIJKLMNOPQRSTUVWXYZ!\"#$%&'()"
               "*+,-./:;<=>?@[\\]_`~")


def attn_dictionary(dict_type="DICT90", with_unknown=True, start_end_same=True):
    """-> (idx2char, unknown_idx, start_idx, end_idx, padding_idx)."""
    idx2char = list(DICT36 if dict_type == "DICT36" else DICT90)
    unknown_idx = None
    if with_unknown:
        idx2char.append("<UKN>")
        unknown_idx = len(idx2char) - 1
    idx2char.append("<BOS/EOS>")
    start_idx = len(idx2char) - 1
    if not start_end_same:
        idx2char.append("<BOS/EOS>")
    end_idx = len(idx2char) - 1
    idx2char.append("<PAD>")
    return idx2char, unknown_idx, start_idx, end_idx, len(idx2char) - 1


def tensor2idx(outputs, end_idx=91, padding_idx=92):
    """AttnConvertor.tensor2idx (attn.py:112-143): per-step arg-max; <PAD> skipped, stop at <EOS>."""
    outputs = _t(outputs)
    indexes, scores = [], []
    for b in range(outputs.shape[0]):
        mv, mi = torch.max(outputs[b], -1)
        si, ss = [], []
        for ci, cs in zip(mi.tolist(), mv.tolist()):
            if ci == padding_idx:
                continue
            if ci == end_idx:
                break
            si.append(ci)
            ss.append(cs)
        indexes.append(si)
        scores.append(ss)
    return indexes, scores


def idx2str(indexes, idx2char):
    """BaseConvertor.idx2str (base.py:87-103)."""
    return ["".join(idx2char[i] for i in idx) for idx in indexes]


def head_simple_test(enc_sd, dec_sd, feat, n_head=8, max_seq_len=40, valid_ratios=None):
    """The part of EncodeDecodeRecognizer.simple_test after the backbone
    (textrecog/recognizer/encode_decode_recognizer.py:196-221) with AttnConvertor(DICT90, unknown)."""
    idx2char, _, start_idx, end_idx, padding_idx = attn_dictionary()
    out_enc = encoder_forward(enc_sd, feat, n_head, valid_ratios)
    out_dec = decoder_forward_test(dec_sd, out_enc, n_head, max_seq_len, start_idx, padding_idx, valid_ratios)
    indexes, scores = tensor2idx(out_dec, end_idx, padding_idx)
    return dict(out_enc=out_enc, out_dec=out_dec, indexes=indexes, scores=scores,
                text=idx2str(indexes, idx2char))
