"""The recogniser head that consumes the rectified features: NRTR encoder, NRTR decoder, the attention
label convertor and the encode-decode recogniser, behind the reference's API.

Mirrors (mmocr/models/): `common/modules/transformer_module.py:36-163` (MultiHeadAttention,
PositionwiseFeedForward, PositionalEncoding), `common/layers/transformer_layers.py:9-163`
(TFEncoderLayer, TFDecoderLayer), `textrecog/encoders/nrtr_encoder.py:12-87`,
`textrecog/decoders/nrtr_decoder.py:14-177`, `textrecog/convertors/{base,attn}.py`,
`textrecog/recognizer/encode_decode_recognizer.py:15-221`: same constructor arguments, the same
`state_dict` keys (released checkpoints load), the same call contracts and return values.

The sub-modules below only HOLD parameters (so `state_dict` matches); the arithmetic of a whole
encoder / decoder call is one C-ABI call (`tpspp_nrtr_encoder_fwd` / `tpspp_nrtr_decoder_fwd`) that
enqueues hand-written HIP kernels: there is no CPU or library-kernel path for inference, and the modules raise on
CPU tensors.  Under `.train()` (round 5) encoder and decoder run as PyTorch compositions of the same layers so that
autograd reaches their parameters (`_forward_graph`, `_forward_train_graph`).  The decoder is incremental (one position per step against cached
keys/values) where the reference re-runs the padded sequence every step; results agree to fp32
rounding (see tests/test_gpu_head.py).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .precision import default_compute_dtype
from . import ops
from .registry import (CONVERTORS, DECODERS, DETECTORS, ENCODERS, build_backbone, build_convertor,
                       build_decoder, build_encoder, build_preprocessor)


def _check_act(act_cfg):
    if act_cfg.get("type") not in ("mmcv.GELU", "GELU"):
        raise NotImplementedError(f"activation {act_cfg!r}: the HIP head implements the reference's default "
                                  "(mmcv.GELU, erf form) only")


class MultiHeadAttention(nn.Module):
    """Parameter holder: `linear_q/k/v` (dim_k -> dim_k), `fc` (dim_v -> d_model)."""

    def __init__(self, n_head=8, d_model=512, d_k=64, d_v=64, dropout=0.1, qkv_bias=False):
        super().__init__()
        if d_k != 64 or d_v != 64:
            raise NotImplementedError("the HIP head is built for d_k = d_v = 64 (every NRTR config of the reference)")
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        self.dim_k, self.dim_v = n_head * d_k, n_head * d_v
        self.linear_q = nn.Linear(self.dim_k, self.dim_k, bias=qkv_bias)
        self.linear_k = nn.Linear(self.dim_k, self.dim_k, bias=qkv_bias)
        self.linear_v = nn.Linear(self.dim_v, self.dim_v, bias=qkv_bias)
        self.fc = nn.Linear(self.dim_v, d_model, bias=qkv_bias)


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_in, d_hid, dropout=0.1, act_cfg=dict(type="Relu")):
        super().__init__()
        _check_act(act_cfg)
        self.w_1 = nn.Linear(d_in, d_hid)
        self.w_2 = nn.Linear(d_hid, d_in)


class PositionalEncoding(nn.Module):
    """Fixed sinusoid table, registered as the buffer `position_table` (1, n_position, d_hid)."""

    def __init__(self, d_hid=512, n_position=200, dropout=0):
        super().__init__()
        # float64 powers rounded to fp32, fp32 product with the position, sin / cos
        # (transformer_module.py:141-153)
        den = torch.Tensor([1.0 / np.power(10000, 2 * (j // 2) / d_hid) for j in range(d_hid)]).view(1, -1)
        tab = torch.arange(n_position).unsqueeze(-1).float() * den
        tab[:, 0::2] = torch.sin(tab[:, 0::2])
        tab[:, 1::2] = torch.cos(tab[:, 1::2])
        self.register_buffer("position_table", tab.unsqueeze(0))


_ENC_ORDER = ("norm", "self_attn", "norm", "ffn")
_DEC_ORDER = ("norm", "self_attn", "norm", "enc_dec_attn", "norm", "ffn")


class TFEncoderLayer(nn.Module):
    def __init__(self, d_model=512, d_inner=256, n_head=8, d_k=64, d_v=64, dropout=0.1, qkv_bias=False,
                 act_cfg=dict(type="mmcv.GELU"), operation_order=None):
        super().__init__()
        self.attn = MultiHeadAttention(n_head, d_model, d_k, d_v, qkv_bias=qkv_bias, dropout=dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.mlp = PositionwiseFeedForward(d_model, d_inner, dropout=dropout, act_cfg=act_cfg)
        self.norm2 = nn.LayerNorm(d_model)
        self.operation_order = operation_order or _ENC_ORDER
        if tuple(self.operation_order) != _ENC_ORDER:
            raise NotImplementedError("the HIP head implements the default pre-norm operation order only")


class TFDecoderLayer(nn.Module):
    def __init__(self, d_model=512, d_inner=256, n_head=8, d_k=64, d_v=64, dropout=0.1, qkv_bias=False,
                 act_cfg=dict(type="mmcv.GELU"), operation_order=None):
        super().__init__()
        if qkv_bias:
            raise NotImplementedError("decoder projections with bias are not supported by the HIP head")
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.self_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout, qkv_bias=qkv_bias)
        self.enc_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout, qkv_bias=qkv_bias)
        self.mlp = PositionwiseFeedForward(d_model, d_inner, dropout=dropout, act_cfg=act_cfg)
        self.operation_order = operation_order or _DEC_ORDER
        if tuple(self.operation_order) != _DEC_ORDER:
            raise NotImplementedError("the HIP head implements the default pre-norm operation order only")


def _state_key(module):
    return tuple((t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers()))


def _f32(t):
    return t.detach().float().contiguous()


def _valid_len(img_metas, n, t, device):
    """`_get_mask` of encoder and decoder: the first min(T, ceil(T * valid_ratio)) tokens are valid."""
    if img_metas is None:
        return None
    ratios = [m.get("valid_ratio", 1.0) for m in img_metas]
    if len(ratios) != n:
        raise ValueError("img_metas must hold one dict per image")
    vals = [min(t, math.ceil(t * r)) for r in ratios]
    if torch.device(device).type != "cuda":
        return torch.tensor(vals, dtype=torch.int32, device=device)
    # pinned staging + asynchronous copy: `torch.tensor(..., device=cuda)` copies from pageable memory, which blocks the HOST until
    # every kernel already queued on the stream has run (the whole backbone, 4 - 23 ms) -- the encoder's and the decoder's launches
    # then start behind an idle gap instead of being queued ahead (round 6; the caching host allocator keeps the staging buffer
    # alive until the copy has executed)
    return torch.tensor(vals, dtype=torch.int32).pin_memory().to(device, non_blocking=True)


def _arranged16(weight, x3=False):
    """nn.Linear weight (out, in) -> bf16, arranged as a 1x1 kernel for tpspp_conv2d_bf16_fwd (x3: hi + lo slabs)."""
    w = weight.detach().float()
    return ops.prep_conv_weight_bf16(w.view(w.shape[0], w.shape[1], 1, 1), x3=x3).arranged



# ---- training graph ("training widening") ---------------------------------------
# `.train()` turns every module of the recogniser into the plain PyTorch composition of its own layers, so that autograd
# reaches the parameters (the reference trains nrtr_tps++.py through mmocr/apis/train.py:56-70); the TPS++ transformation
# stage inside it still runs on the HIP kernels in both directions (tps_pp.TPS_PP._forward_autograd -> ops.warp_autograd).
# Dropout sits where the reference has it (transformer_module.py:30,93,120; nrtr_decoder.py:99) with the rate the module
# was built with; in eval mode these functions reproduce the HIP kernels' arithmetic up to summation order and are what
# the host-side tests compare against the oracle.
def _mha_graph(m, q_in, kv_in, mask, p_drop, training):
    """MultiHeadAttention.forward (transformer_module.py:71-96) on the parameters `m` holds.  mask: None, (N, Lk) or
    (N, Lq, Lk); 0 = masked."""
    import torch.nn.functional as Fn
    n, lq, _ = q_in.shape
    lk = kv_in.shape[1]
    q = Fn.linear(q_in, m.linear_q.weight, m.linear_q.bias).view(n, lq, m.n_head, m.d_k).transpose(1, 2)
    k = Fn.linear(kv_in, m.linear_k.weight, m.linear_k.bias).view(n, lk, m.n_head, m.d_k).transpose(1, 2)
    v = Fn.linear(kv_in, m.linear_v.weight, m.linear_v.bias).view(n, lk, m.n_head, m.d_v).transpose(1, 2)
    att = torch.matmul(q / (m.d_k ** 0.5), k.transpose(2, 3))
    if mask is not None:
        mk = mask.unsqueeze(1) if mask.dim() == 3 else mask.unsqueeze(1).unsqueeze(1)
        att = att.masked_fill(mk == 0, float("-inf"))
    att = Fn.dropout(Fn.softmax(att, dim=-1), p_drop, training)
    out = torch.matmul(att, v).transpose(1, 2).contiguous().view(n, lq, m.dim_v)
    return Fn.dropout(Fn.linear(out, m.fc.weight, m.fc.bias), p_drop, training)


def _ffn_graph(m, x, p_drop, training):
    """PositionwiseFeedForward.forward (transformer_module.py:122-128), GELU in its erf form."""
    import torch.nn.functional as Fn
    h = Fn.gelu(Fn.linear(x, m.w_1.weight, m.w_1.bias))
    return Fn.dropout(Fn.linear(h, m.w_2.weight, m.w_2.bias), p_drop, training)


def _ratio_mask(img_metas, n, t, device):
    """`_get_mask` of encoder / decoder as a (N, T) 0/1 tensor, None without metas."""
    vl = _valid_len(img_metas, n, t, device)
    if vl is None:
        return None
    return (torch.arange(t, device=device)[None, :] < vl[:, None]).to(torch.float32)


def _head_flags(compute_dtype):
    return ops.HEAD_BF16 if compute_dtype == torch.bfloat16 else ops.HEAD_BF16X3 if compute_dtype == "bf16x3" else 0


@ENCODERS.register_module()
class NRTREncoder(nn.Module):
    """Transformer encoder; `forward(feat (N, C, H, W), img_metas=None) -> (N, H*W, C)`."""

    def __init__(self, n_layers=6, n_head=8, d_k=64, d_v=64, d_model=512, d_inner=256, dropout=0.1,
                 init_cfg=None, **kwargs):
        super().__init__()
        self.init_cfg = init_cfg
        if d_model != n_head * d_k:
            raise ValueError("d_model must equal n_head * d_k (linear_q maps dim_k -> dim_k)")
        self.d_model, self.d_inner, self.n_head = d_model, d_inner, n_head
        self.dropout_p = float(dropout)    # (only the training graph applies it)
        self.compute_dtype = default_compute_dtype()   # torch.bfloat16: the wide projections on the bf16 matrix cores
        self.layer_stack = nn.ModuleList([
            TFEncoderLayer(d_model, d_inner, n_head, d_k, d_v, dropout=dropout, **kwargs) for _ in range(n_layers)])
        self.layer_norm = nn.LayerNorm(d_model)

    def init_weights(self):
        pass

    def _weights(self):
        x3 = self.compute_dtype == "bf16x3"
        b16 = self.compute_dtype == torch.bfloat16 or x3
        key = (_state_key(self), b16, x3)
        cache = getattr(self, "_w_cache", None)
        if cache is None or cache[0] != key:
            ts = []
            # fp32: k-major (in, out); bf16 / bf16x3 (TPSPP_HEAD_BF16[X3]): arranged for the bf16 1x1 convolution kernel
            km = (lambda w: _arranged16(w, x3)) if b16 else ops.kmajor
            for lyr in self.layer_stack:
                a = lyr.attn
                if b16:
                    wqkv = _arranged16(torch.cat([a.linear_q.weight, a.linear_k.weight, a.linear_v.weight], dim=0), x3)
                else:
                    wqkv = torch.cat([ops.kmajor(a.linear_q.weight), ops.kmajor(a.linear_k.weight),
                                      ops.kmajor(a.linear_v.weight)], dim=1).contiguous()
                bqkv = None if a.linear_q.bias is None else \
                    torch.cat([_f32(a.linear_q.bias), _f32(a.linear_k.bias), _f32(a.linear_v.bias)]).contiguous()
                ts += [_f32(lyr.norm1.weight), _f32(lyr.norm1.bias), wqkv, bqkv, km(a.fc.weight),
                       None if a.fc.bias is None else _f32(a.fc.bias), _f32(lyr.norm2.weight), _f32(lyr.norm2.bias),
                       km(lyr.mlp.w_1.weight), _f32(lyr.mlp.w_1.bias), km(lyr.mlp.w_2.weight),
                       _f32(lyr.mlp.w_2.bias)]
            cache = (key, ops.PtrTable(ts), _f32(self.layer_norm.weight), _f32(self.layer_norm.bias))
            self._w_cache = cache
        return cache[1:]

    def _forward_graph(self, feat, img_metas=None):
        """`NRTREncoder.forward` (nrtr_encoder.py:66-87) as a PyTorch composition of this module's layers: the TRAINING
        graph (autograd reaches the parameters; dropout active under .train()) and the host-side tests' reference."""
        import torch.nn.functional as Fn
        n, c, h, w = feat.shape
        x = feat.reshape(n, c, h * w).permute(0, 2, 1).contiguous()
        mask = _ratio_mask(img_metas, n, h * w, feat.device)
        p, tr = self.dropout_p, self.training
        for lyr in self.layer_stack:
            y = Fn.layer_norm(x, (c,), lyr.norm1.weight, lyr.norm1.bias, lyr.norm1.eps)
            x = x + _mha_graph(lyr.attn, y, y, mask, p, tr)
            x = x + _ffn_graph(lyr.mlp, Fn.layer_norm(x, (c,), lyr.norm2.weight, lyr.norm2.bias, lyr.norm2.eps), p, tr)
        return Fn.layer_norm(x, (c,), self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)

    def forward(self, feat, img_metas=None):
        if self.training or (torch.is_grad_enabled() and feat.requires_grad):
            ops.require_gpu(feat, "NRTREncoder")
            return self._forward_graph(feat.float(), img_metas)
        ops.require_gpu(feat, "NRTREncoder")
        ops.warn_detached_once(self, "NRTREncoder")
        table, g, b = self._weights()
        n, c, h, w = feat.shape
        if c != self.d_model:
            raise ValueError(f"NRTREncoder: feature width {c} != d_model {self.d_model}")
        vl = _valid_len(img_metas, n, h * w, feat.device)
        out, out_cm = ops.nrtr_encoder(feat.float(), table, len(self.layer_stack), self.d_inner, g, b, vl, holder=self,
                                       flags=_head_flags(self.compute_dtype))
        out._tpspp_cm = out_cm          # lets NRTRDecoder skip the re-layout of its input
        return out


@DECODERS.register_module()
class NRTRDecoder(nn.Module):
    """Transformer decoder; `forward(feat, out_enc, targets_dict=None, img_metas=None, train_mode=True)`.
    `train_mode=False`: greedy decoding -> per-step softmax scores (N, max_seq_len, num_classes - 1).
    `train_mode=True`: teacher-forced raw logits of `targets_dict['padded_targets']` (inference of the
    training graph only: no autograd)."""

    def __init__(self, n_layers=6, d_embedding=512, n_head=8, d_k=64, d_v=64, d_model=512, d_inner=256,
                 n_position=200, dropout=0.1, num_classes=93, max_seq_len=40, start_idx=1, padding_idx=92,
                 init_cfg=None, **kwargs):
        super().__init__()
        self.init_cfg = init_cfg
        if d_model != n_head * d_k or d_embedding != d_model:
            raise ValueError("d_model must equal n_head * d_k and d_embedding")
        self.padding_idx, self.start_idx, self.max_seq_len = padding_idx, start_idx, max_seq_len
        self.d_model, self.d_inner, self.n_head = d_model, d_inner, n_head
        self.dropout_p = float(dropout)    # (only the training graph applies it)
        self.compute_dtype = default_compute_dtype()   # torch.bfloat16: encoder K/V projected on the bf16 matrix cores, kept as bf16
        self.trg_word_emb = nn.Embedding(num_classes, d_embedding, padding_idx=padding_idx)
        self.position_enc = PositionalEncoding(d_embedding, n_position=n_position)
        self.layer_stack = nn.ModuleList([
            TFDecoderLayer(d_model, d_inner, n_head, d_k, d_v, dropout=dropout, **kwargs) for _ in range(n_layers)])
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.classifier = nn.Linear(d_model, num_classes - 1)      # <PAD> is never predicted

    def init_weights(self):
        pass

    def _weights(self):
        x3 = self.compute_dtype == "bf16x3"
        b16 = self.compute_dtype == torch.bfloat16
        key = (_state_key(self), b16, x3)
        cache = getattr(self, "_w_cache", None)
        if cache is None or cache[0] != key:
            ts = []
            # the token-major step pipeline (tpspp_head.hip) takes these shapes only; for every other one the arranged
            # entries stay NULL and the channel-major step kernels run (no arrangement constraint, no wasted copies)
            d_inner = self.layer_stack[0].mlp.w_1.weight.shape[0]
            fast = self.d_model in (256, 512) and d_inner in (256, 512) and self.classifier.weight.shape[0] % 4 == 0
            # the one-off key / value projections of the encoder output: bf16 both; bf16x3 the keys only
            kk = (lambda w: _arranged16(w, x3)) if (b16 or x3) else ops.kmajor
            kv = _arranged16 if b16 else ops.kmajor
            for lyr in self.layer_stack:
                sa, ea = lyr.self_attn, lyr.enc_attn
                wqkv = torch.cat([ops.kmajor(sa.linear_q.weight), ops.kmajor(sa.linear_k.weight),
                                  ops.kmajor(sa.linear_v.weight)], dim=1).contiguous()
                # each LayerNorm is folded into the projection that follows it (tpspp_linear_ln_fwd)
                qkv = ops.fold_layernorm(lyr.norm1.weight, lyr.norm1.bias, wqkv)
                q = ops.fold_layernorm(lyr.norm2.weight, lyr.norm2.bias, ops.kmajor(ea.linear_q.weight))
                w1 = ops.fold_layernorm(lyr.norm3.weight, lyr.norm3.bias, ops.kmajor(lyr.mlp.w_1.weight), lyr.mlp.w_1.bias)
                wfc, wfc2, w2 = ops.kmajor(sa.fc.weight), ops.kmajor(ea.fc.weight), ops.kmajor(lyr.mlp.w_2.weight)
                ts += [qkv[0], qkv[1], qkv[2], wfc, None,
                       q[0], q[1], q[2], kk(ea.linear_k.weight), None, kv(ea.linear_v.weight),
                       wfc2, None, w1[0], w1[1], w1[2], w2, _f32(lyr.mlp.w_2.bias)]
                # the six per-step projections arranged for the step GEMM: fp32 fragments, or split hi / lo bf16 for the
                # reduced-precision head
                arr = ops.arrange_x3 if (b16 or x3) else ops.arrange_f32
                ts += [arr(t) if fast else None for t in (qkv[0], wfc, q[0], wfc2, w1[0], w2)]
            cls = ops.fold_layernorm(self.layer_norm.weight, self.layer_norm.bias, ops.kmajor(self.classifier.weight),
                                     self.classifier.bias)
            # behind the layers: the classifier
            ts.append((ops.arrange_x3 if (b16 or x3) else ops.arrange_f32)(cls[0]) if fast else None)
            cache = (key, ops.PtrTable(ts), _f32(self.trg_word_emb.weight), _f32(self.position_enc.position_table[0]),
                     cls)
            self._w_cache = cache
        return cache[1:]

    def _run(self, out_enc, img_metas, forced):
        ops.require_gpu(out_enc, "NRTRDecoder", torch.is_grad_enabled() and out_enc.requires_grad)
        ops.warn_detached_once(self, "NRTRDecoder")
        n, t, c = out_enc.shape
        if c != self.d_model:
            raise ValueError(f"NRTRDecoder: encoder width {c} != d_model {self.d_model}")
        enc_cm = getattr(out_enc, "_tpspp_cm", None)
        if enc_cm is None or tuple(enc_cm.shape) != (c, n * t):
            enc_cm = ops.transpose2d(out_enc.float().reshape(n * t, c))
        table, emb, pos, cls = self._weights()
        vl = _valid_len(img_metas, n, t, out_enc.device)
        seq_len = self.max_seq_len if forced is None else forced.shape[1]
        out, tokens, status = ops.nrtr_decoder(enc_cm, n, t, table, len(self.layer_stack), self.d_inner, emb, pos, cls,
                                               seq_len, self.start_idx, self.padding_idx, vl, forced, holder=self,
                                               flags=_head_flags(self.compute_dtype))
        # The decode is asynchronous; its status word (0, or 1 after a barrier timeout of the persistent step kernel:
        # include/tpspp.h) travels with the scores: `AttnConvertor.tensor2idx` / `dist.recognize_sharded` fetch it in the copy
        # they make anyway and raise `TpsppError`; `check_status()` is the explicit (synchronising) form.
        self.last_tokens, self.last_status = tokens, status
        out._tpspp_status = status
        return out

    def check_status(self):
        """Raises `TpsppError` if the last decode reported a barrier timeout (synchronises with that decode)."""
        ops.check_decoder_status(getattr(self, "last_status", None))

    def _forward_train_graph(self, out_enc, targets, img_metas):
        """`NRTRDecoder.forward_train` (nrtr_decoder.py:95-151: embedding + position table, pad & causal self-attention
        mask, valid-ratio cross-attention mask, six pre-norm layers, LayerNorm(eps 1e-6), classifier) as a PyTorch
        composition of this module's layers: the TRAINING graph.  Returns raw logits (N, T, num_classes - 1)."""
        import torch.nn.functional as Fn
        n, t, c = out_enc.shape
        L = targets.shape[1]
        p, tr = self.dropout_p, self.training
        x = Fn.embedding(targets, self.trg_word_emb.weight, padding_idx=self.padding_idx)
        x = Fn.dropout(x + self.position_enc.position_table[:, :L], p, tr)
        causal = torch.tril(torch.ones((L, L), device=targets.device, dtype=torch.bool))[None]
        self_mask = ((targets != self.padding_idx)[:, None, :] & causal).to(torch.float32)       # (N, L, L)
        src_mask = _ratio_mask(img_metas, n, t, out_enc.device)
        for lyr in self.layer_stack:
            y = Fn.layer_norm(x, (c,), lyr.norm1.weight, lyr.norm1.bias, lyr.norm1.eps)
            x = x + _mha_graph(lyr.self_attn, y, y, self_mask, p, tr)
            y = Fn.layer_norm(x, (c,), lyr.norm2.weight, lyr.norm2.bias, lyr.norm2.eps)
            x = x + _mha_graph(lyr.enc_attn, y, out_enc, src_mask, p, tr)
            x = x + _ffn_graph(lyr.mlp, Fn.layer_norm(x, (c,), lyr.norm3.weight, lyr.norm3.bias, lyr.norm3.eps), p, tr)
        x = Fn.layer_norm(x, (c,), self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        return Fn.linear(x, self.classifier.weight, self.classifier.bias)

    def forward_train(self, feat, out_enc, targets_dict, img_metas):
        if self.training or (torch.is_grad_enabled() and out_enc.requires_grad):
            ops.require_gpu(out_enc, "NRTRDecoder")
            return self._forward_train_graph(out_enc.float(), targets_dict["padded_targets"].to(out_enc.device).long(),
                                             img_metas)
        targets = targets_dict["padded_targets"].to(out_enc.device).to(torch.int32).contiguous()
        return self._run(out_enc, img_metas, targets)

    def forward_test(self, feat, out_enc, img_metas):
        return self._run(out_enc, img_metas, None)

    def forward(self, feat, out_enc, targets_dict=None, img_metas=None, train_mode=True):
        self.train_mode = train_mode
        if train_mode:
            return self.forward_train(feat, out_enc, targets_dict, img_metas)
        return self.forward_test(feat, out_enc, img_metas)


@CONVERTORS.register_module()
class BaseConvertor:
    """Text <-> index conversion (`convertors/base.py`)."""
    start_idx = end_idx = padding_idx = 0
    unknown_idx = None
    lower = False
    DICT36 = tuple("0123456789abcdefghijklmnopqrstuvwxyz")
    DICT90 = tuple("0123456789abcdefghijklmnopqrstuvwxyz"
                   "ABCDEFGHIJKLMNOPQRSTUVWXYZ!\"#$%&'()"
                   "*+,-./:;<=>?@[\\]_`~")

    def __init__(self, dict_type="DICT90", dict_file=None, dict_list=None):
        assert dict_type in ("DICT36", "DICT90")
        assert dict_file is None or isinstance(dict_file, str)
        assert dict_list is None or isinstance(dict_list, list)
        if dict_file is not None:
            with open(dict_file, encoding="utf-8") as f:
                self.idx2char = [ln.strip() for ln in (x.rstrip("\n\r") for x in f) if ln.strip() != ""]
        elif dict_list is not None:
            self.idx2char = dict_list
        else:
            self.idx2char = list(self.DICT36 if dict_type == "DICT36" else self.DICT90)
        self.char2idx = {ch: i for i, ch in enumerate(self.idx2char)}

    def num_classes(self):
        return len(self.idx2char)

    def str2idx(self, strings):
        assert isinstance(strings, list)
        indexes = []
        for string in strings:
            if self.lower:
                string = string.lower()
            index = []
            for char in string:
                char_idx = self.char2idx.get(char, self.unknown_idx)
                if char_idx is None:
                    raise Exception(f"Chararcter: {char} not in dict, please check gt_label and use custom "
                                    'dict file, or set "with_unknown=True"')
                index.append(char_idx)
            indexes.append(index)
        return indexes

    def str2tensor(self, strings):
        raise NotImplementedError

    def idx2str(self, indexes):
        assert isinstance(indexes, list)
        return ["".join(self.idx2char[i] for i in index) for index in indexes]

    def tensor2idx(self, output):
        raise NotImplementedError


@CONVERTORS.register_module()
class AttnConvertor(BaseConvertor):
    """`convertors/attn.py`: <UKN>, <BOS/EOS>, <PAD> appended to the dictionary; `tensor2idx` takes the
    per-step arg-max, skips <PAD> and stops at <EOS>."""

    def __init__(self, dict_type="DICT90", dict_file=None, dict_list=None, with_unknown=True, max_seq_len=40,
                 lower=False, start_end_same=True, **kwargs):
        super().__init__(dict_type, dict_file, dict_list)
        assert isinstance(with_unknown, bool) and isinstance(max_seq_len, int) and isinstance(lower, bool)
        self.with_unknown, self.max_seq_len, self.lower = with_unknown, max_seq_len, lower
        self.start_end_same = start_end_same
        self.update_dict()

    def update_dict(self):
        self.unknown_idx = None
        if self.with_unknown:
            self.idx2char.append("<UKN>")
            self.unknown_idx = len(self.idx2char) - 1
        self.idx2char.append("<BOS/EOS>")
        self.start_idx = len(self.idx2char) - 1
        if not self.start_end_same:
            self.idx2char.append("<BOS/EOS>")
        self.end_idx = len(self.idx2char) - 1
        self.idx2char.append("<PAD>")
        self.padding_idx = len(self.idx2char) - 1
        self.char2idx = {ch: i for i, ch in enumerate(self.idx2char)}

    def str2tensor(self, strings):
        assert isinstance(strings, list) and all(isinstance(s, str) for s in strings)
        tensors, padded_targets = [], []
        for index in self.str2idx(strings):
            tensor = torch.LongTensor(index)
            tensors.append(tensor)
            src = torch.LongTensor(tensor.size(0) + 2).fill_(0)
            src[-1], src[0] = self.end_idx, self.start_idx
            src[1:-1] = tensor
            padded = (torch.ones(self.max_seq_len) * self.padding_idx).long()
            if src.size(0) > self.max_seq_len:
                padded = src[:self.max_seq_len]
            else:
                padded[:src.size(0)] = src
            padded_targets.append(padded)
        return {"targets": tensors, "padded_targets": torch.stack(padded_targets, 0).long()}

    def tensor2idx(self, outputs, img_metas=None):
        """`attn.py:112-143`.  A GPU tensor: arg-max, maximum and the scan (skip <PAD>, stop at the first <EOS>) run in one HIP
        kernel and ONE device->host copy brings (N, L) indices + scores (the reference: two copies per image) -- together
        with the decoder's status word when `outputs` comes from `NRTRDecoder` (a barrier timeout raises `TpsppError`
        instead of decoding NaN scores into strings).  A CPU tensor (host-side tests, outputs gathered on the host): the
        reference's own `torch.max` per batch."""
        idx, val, keep = self._scan(outputs)
        counts = keep.sum(1)
        flat_i, flat_v = idx[keep].tolist(), val[keep].astype(np.float64).tolist()
        indexes, scores, o = [], [], 0
        for c in counts.tolist():
            indexes.append(flat_i[o:o + c])
            scores.append(flat_v[o:o + c])
            o += c
        return indexes, scores

    def _scan(self, outputs):
        """-> (idx (N, L) int, val (N, L) float32, keep (N, L) bool) on the host: per-position arg-max / maximum and the
        positions the reference's scan keeps (attn.py:124-137)."""
        status = getattr(outputs, "_tpspp_status", None)            # (an attribute: read it before detach() makes a new tensor)
        outputs = outputs.detach()                                  # (the reference detaches as well: attn.py:129-130)
        if outputs.is_cuda:
            idx, val = ops.attn_tensor2idx(outputs.float(), self.end_idx, self.padding_idx, status=status)
            return idx, val, idx >= 0
        max_value, max_idx = torch.max(outputs, -1)
        idx, val = max_idx.numpy().astype(np.int64), max_value.numpy()
        n, L = idx.shape
        is_end = idx == self.end_idx
        first_end = np.where(is_end.any(1), is_end.argmax(1), L)
        return idx, val, (np.arange(L)[None, :] < first_end[:, None]) & (idx != self.padding_idx)

    def tensor2str(self, outputs, img_metas=None):
        """`idx2str(tensor2idx(outputs)[0])` and the scores in one pass -> (strings, scores): what `simple_test` needs
        (encode_decode_recognizer.py:219-232).  The strings are decoded from ONE code-point array for the whole batch instead
        of a Python join per character; rows holding a multi-character token (<UKN>, ...) take the per-character join."""
        idx, val, keep = self._scan(outputs)
        table = getattr(self, "_code_table", None)
        if table is None or table[1] is not self.idx2char or len(table[0]) != len(self.idx2char):
            codes = np.array([ord(ch) if len(ch) == 1 else 0 for ch in self.idx2char], dtype=np.uint32)
            table = self._code_table = (codes, self.idx2char)
        counts = keep.sum(1).tolist()
        flat_i = idx[keep]
        flat_c = table[0][flat_i]
        text = flat_c.astype("<u4").tobytes().decode("utf-32-le")
        flat_v = val[keep].astype(np.float64).tolist()
        slow = bool((flat_c == 0).any())
        strings, scores, o = [], [], 0
        for c in counts:
            strings.append(text[o:o + c])
            scores.append(flat_v[o:o + c])
            o += c
        if slow:                                                    # a multi-character dictionary entry somewhere in the batch
            o = 0
            for r, c in enumerate(counts):
                if c and (flat_c[o:o + c] == 0).any():
                    strings[r] = "".join(self.idx2char[i] for i in flat_i[o:o + c].tolist())
                o += c
        return strings, scores


@DETECTORS.register_module()
class EncodeDecodeRecognizer(nn.Module):
    """`recognizer/encode_decode_recognizer.py`: preprocessor -> backbone (with the TPS++ network called
    inside it) -> encoder -> decoder -> label convertor.  Inference (`simple_test`, `aug_test`,
    `forward(..., return_loss=False)`) runs on the HIP kernels; `forward_train` (round 5) builds the training graph:
    PyTorch compositions of every stage's layers around the HIP transformation stage (forward and backward kernels)."""

    def __init__(self, preprocessor=None, backbone=None, encoder=None, decoder=None, tpsnet=None, loss=None,
                 label_convertor=None, train_cfg=None, test_cfg=None, max_seq_len=40, pretrained=None,
                 kd_loss=False, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg
        assert label_convertor is not None
        label_convertor = dict(label_convertor, max_seq_len=max_seq_len)
        self.label_convertor = build_convertor(label_convertor)
        self.preprocessor = build_preprocessor(preprocessor) if preprocessor is not None else None
        assert backbone is not None
        self.backbone = build_backbone(backbone)
        self.tpsnet = build_backbone(tpsnet) if tpsnet is not None else None
        # configs/textrecog/nrtr/nrtr_tps++.py:34-38 builds `tpsnet=dict(type='TPS_PP')` next to backbone strides
        # [2,1,2,1,2]; the reference's hard-coded wiring cannot take that geometry.  A TPS_PP
        # whose wiring was not chosen explicitly follows the strides of the backbone that will call it.
        if self.tpsnet is not None and hasattr(self.tpsnet, "variant_for_strides") and not self.tpsnet.variant_explicit:
            want = self.tpsnet.variant_for_strides(getattr(self.backbone, "strides", None) or [])
            if want is not None:
                self.tpsnet.set_variant(want, explicit=False)
        self.kd_loss = kd_loss
        self.encoder = build_encoder(encoder) if encoder is not None else None
        if decoder is not None:
            decoder = dict(decoder, num_classes=self.label_convertor.num_classes(),
                           start_idx=self.label_convertor.start_idx, padding_idx=self.label_convertor.padding_idx,
                           max_seq_len=max_seq_len)
            self.decoder = build_decoder(decoder)
        else:
            self.decoder = None
        self.loss_cfg = None if loss is None else dict(loss, ignore_index=self.label_convertor.padding_idx)
        from .losses import build_loss
        self.loss = None if loss is None else build_loss(self.loss_cfg)
        self.train_cfg, self.test_cfg, self.max_seq_len = train_cfg, test_cfg, max_seq_len

    def set_compute_dtype(self, mode):
        """One switch for the precision of the wide matrix products of every stage (not in the reference: its modules
        are fp32).  None: the exact fp32 kernels; "bf16x3": fp32 tensors, three-term bf16 split (still within the
        1e-4 bar); torch.bfloat16: bf16 operands / feature maps / encoder keys and values (control points, TPS solve,
        grid, LayerNorms, softmaxes and the decoder's per-step projections stay fp32 in every mode)."""
        if mode not in (None, "bf16x3", torch.bfloat16):
            raise ValueError('set_compute_dtype: None, "bf16x3" or torch.bfloat16')
        self.backbone.compute_dtype = mode
        if self.tpsnet is not None:          # on bf16 activations TPS_PP follows its input; "bf16x3" must be told
            self.tpsnet.compute_dtype = mode if mode == "bf16x3" else None
        for m in (self.encoder, self.decoder):
            if m is not None and hasattr(m, "compute_dtype"):
                m.compute_dtype = mode
        if self.preprocessor is not None and hasattr(self.preprocessor, "LocalizationNetwork"):
            self.preprocessor.LocalizationNetwork.compute_dtype = mode
        return self

    def extract_feat(self, img, test=False, **kwargs):
        if self.preprocessor is not None:
            img = self.preprocessor(img)
        if self.tpsnet is not None:
            return self.backbone(img, self.tpsnet, test)
        return self.backbone(img)

    def forward_train(self, img, img_metas, **kwargs):
        """`EncodeDecodeRecognizer.forward_train` (encode_decode_recognizer.py:131-183): valid ratios, features, targets from
        `img_meta['text']`, encoder, teacher-forced decoder, loss dict.  In `.train()` mode every stage is the PyTorch
        composition of its layers (gradients reach all parameters) and the TPS++ transformation stage runs on the HIP
        kernels forward and backward; the loss is the config's (`TFLoss` / `CELoss`: tps_pp_amd/losses.py)."""
        for img_meta in img_metas:
            img_meta["valid_ratio"] = 1.0 * img_meta["resize_shape"][1] / img.size(-1)
        feat = self.extract_feat(img, False, **kwargs)
        if isinstance(feat, dict):
            feat = feat["output"]
        targets_dict = self.label_convertor.str2tensor([img_meta["text"] for img_meta in img_metas])
        out_enc = self.encoder(feat, img_metas) if self.encoder is not None else None
        if self.decoder is not None:
            out_dec = self.decoder(feat, out_enc, targets_dict, img_metas, train_mode=True)
        else:
            out_dec = out_enc
        if self.loss is None:
            raise ValueError("forward_train: the recogniser was built without a `loss` config")
        return self.loss(out_dec, targets_dict, img_metas)

    @torch.no_grad()          # the callers (mmdet's single/multi_gpu_test) run it under no_grad as well
    def simple_test(self, img, img_metas, **kwargs):
        for img_meta in img_metas:
            img_meta["valid_ratio"] = 1.0 * img_meta["resize_shape"][1] / img.size(-1)
        feat = self.extract_feat(img, test=True)
        if isinstance(feat, dict):
            feat = feat["output"]
        out_enc = self.encoder(feat, img_metas) if self.encoder is not None else None
        if self.decoder is not None:
            out_dec = self.decoder(feat, out_enc, None, img_metas, train_mode=False)
        else:
            out_dec = out_enc
        if hasattr(self.label_convertor, "tensor2str"):     # tensor2idx + idx2str in one pass (same strings, same scores)
            label_strings, label_scores = self.label_convertor.tensor2str(out_dec, img_metas)
        else:
            label_indexes, label_scores = self.label_convertor.tensor2idx(out_dec, img_metas)
            label_strings = self.label_convertor.idx2str(label_indexes)
        return [dict(text=s, score=sc) for s, sc in zip(label_strings, label_scores)]

    def merge_aug_results(self, aug_results):
        out_text, out_score = "", -1
        for result in aug_results:
            text = result[0]["text"]
            score = sum(result[0]["score"]) / max(1, len(text))
            if score > out_score:
                out_text, out_score = text, score
        return [dict(text=out_text, score=out_score)]

    def aug_test(self, imgs, img_metas, **kwargs):
        return self.merge_aug_results([self.simple_test(i, m, **kwargs) for i, m in zip(imgs, img_metas)])

    def forward_test(self, imgs, img_metas, **kwargs):
        """`BaseRecognizer.forward_test` (recognizer/base.py:49-72): a list = test-time augmentation."""
        if isinstance(imgs, list):
            assert len(imgs) > 0
            assert imgs[0].size(0) == 1, f"aug test does not support inference with batch size {imgs[0].size(0)}"
            assert len(imgs) == len(img_metas)
            return self.aug_test(imgs, img_metas, **kwargs)
        return self.simple_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        """`BaseRecognizer.forward` (recognizer/base.py:74-92)."""
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        if isinstance(img, list):
            for idx, each_img in enumerate(img):
                if each_img.dim() == 3:
                    img[idx] = each_img.unsqueeze(0)
        elif len(img_metas) == 1 and isinstance(img_metas[0], list):
            img_metas = img_metas[0]
        return self.forward_test(img, img_metas, **kwargs)


@DETECTORS.register_module()
class NRTR(EncodeDecodeRecognizer):
    """`recognizer/nrtr.py`."""
