"""The recognition losses the NRTR / TPS++ configs name (`loss=dict(type='TFLoss')`, configs/textrecog/nrtr/nrtr_tps++.py:40),
buildable from the same config dicts (reference: `mmocr/models/textrecog/losses/ce_loss.py`, built by
`encode_decode_recognizer.py:68-70` with `ignore_index = label_convertor.padding_idx`).

Only `EncodeDecodeRecognizer.forward_train` -- the training graph of round 5 -- needs them; the cross-entropy itself is
`torch.nn.functional.cross_entropy`.  One function does the work; the two registered classes only fix how outputs and
targets are aligned.  Everything else under the reference's `losses/` stays out of scope.
"""
import torch.nn as nn
import torch.nn.functional as Fn

from .registry import Registry

LOSSES = Registry("loss")
_REDUCTIONS = ("none", "mean", "sum")


def sequence_cross_entropy(logits, targets, ignore_index, reduction, shift, flatten):
    """Cross-entropy of (N, T, C) logits against (N, T) class indices.
    shift:   position t is scored against target t + 1 (the targets start with <SOS>: the last output and the first
             target are dropped);
    flatten: score (N T', C) rows against N T' targets, else (N, C, T') against (N, T') -- this only changes the shape of
             an unreduced loss."""
    targets = targets.to(logits.device)
    if shift:
        logits, targets = logits[:, :-1, :], targets[:, 1:]
    if flatten:
        return Fn.cross_entropy(logits.reshape(-1, logits.size(-1)), targets.reshape(-1), ignore_index=ignore_index,
                                reduction=reduction)
    return Fn.cross_entropy(logits.permute(0, 2, 1).contiguous(), targets.contiguous(), ignore_index=ignore_index,
                            reduction=reduction)


class _SequenceLoss(nn.Module):
    shift_default = False

    def __init__(self, ignore_index, reduction, shift, flatten):
        super().__init__()
        if not isinstance(ignore_index, int) or reduction not in _REDUCTIONS:
            raise AssertionError("ignore_index must be an int, reduction one of " + ", ".join(_REDUCTIONS))
        self.ignore_index, self.reduction, self.shift, self.flatten = ignore_index, reduction, bool(shift), bool(flatten)

    def forward(self, outputs, targets_dict, img_metas=None):
        """-> {'loss_ce': tensor}; `targets_dict['padded_targets']` (N, T) as `AttnConvertor.str2tensor` builds it."""
        return {"loss_ce": sequence_cross_entropy(outputs, targets_dict["padded_targets"], self.ignore_index,
                                                  self.reduction, self.shift, self.flatten)}


@LOSSES.register_module()
class CELoss(_SequenceLoss):
    """`ignore_first_char=True` aligns output t with target t + 1."""

    def __init__(self, ignore_index=-1, reduction="none", ignore_first_char=False):
        assert isinstance(ignore_first_char, bool)
        super().__init__(ignore_index, reduction, shift=ignore_first_char, flatten=False)


@LOSSES.register_module()
class TFLoss(_SequenceLoss):
    """The transformer recognisers' loss: always shifted; `flatten` (default) scores all positions as one batch."""

    def __init__(self, ignore_index=-1, reduction="none", flatten=True, **kwargs):
        assert isinstance(flatten, bool)
        super().__init__(ignore_index, reduction, shift=True, flatten=flatten)


def build_loss(cfg):
    return LOSSES.build(cfg)
