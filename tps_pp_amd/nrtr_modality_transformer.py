"""NRTR's "modality transform" conv stem behind the reference's BACKBONES API.

Mirror of `mmocr/models/textrecog/backbones/nrtr_modality_transformer.py:8-56`: conv3x3 s2 (3->32), ReLU,
BatchNorm, conv3x3 s2 (32->64), ReLU, BatchNorm, then a Linear(512, 512) over the (h, c) axis of every
image column.  Same `state_dict` keys (`conv_1`, `bn_1`, `conv_2`, `bn_2`, `linear`).  Both convolutions run on the fp32 MFMA kernel with ReLU and the *following* BatchNorm fused into
the epilogue (per-channel affine after the activation), and the Linear runs as a 1x1 convolution over
the (image, column) positions.
"""
import torch
import torch.nn as nn

from . import ops
from .registry import BACKBONES


@BACKBONES.register_module()
class NRTRModalityTransform(nn.Module):

    def __init__(self, input_channels=3, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg
        self.conv_1 = nn.Conv2d(input_channels, 32, kernel_size=3, stride=2, padding=1)
        self.relu_1 = nn.ReLU(True)
        self.bn_1 = nn.BatchNorm2d(32)
        self.conv_2 = nn.Conv2d(32, 64, kernel_size=3, stride=2, padding=1)
        self.relu_2 = nn.ReLU(True)
        self.bn_2 = nn.BatchNorm2d(64)
        self.linear = nn.Linear(512, 512)

    def init_weights(self):
        pass

    def _weights(self):
        mods = [self.conv_1, self.bn_1, self.conv_2, self.bn_2, self.linear]
        key = tuple((t.data_ptr(), t._version) for m in mods for t in list(m.parameters()) + list(m.buffers()))
        cache = getattr(self, "_cw_cache", None)
        if cache is None or cache[0] != key:
            def stage(conv, bn):
                return ops.prep_conv_weight(conv.weight, conv_bias=conv.bias, eps=bn.eps,
                                            post_bn=(bn.weight, bn.bias, bn.running_mean, bn.running_var))
            lin = ops.prep_conv_weight(self.linear.weight.view(512, 512, 1, 1), conv_bias=self.linear.bias)
            self._cw_cache = cache = (key, (stage(self.conv_1, self.bn_1), stage(self.conv_2, self.bn_2), lin))
        return cache[1]

    def forward(self, x):
        ops.require_gpu(x, "NRTRModalityTransform")
        if self.training or (torch.is_grad_enabled() and x.requires_grad):
            return self._forward_torch(x.float())           # training graph (round 5): autograd, BatchNorm batch statistics
        ops.warn_detached_once(self, "NRTRModalityTransform")
        c1, c2, lin = self._weights()
        x = ops.conv2d([x.float().contiguous()], c1, 2, True)
        x = ops.conv2d([x], c2, 2, True)
        n, c, h, w = x.size()
        # (n, c, h, w) -> rows (n, w) x features (h, c): layout plumbing for the Linear
        rows = x.permute(0, 3, 2, 1).contiguous().view(n * w, h * c)
        y = ops.linear(rows, lin)                               # (n*w, 512)
        return y.view(n, w, 512).permute(0, 2, 1).contiguous().view(n, -1, 1, w)

    def _forward_torch(self, x):
        """Plain PyTorch composition of the same layers: the TRAINING graph and the host-side tests' reference."""
        x = self.bn_1(self.relu_1(self.conv_1(x)))
        x = self.bn_2(self.relu_2(self.conv_2(x)))
        n, c, h, w = x.size()
        x = x.permute(0, 3, 2, 1).contiguous().view(n, w, h * c)
        x = self.linear(x)
        return x.permute(0, 2, 1).contiguous().view(n, -1, 1, w)
