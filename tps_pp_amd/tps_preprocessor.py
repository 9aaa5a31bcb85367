"""Classic RARE TPS-STN rectifier behind the reference's PREPROCESSOR API.

Mirror of `mmocr/models/textrecog/preprocessor/tps_preprocessor.py` (reference): same registry name,
constructor signature and assertion behaviour (`:37-49`), same sub-module / buffer names, hence the same
`state_dict` keys (`LocalizationNetwork.conv.{0,1,4,5,8,9,12,13}.*`, `LocalizationNetwork.
localization_fc1.0.*`, `LocalizationNetwork.localization_fc2.*`, `GridGenerator.inv_delta_C`,
`GridGenerator.P_hat`), same `forward(batch_img) -> Tensor`.

What differs is underneath: `GridGenerator.build_P_prime` + `F.grid_sample` (`:71-83`) run as ONE
hand-written HIP kernel (`tps_pp_amd/csrc/tpspp_warp.hip` through the C ABI of `include/tpspp.h`), and
the localisation network's convolutions (BatchNorm folded), pooling and FCs run on the hand-written
fp32 MFMA conv / pooling kernels when the module is in eval mode on a GPU.
There is no PyTorch / CPU fallback for the rectification path: CPU tensors raise.
"""
import numpy as np
import torch
import torch.nn as nn

from . import constants, ops
from .precision import default_compute_dtype
from .registry import PREPROCESSOR


class BasePreprocessor(nn.Module):
    """`preprocessor/base_preprocessor.py:8-13`: identity preprocessor with an `init_cfg` slot."""

    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass

    def forward(self, x, **kwargs):
        return x


class LocalizationNetwork(nn.Module):
    """Predicts the fiducial points C' (N, F, 2) from the image (`tps_preprocessor.py:88-156`)."""

    def __init__(self, num_fiducial, num_img_channel):
        super().__init__()
        self.num_fiducial = num_fiducial
        self.num_img_channel = num_img_channel
        self.conv = nn.Sequential(
            nn.Conv2d(num_img_channel, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64), nn.ReLU(True),
            nn.MaxPool2d(2, 2),
            nn.Conv2d(64, 128, 3, 1, 1, bias=False), nn.BatchNorm2d(128), nn.ReLU(True),
            nn.MaxPool2d(2, 2),
            nn.Conv2d(128, 256, 3, 1, 1, bias=False), nn.BatchNorm2d(256), nn.ReLU(True),
            nn.MaxPool2d(2, 2),
            nn.Conv2d(256, 512, 3, 1, 1, bias=False), nn.BatchNorm2d(512), nn.ReLU(True),
            nn.AdaptiveAvgPool2d(1))
        self.localization_fc1 = nn.Sequential(nn.Linear(512, 256), nn.ReLU(True))
        self.localization_fc2 = nn.Linear(256, num_fiducial * 2)
        # fc2 starts as "weight 0, bias = initial fiducials" (tps_preprocessor.py:130-140)
        self.localization_fc2.weight.data.fill_(0)
        self.localization_fc2.bias.data = torch.from_numpy(
            constants.classic_initial_ctrl(num_fiducial)).float().view(-1)

    def _hip_weights(self):
        """ConvWeights (eval-mode BatchNorm folded into the convolutions) per layer, rebuilt when a
        parameter or running statistic changes."""
        mods = [self.conv[i] for i in (0, 1, 4, 5, 8, 9, 12, 13)] + [self.localization_fc1[0], self.localization_fc2]
        key = tuple((t.data_ptr(), t._version) for m in mods for t in list(m.parameters()) + list(m.buffers()))
        cache = getattr(self, "_cw_cache", None)
        if cache is None or cache[0] != key:
            cw = []
            for ci, bi in ((0, 1), (4, 5), (8, 9), (12, 13)):
                bn = self.conv[bi]
                cw.append(ops.prep_conv_weight(self.conv[ci].weight, bn=(bn.weight, bn.bias, bn.running_mean,
                                                                       bn.running_var), eps=bn.eps))
            fc1, fc2 = self.localization_fc1[0], self.localization_fc2
            cw.append(ops.prep_conv_weight(fc1.weight.view(fc1.out_features, fc1.in_features, 1, 1), conv_bias=fc1.bias))
            cw.append(ops.prep_conv_weight(fc2.weight.view(fc2.out_features, fc2.in_features, 1, 1), conv_bias=fc2.bias))
            self._cw_cache = cache = (key, cw)
        return cache[1]

    def _hip_weights_bf16(self, x3=False):
        mods = [self.conv[i] for i in (0, 1, 4, 5, 8, 9, 12, 13)]
        key = (tuple((t.data_ptr(), t._version) for m in mods for t in list(m.parameters()) + list(m.buffers())), x3)
        cache = getattr(self, "_cw16_cache", None)
        if cache is None or cache[0] != key:
            cw = []
            for ci, bi in ((0, 1), (4, 5), (8, 9), (12, 13)):
                bn = self.conv[bi]
                cw.append(ops.prep_conv_weight_bf16(self.conv[ci].weight, eps=bn.eps, x3=x3,
                                                    bn=(bn.weight, bn.bias, bn.running_mean, bn.running_var)))
            self._cw16_cache = cache = (key, cw)
        return cache[1]

    def forward(self, batch_img):
        """fp32 MFMA convolutions with BatchNorm folded in, HIP pooling kernels, the two FCs as 1x1
        convolutions over the batch.  No CPU / library-kernel path."""
        ops.require_gpu(batch_img, "LocalizationNetwork", self.training)
        n = batch_img.size(0)
        cw = self._hip_weights()
        x = batch_img.float().contiguous()
        mode = getattr(self, "compute_dtype", None)
        if mode is None and not hasattr(self, "compute_dtype"):
            mode = default_compute_dtype()
        if mode == torch.bfloat16 or mode == "bf16x3":
            # bf16 configuration: the four convolutions on the bf16 matrix cores (fp32 maps in and out: operands are
            # rounded as they are staged, accumulation / bias / ReLU fp32); pooling and the two FCs stay fp32.
            # "bf16x3": the three-term split of the fp32 operands instead of a plain rounding (within the 1e-4 bar).
            c16 = self._hip_weights_bf16(mode == "bf16x3")
            for i in range(3):
                x = ops.maxpool2x2(ops.conv2d_bf16([x], c16[i], 1, True, out_dtype=torch.float32))
            x = ops.global_avgpool(ops.conv2d_bf16([x], c16[3], 1, True, out_dtype=torch.float32))
            x = ops.linear(x, cw[4], relu=True)
            return ops.linear(x, cw[5], relu=False).view(n, self.num_fiducial, 2)
        for i in range(3):
            x = ops.maxpool2x2(ops.conv2d([x], cw[i], 1, True))
        x = ops.global_avgpool(ops.conv2d([x], cw[3], 1, True))
        x = ops.linear(x, cw[4], relu=True)
        return ops.linear(x, cw[5], relu=False).view(n, self.num_fiducial, 2)

    def _forward_torch(self, batch_img):
        """TEST HOOK, never called by forward(): plain PyTorch composition of the same layers."""
        n = batch_img.size(0)
        feat = self.conv(batch_img).view(n, -1)
        return self.localization_fc2(self.localization_fc1(feat)).view(n, self.num_fiducial, 2)


class GridGenerator(nn.Module):
    """Holds inv_delta_C / P_hat and expands the sampling grid (`tps_preprocessor.py:159-282`)."""

    def __init__(self, num_fiducial, rectified_img_size):
        super().__init__()
        self.eps = constants.EPS
        self.rectified_img_height = rectified_img_size[0]
        self.rectified_img_width = rectified_img_size[1]
        self.num_fiducial = num_fiducial
        k = constants.classic(num_fiducial, rectified_img_size)
        self.C, self.P = k["C"], k["P"]
        self.register_buffer("inv_delta_C", torch.from_numpy(k["inv_delta_C"]))
        self.register_buffer("P_hat", torch.from_numpy(k["P_hat"]))
        self._prep = None        # (key, P_hat_t, table_flags): device-side view of P_hat for the kernels

    def prepared_table(self):
        """(P_hat_t, table_flags) for the current P_hat buffer: the transposed copy the coalesced
        kernels read (with the packed copy of the image-pair kernel behind it when the geometry has one:
        `ops.prepare_mirror_table`), and whether the table has the reference's mirror symmetry (checked
        bitwise, once per buffer version; a checkpoint that loads a different table simply gets 0)."""
        p = self.P_hat
        key = (p.data_ptr(), p._version, str(p.device))
        if self._prep is None or self._prep[0] != key:
            hw = (self.rectified_img_height, self.rectified_img_width)
            if ops.table_mirror_symmetry(p, hw, self.num_fiducial):
                P_hat_t, flags = ops.prepare_mirror_table(p, hw)
                flags |= ops.TABLE_MIRROR4
            else:
                P_hat_t, flags = ops.transpose_p_hat(p), 0
            self._prep = (key, P_hat_t, flags)
        return self._prep[1], self._prep[2]

    def build_P_prime(self, batch_C_prime, device="cuda"):
        """(N, F, 2) -> (N, n, 2): `bmm(P_hat, bmm(inv_delta_C, [C'; 0]))` as two HIP kernels with the
        reference's summation order (`tps_preprocessor.py:270-282`)."""
        T = ops.solve_T(self.inv_delta_C, batch_C_prime)
        return ops.build_grid(self.P_hat, T)


@PREPROCESSOR.register_module()
class TPSPreprocessor(BasePreprocessor):
    """Rectification network of RARE (TPS-based STN), `tps_preprocessor.py:24-85`.

    Args:
        num_fiducial (int): number of fiducial points.
        img_size (tuple(int, int)): (H, W) of the input image.
        rectified_img_size (tuple(int, int)): (H_r, W_r) of the rectified image.
        num_img_channel (int): input channels.
        init_cfg (dict or list[dict], optional): kept for config compatibility.
    """

    def __init__(self, num_fiducial=20, img_size=(32, 100), rectified_img_size=(32, 100),
                 num_img_channel=1, init_cfg=None):
        super().__init__(init_cfg=init_cfg)
        assert isinstance(num_fiducial, int)
        assert num_fiducial > 0
        assert isinstance(img_size, tuple)
        assert isinstance(rectified_img_size, tuple)
        assert isinstance(num_img_channel, int)
        self.num_fiducial = num_fiducial
        self.img_size = img_size
        self.rectified_img_size = rectified_img_size
        self.num_img_channel = num_img_channel
        self.LocalizationNetwork = LocalizationNetwork(num_fiducial, num_img_channel)
        self.GridGenerator = GridGenerator(num_fiducial, rectified_img_size)

    def rectify(self, batch_img, batch_C_prime, want_grid=False, want_idx=False):
        """The hot path alone: control points -> rectified image (fused HIP kernel)."""
        gg = self.GridGenerator
        P_hat_t, flags = gg.prepared_table()
        out, _, grid, idx = ops.warp(batch_img, batch_C_prime, gg.inv_delta_C, gg.P_hat,
                                     self.rectified_img_size, want_grid=want_grid,
                                     want_idx=want_idx, P_hat_t=P_hat_t, table_flags=flags)
        if want_grid or want_idx:
            return out, grid, idx
        return out

    def forward(self, batch_img):
        """(N, C, H, W) -> (N, C, H_r, W_r)."""
        if self.training or (torch.is_grad_enabled() and batch_img.requires_grad):
            # training graph, or an eval-mode module whose input carries gradients
            # (saliency / adversarial gradients w.r.t. the image: the reference is differentiable in eval mode too):
            # HIP warp forward + backward; the localisation network is the plain PyTorch composition (fp32) so that
            # autograd reaches its parameters.  Plain eval inference takes the HIP kernels (no autograd graph).
            ops.require_gpu(batch_img, "TPSPreprocessor")
            if not getattr(self, "_logged_autograd", False):
                import logging
                logging.getLogger("tps_pp_amd").warning(
                    "TPSPreprocessor.train(): localisation network as a PyTorch composition (library kernels, fp32); "
                    "warp forward / backward on the HIP kernels. Call .eval() for the all-HIP inference path.")
                self._logged_autograd = True
            gg = self.GridGenerator
            P_hat_t, flags = gg.prepared_table()
            ctrl = self.LocalizationNetwork._forward_torch(batch_img.float())
            return ops.warp_autograd(batch_img.float(), ctrl.float(), gg.inv_delta_C, gg.P_hat,
                                     self.rectified_img_size, P_hat_t=P_hat_t, table_flags=flags)
        batch_C_prime = self.LocalizationNetwork(batch_img)
        return self.rectify(batch_img.float(), batch_C_prime.float())
