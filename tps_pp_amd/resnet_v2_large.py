"""The recognizer backbone that calls TPS++: `ResNetABI_v2_large` behind the reference's API.

Mirror of `mmocr/models/textrecog/backbones/resnet_v2_large.py:25-196` + `layers/conv_layer.py:12-33`
(BasicBlock with `use_conv1x1=True`: conv1 = 1x1, conv2 = 3x3 with the block's stride, 1x1 + BN
downsample when the stride or the width changes): same constructor arguments, same `state_dict`
keys (`conv1.*`, `bn1.*`, `layer{1..5}.{i}.{conv1,bn1,conv2,bn2,downsample.0,downsample.1}.*`), same
call contract `forward(x, tpsnet=None, test=False, **kw) -> dict(output, img_ref)` with the TPS
network invoked before stage index 2 on `(x, outs)` (`:183-191`).

Every convolution runs on the hand-written fp32 MFMA kernel with its BatchNorm folded in and the
residual add + ReLU fused into the second convolution's epilogue (eval mode, GPU tensors; CPU tensors
raise: there is no CPU path).  Under `.train()` (round 5) the backbone is the PyTorch composition of its own
layers so that autograd and BatchNorm's batch statistics work; the TPS++ network it calls keeps its
transformation stage on the HIP kernels, forward and backward.
"""
import torch
import torch.nn as nn

from .precision import default_compute_dtype
from . import ops
from .registry import BACKBONES


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, use_conv1x1=False):
        super().__init__()
        if use_conv1x1:
            self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=1, bias=False)
            self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        else:   # mmcv.cnn.resnet.BasicBlock: 3x3 (stride) then 3x3
            self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
            self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride
        self.use_conv1x1 = use_conv1x1

    def forward(self, x):
        ops.require_gpu(x, "BasicBlock")
        if self.training or (torch.is_grad_enabled() and x.requires_grad):
            return self._forward_torch(x)                 # training graph (round 5): autograd, BatchNorm batch statistics
        ops.warn_detached_once(self, "BasicBlock")
        return self._forward_hip(x)

    def _forward_torch(self, x):
        """Plain PyTorch composition of the same layers: the TRAINING graph (`.train()`: BatchNorm uses and updates batch
        statistics, autograd reaches the parameters) and the host-side tests' reference; eval-mode forward() never takes it."""
        residual = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        return self.relu(out + residual)

    def _weights(self):
        mods = [self.conv1, self.bn1, self.conv2, self.bn2] + \
            ([self.downsample[0], self.downsample[1]] if self.downsample is not None else [])
        key = tuple((t.data_ptr(), t._version) for m in mods for t in list(m.parameters()) + list(m.buffers()))
        cache = getattr(self, "_cw_cache", None)
        if cache is None or cache[0] != key:
            def fold(conv, bn):
                return ops.prep_conv_weight(conv.weight, bn=(bn.weight, bn.bias, bn.running_mean, bn.running_var),
                                            eps=bn.eps)
            cw = [fold(self.conv1, self.bn1), fold(self.conv2, self.bn2),
                  fold(self.downsample[0], self.downsample[1]) if self.downsample is not None else None]
            self._cw_cache = cache = (key, cw)
        return cache[1]

    def _weights_bf16(self, x3=False):
        mods = [self.conv1, self.bn1, self.conv2, self.bn2] + \
            ([self.downsample[0], self.downsample[1]] if self.downsample is not None else [])
        key = tuple((t.data_ptr(), t._version) for m in mods for t in list(m.parameters()) + list(m.buffers()))
        name = "_cw16x3_cache" if x3 else "_cw16_cache"
        cache = getattr(self, name, None)
        if cache is None or cache[0] != key:
            def fold(conv, bn):
                return ops.prep_conv_weight_bf16(conv.weight, eps=bn.eps, x3=x3,
                                                 bn=(bn.weight, bn.bias, bn.running_mean, bn.running_var))
            cw = [fold(self.conv1, self.bn1), fold(self.conv2, self.bn2),
                  fold(self.downsample[0], self.downsample[1]) if self.downsample is not None else None]
            cache = (key, cw)
            setattr(self, name, cache)
        return cache[1]

    def _forward_hip_bf16(self, x, out_dtype=torch.bfloat16, x3=False, out_blocked=False):
        """The block on the bf16 matrix cores (BatchNorm folded in fp32, then rounded): activations bf16 in
        HBM, accumulation / bias / residual add / ReLU in fp32.  x3: fp32 activations, three-term bf16 split.
        The map between the block's two convolutions, and with `out_blocked` the block's result, are in the blocked layout
        (`ops.Blocked` / `ops.Blocked32`: 16-byte loads per patch position in the 3x3 convolution); `x` may be blocked."""
        c1, c2, cd = self._weights_bf16(x3)
        mid = torch.float32 if x3 else torch.bfloat16
        out = ops.conv2d_bf16([x], c1, self.conv1.stride, True, out_dtype=mid, out_blocked=True)
        if cd is None:
            residual = x
        else:
            residual = ops.conv2d_bf16([x], cd, self.downsample[0].stride, False, out_dtype=mid, out_blocked=True)
        if out_blocked:
            return ops.conv2d_bf16([out], c2, self.conv2.stride, True, residual=residual, res_mode=2, out_dtype=mid,
                                   out_blocked=True)
        return ops.conv2d_bf16([out], c2, self.conv2.stride, True, residual=residual, res_mode=2,
                               out_dtype=out_dtype)

    def _forward_hip(self, x):
        c1, c2, cd = self._weights()
        x = x.float().contiguous()
        s1 = self.conv1.stride
        out = ops.conv2d([x], c1, s1, True)
        residual = x if cd is None else ops.conv2d([x], cd, self.downsample[0].stride, False)
        # relu(bn2(conv2(out)) + residual): residual add and ReLU live in the conv epilogue
        return ops.conv2d([out], c2, self.conv2.stride, True, residual=residual, res_mode=2)


@BACKBONES.register_module()
class ResNetABI_v2_large(nn.Module):
    """`resnet_v2_large.py:25-196`."""

    def __init__(self, in_channels=3, stem_channels=32, base_channels=32, arch_settings=[3, 4, 6, 6, 3],
                 strides=[2, 1, 2, 1, 1], p_strides=[2, 1, 2, 1, 1], out_indices=None,
                 last_stage_pool=False, init_cfg=None):
        super().__init__()
        assert isinstance(in_channels, int)
        assert isinstance(stem_channels, int)
        assert isinstance(arch_settings, list) and all(isinstance(a, int) for a in arch_settings)
        assert isinstance(strides, list) and all(isinstance(a, int) for a in strides)
        assert len(arch_settings) == len(strides)
        assert out_indices is None or isinstance(out_indices, (list, tuple))
        assert isinstance(last_stage_pool, bool)
        self.init_cfg = init_cfg
        self.strides = list(strides)       # read by EncodeDecodeRecognizer to pick the TPS_PP wiring that fits
        # None: follow the input dtype; torch.bfloat16: bf16 convolutions; "bf16x3": fp32 tensors, three-term split
        self.compute_dtype = default_compute_dtype()
        self.out_indices = out_indices
        self.last_stage_pool = last_stage_pool
        self.block = BasicBlock
        self.inplanes = stem_channels
        self.conv1 = nn.Conv2d(in_channels, stem_channels, kernel_size=3, stride=1, padding=1)
        self.bn1 = nn.BatchNorm2d(stem_channels)
        self.relu1 = nn.ReLU()
        self.res_layers = []
        planes = base_channels
        for i, num_blocks in enumerate(arch_settings):
            layer = self._make_layer(self.inplanes, planes, num_blocks, strides[i])
            self.inplanes = planes * BasicBlock.expansion
            planes *= 2
            name = f"layer{i + 1}"
            self.add_module(name, layer)
            self.res_layers.append(name)

    def init_weights(self):
        pass

    @staticmethod
    def _make_layer(inplanes, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes))
        layers = [BasicBlock(inplanes, planes, stride=stride, downsample=downsample, use_conv1x1=True)]
        for _ in range(1, blocks):
            layers.append(BasicBlock(planes, planes, use_conv1x1=True))
        return nn.Sequential(*layers)

    def _stem(self, x):
        if True:
            mods = [self.conv1, self.bn1]
            key = tuple((t.data_ptr(), t._version) for m in mods for t in list(m.parameters()) + list(m.buffers()))
            cache = getattr(self, "_cw_cache", None)
            if cache is None or cache[0] != key:
                bn = self.bn1
                cw = ops.prep_conv_weight(self.conv1.weight, conv_bias=self.conv1.bias, eps=bn.eps,
                                          bn=(bn.weight, bn.bias, bn.running_mean, bn.running_var))
                self._cw_cache = cache = (key, cw)
            return ops.conv2d([x.float().contiguous()], cache[1], 1, True)

    def forward(self, x, tpsnet=None, test=False, **kwargs):
        """(N, 3, H, W) -> dict(output, img_ref); `tpsnet(x, outs, **kwargs)` runs before stage 2 and
        its 'output' replaces x (`resnet_v2_large.py:183-191`)."""
        ops.require_gpu(x, "ResNetABI_v2_large")
        if self.training or (torch.is_grad_enabled() and x.requires_grad):
            # training graph (round 5; the reference trains through mmocr/apis/train.py:56-70): the stem and the blocks as
            # PyTorch compositions of their own layers; `tpsnet` (in .train() mode as well) regresses with PyTorch layers and
            # runs the transformation stage on the HIP kernels in both directions (tps_pp.TPS_PP._forward_autograd)
            return self._forward_torch(x.float(), tpsnet, **kwargs)
        ops.warn_detached_once(self, "ResNetABI_v2_large")
        if x.dtype == torch.bfloat16 or self.compute_dtype == torch.bfloat16:
            # bf16 configuration (BASELINE.json configs[4]): every convolution on the bf16 matrix cores, bf16
            # activations in HBM (also through `tpsnet`, which follows its input dtype); the feature map handed
            # to the encoder leaves the last block in fp32
            last = getattr(self, self.res_layers[-1])[-1]
            # round 6: the stem's and the first stage's results are read by convolutions only when `tpsnet` is this package's
            # TPS_PP in the 'ResNet45' wiring (its three down convolutions) -- they then stay in the blocked layout as well
            # (the stem kernel writes it directly, the first stage's last block keeps it); same bits either way
            blocked_outs = tpsnet is not None and getattr(tpsnet, "accepts_blocked_outs", lambda: False)() and \
                len(self.res_layers) > 2
            stem = (lambda t: self._stem_bf16(t, out_blocked=True)) if blocked_outs else self._stem_bf16
            return self._run(x, tpsnet, stem, lambda blk, t, inner: blk._forward_hip_bf16(
                t, torch.float32 if blk is last else torch.bfloat16, out_blocked=inner), blocked_stage0=blocked_outs, **kwargs)
        if self.compute_dtype == "bf16x3":
            # fp32 tensors everywhere, every convolution product the three-term bf16 split (~5e-6 per layer); a
            # `tpsnet` that should do the same needs its own compute_dtype = "bf16x3"
            # (round 6: the stem -- 3 input channels, fp32 in and out -- on the exact-fp32 kernel: the three-term split filled
            # 3 of its 16-channel chunk, 276 us per 512 images against 149)
            return self._run(x.float(), tpsnet, self._stem,
                             lambda blk, t, inner: blk._forward_hip_bf16(t, torch.float32, True, out_blocked=inner), **kwargs)
        return self._run(x, tpsnet, self._stem, lambda blk, t, inner: blk(t), **kwargs)

    def _stem_bf16(self, x, x3=False, out_blocked=False):
        mods = [self.conv1, self.bn1]
        key = tuple((t.data_ptr(), t._version) for m in mods for t in list(m.parameters()) + list(m.buffers()))
        name = "_cw16x3_cache" if x3 else "_cw16_cache"
        cache = getattr(self, name, None)
        if cache is None or cache[0] != key:
            bn = self.bn1
            cw = ops.prep_conv_weight_bf16(self.conv1.weight, conv_bias=self.conv1.bias, eps=bn.eps, x3=x3,
                                           bn=(bn.weight, bn.bias, bn.running_mean, bn.running_var))
            cache = (key, cw)
            setattr(self, name, cache)
        if out_blocked and not x3:
            return ops.conv2d_bf16([x], cache[1], 1, True, out_blocked=True)
        return ops.conv2d_bf16([x], cache[1], 1, True, out_dtype=torch.float32 if x3 else torch.bfloat16)

    def _run(self, x, tpsnet, stem, apply_block, blocked_stage0=False, **kwargs):
        x = stem(x)
        outs = []
        outputs = None
        for i, name in enumerate(self.res_layers):
            if i == 2 and tpsnet is not None:
                outputs = tpsnet(x, outs, **kwargs)
                if outputs.get("output", None) is not None:
                    x = outputs["output"]
            outs.append(x)
            blocks = list(getattr(self, name))
            # the results of stages 0 / 1 are handed to `tpsnet` (with `outs`) and the last stage's to the caller: those stay
            # NCHW; the others only feed the next stage's first block, whose convolutions take the blocked layout as well
            # (its two 1x1 layers then run on tpspp_conv1x1_blk.hip)
            # (blocked_stage0, round 6: TPS++ takes blocked maps -- stage 0's result stays blocked; stage 1's last 3x3 runs on the
            # persistent blocked kernel as well and its result is brought to NCHW planes for the sampler by one copy kernel,
            # remembering its blocked twin for TPS++'s down2)
            hidden = 2 <= i < len(self.res_layers) - 1 or (i <= 1 and blocked_stage0)
            for j, blk in enumerate(blocks):
                # `inner`: nobody but this backbone's next block reads the result
                x = apply_block(blk, x, j + 1 < len(blocks) or hidden)
            if i == 1 and blocked_stage0 and isinstance(x, ops.Blocked):
                x = x.nchw_hip()
        return {"output": x, "img_ref": outputs.get("output", None) if outputs is not None else None}

    def _forward_torch(self, x, tpsnet=None, **kwargs):
        """Plain PyTorch composition of the same layers: the TRAINING graph and the host-side tests' reference."""
        return self._run(x, tpsnet, lambda t: self.relu1(self.bn1(self.conv1(t))),
                         lambda blk, t, inner: blk._forward_torch(t), **kwargs)
