"""Load the reference's hot-path files BY PATH under small stubs (build container only).

`/root/reference` does not import as a package (>=20 modules were never
released, and mmcv / mmdet / timm are not installed), so `make_golden.py` seeds `sys.modules`
with the few names the hot-path files import and then executes those files in place with
`importlib`.  Nothing from the reference is copied: the stubs restate the *public* mmcv 1.x
behaviour the files rely on:

* `mmcv.cnn.ConvModule(in, out, k, stride, padding)`  ==  `conv` = nn.Conv2d(bias=True)
  (`bias='auto'` => bias iff no norm layer) followed by `activate` = nn.ReLU(inplace=True)
  (default `act_cfg=dict(type='ReLU')`, order conv -> norm -> act).
* `mmcv.runner.BaseModule`  ==  nn.Module + `init_cfg`.
* `mmcv.cnn.resnet.BasicBlock` / `conv3x3`: conv3x3-bn-relu-conv3x3-bn (+downsample) -relu.
* `timm.models.layers.DropPath`: identity at drop_prob 0 (the only use, DGAB.py:67, never
  instantiates it).
* `mmocr.models.builder.{BACKBONES, PREPROCESSOR, ENCODERS, DECODERS, CONVERTORS}`: dict-backed
  `register_module()`; `build_activation_layer`: 'mmcv.GELU' -> nn.GELU(), 'Relu' -> nn.ReLU().
* `mmcv.runner.ModuleList` == nn.ModuleList.

This file is test tooling for fixture generation; it never travels to the GPU box in a form that
matters (nothing there reads /root/reference) and is not imported by the product.
"""
import importlib.util
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"


class _Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def build(self, cfg):
        cfg = dict(cfg)
        return self.module_dict[cfg.pop("type")](**cfg)


class _ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias="auto", conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"),
                 inplace=True, **kw):
        super().__init__()
        assert norm_cfg is None and conv_cfg is None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride,
                              padding=padding, dilation=dilation, groups=groups,
                              bias=True if bias == "auto" else bias)
        self.with_activation = act_cfg is not None
        if self.with_activation:
            assert act_cfg["type"] == "ReLU"
            self.activate = nn.ReLU(inplace=inplace)

    def forward(self, x):
        x = self.conv(x)
        if self.with_activation:
            x = self.activate(x)
        return x


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


def _conv3x3(in_planes, out_planes, stride=1, dilation=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation,
                     dilation=dilation, bias=False)


class _MMCVBasicBlock(nn.Module):
    """mmcv.cnn.resnet.BasicBlock (public 1.x definition)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style="pytorch",
                 with_cp=False):
        super().__init__()
        assert style in ["pytorch", "caffe"]
        self.conv1 = _conv3x3(inplanes, planes, stride, dilation)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation
        assert not with_cp

    def forward(self, x):
        residual = x
        out = self.conv1(x)
        out = self.bn1(out)
        out = self.relu(out)
        out = self.conv2(out)
        out = self.bn2(out)
        if self.downsample is not None:
            residual = self.downsample(x)
        out += residual
        out = self.relu(out)
        return out


def _build_activation_layer(cfg):
    """mmcv.cnn.build_activation_layer for the two types the transformer blocks ask for
    ('mmcv.GELU' -> nn.GELU, exact erf form; 'Relu'/'ReLU' -> nn.ReLU)."""
    t = cfg["type"]
    if t in ("mmcv.GELU", "GELU"):
        return nn.GELU()
    if t in ("Relu", "ReLU"):
        return nn.ReLU()
    raise KeyError(t)


class _DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        assert drop_prob == 0.0

    def forward(self, x):
        return x


def _pkg(name):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    return m


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, f"{REF}/{relpath}")
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


_loaded = {}


def load_reference():
    """Returns a dict of the reference's hot-path modules (executed from /root/reference)."""
    if _loaded:
        return _loaded
    # third-party stubs
    mmcv = _pkg("mmcv")
    cnn = _pkg("mmcv.cnn")
    cnn.ConvModule = _ConvModule
    resnet = _pkg("mmcv.cnn.resnet")
    resnet.BasicBlock = _MMCVBasicBlock
    resnet.conv3x3 = _conv3x3
    runner = _pkg("mmcv.runner")
    runner.BaseModule = _BaseModule
    runner.Sequential = nn.Sequential
    runner.ModuleList = nn.ModuleList
    mmcv.cnn, mmcv.runner = cnn, runner
    _pkg("timm")
    _pkg("timm.models")
    layers = _pkg("timm.models.layers")
    layers.DropPath = _DropPath
    # mmocr package skeleton + registries
    for p in ["mmocr", "mmocr.models", "mmocr.models.textrecog",
              "mmocr.models.textrecog.backbones", "mmocr.models.textrecog.backbones.tps_pp",
              "mmocr.models.textrecog.preprocessor"]:
        _pkg(p)
    builder = _pkg("mmocr.models.builder")
    builder.BACKBONES = _Registry("backbone")
    builder.PREPROCESSOR = _Registry("preprocessor")
    builder.ENCODERS = _Registry("encoder")
    builder.DECODERS = _Registry("decoder")
    builder.CONVERTORS = _Registry("convertor")
    builder.build_activation_layer = _build_activation_layer
    utils = _pkg("mmocr.utils")
    utils.is_type_list = lambda seq, t: isinstance(seq, list) and all(isinstance(x, t) for x in seq)
    utils.list_from_file = lambda fn, encoding="utf-8": [ln.rstrip("\n\r") for ln in open(fn, encoding=encoding)]
    sys.modules["mmocr"].utils = utils
    # names the backbone file imports but that were never released / are unrelated tooling
    tps = _pkg("mmocr.models.textrecog.backbones.tps")
    for n in ["U_TPSnet", "Deform_net", "DAttentionBaseline", "UDAT_Net", "TPSnet", "TPSnet_Warp",
              "TPSnetv2"]:
        setattr(tps, n, None)
    _pkg("tools")
    _pkg("tools.data")
    _pkg("tools.data.textrecog")
    vf = _pkg("tools.data.textrecog.visual_feat")
    vf.draw_feature_map = lambda *a, **k: None

    base = "mmocr/models/textrecog"
    _loaded["DGAB"] = _load("mmocr.models.textrecog.backbones.tps_pp.DGAB",
                            f"{base}/backbones/tps_pp/DGAB.py")
    _loaded["tps_pp"] = _load("mmocr.models.textrecog.backbones.tps_pp.tps_pp",
                              f"{base}/backbones/tps_pp/tps_pp.py")
    _loaded["base_preprocessor"] = _load("mmocr.models.textrecog.preprocessor.base_preprocessor",
                                         f"{base}/preprocessor/base_preprocessor.py")
    _loaded["tps_preprocessor"] = _load("mmocr.models.textrecog.preprocessor.tps_preprocessor",
                                        f"{base}/preprocessor/tps_preprocessor.py")
    conv_layer = _load("mmocr.models.textrecog.layers.conv_layer", f"{base}/layers/conv_layer.py")
    lay = _pkg("mmocr.models.textrecog.layers")
    lay.BasicBlock = conv_layer.BasicBlock
    _loaded["conv_layer"] = conv_layer
    _loaded["resnet_v2_large"] = _load("mmocr.models.textrecog.backbones.resnet_v2_large",
                                       f"{base}/backbones/resnet_v2_large.py")
    _loaded["nrtr_modality_transformer"] = _load(
        "mmocr.models.textrecog.backbones.nrtr_modality_transformer",
        f"{base}/backbones/nrtr_modality_transformer.py")
    # ---- recogniser head: transformer blocks, encoder, decoder, convertor
    for p in ["mmocr.models.common", "mmocr.models.common.modules", "mmocr.models.common.layers",
              "mmocr.models.textrecog.encoders", "mmocr.models.textrecog.decoders",
              "mmocr.models.textrecog.convertors"]:
        _pkg(p)
    tm = _load("mmocr.models.common.modules.transformer_module",
               "mmocr/models/common/modules/transformer_module.py")
    mods = sys.modules["mmocr.models.common.modules"]
    for n in ["ScaledDotProductAttention", "MultiHeadAttention", "PositionwiseFeedForward",
              "PositionalEncoding"]:
        setattr(mods, n, getattr(tm, n))
    tl = _load("mmocr.models.common.layers.transformer_layers",
               "mmocr/models/common/layers/transformer_layers.py")
    common = sys.modules["mmocr.models.common"]
    common.TFEncoderLayer, common.TFDecoderLayer = tl.TFEncoderLayer, tl.TFDecoderLayer
    common.PositionalEncoding = tm.PositionalEncoding
    _loaded["transformer_module"], _loaded["transformer_layers"] = tm, tl
    _load("mmocr.models.textrecog.encoders.base_encoder", f"{base}/encoders/base_encoder.py")
    _loaded["nrtr_encoder"] = _load("mmocr.models.textrecog.encoders.nrtr_encoder",
                                    f"{base}/encoders/nrtr_encoder.py")
    _load("mmocr.models.textrecog.decoders.base_decoder", f"{base}/decoders/base_decoder.py")
    _loaded["nrtr_decoder"] = _load("mmocr.models.textrecog.decoders.nrtr_decoder",
                                    f"{base}/decoders/nrtr_decoder.py")
    _load("mmocr.models.textrecog.convertors.base", f"{base}/convertors/base.py")
    _loaded["attn_convertor"] = _load("mmocr.models.textrecog.convertors.attn",
                                      f"{base}/convertors/attn.py")
    return _loaded
