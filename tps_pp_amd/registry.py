"""Registry boundary: the names the reference registers, buildable from the same config dicts.

Reference: `mmocr/models/builder.py:14,18,50-52,70-72` (`PREPROCESSOR`, `BACKBONES`,
`build_preprocessor`, `build_backbone`), used by `EncodeDecodeRecognizer.__init__`
(`recognizer/encode_decode_recognizer.py:43-44,50-51`) with `dict(type='TPSPreprocessor', ...)` /
`dict(type='TPS_PP')` (`configs/textrecog/nrtr/nrtr_tps++.py:38`).

When a real MMOCR is importable the modules are ALSO registered into its registries (so its own
`build_backbone` finds them and the config runs unchanged); otherwise the small mmcv-compatible
`Registry` below is all there is.  mmcv/mmdet are not required.
"""
import inspect


class Registry:
    """Minimal mmcv-style registry: `register_module()` decorator + `build(cfg)`."""

    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def __contains__(self, key):
        return key in self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def _register(self, cls, name=None, force=False):
        if not inspect.isclass(cls):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        names = [name or cls.__name__] if not isinstance(name, (list, tuple)) else list(name)
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f"{n} is already registered in {self._name}")
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict):
            raise TypeError(f"cfg must be a dict, but got {type(cfg)}")
        if "type" not in cfg:
            raise KeyError('`cfg` must contain the key "type"')
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        t = args.pop("type")
        if isinstance(t, str):
            cls = self.get(t)
            if cls is None:
                raise KeyError(f"{t} is not in the {self._name} registry")
        elif inspect.isclass(t):
            cls = t
        else:
            raise TypeError(f"type must be a str or a class, but got {type(t)}")
        return cls(**args)


BACKBONES = Registry("backbone")
PREPROCESSOR = Registry("preprocessor")
ENCODERS = Registry("encoder")
DECODERS = Registry("decoder")
CONVERTORS = Registry("convertor")
DETECTORS = BACKBONES          # builder.py:18-20: recognisers share the 'models' registry with the backbones


def build_backbone(cfg):
    """`mmocr.models.builder.build_backbone` (builder.py:70-72)."""
    return BACKBONES.build(cfg)


def build_preprocessor(cfg):
    """`mmocr.models.builder.build_preprocessor` (builder.py:50-52)."""
    return PREPROCESSOR.build(cfg)


def build_encoder(cfg):
    """`mmocr.models.builder.build_encoder` (builder.py:40-42)."""
    return ENCODERS.build(cfg)


def build_decoder(cfg):
    """`mmocr.models.builder.build_decoder` (builder.py:45-47)."""
    return DECODERS.build(cfg)


def build_convertor(cfg):
    """`mmocr.models.builder.build_convertor` (builder.py:35-37)."""
    return CONVERTORS.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """`mmocr.models.builder.build_detector` / `build_recognizer`: the recogniser from the `model` dict
    of a config such as configs/textrecog/nrtr/nrtr_tps++.py:26-42."""
    return DETECTORS.build(cfg, dict(train_cfg=train_cfg, test_cfg=test_cfg))


def register_into_mmocr(force=True):
    """If MMOCR is installed, make its own registries build our modules.  Returns True on success."""
    try:
        from mmocr.models import builder as _b  # noqa: WPS433 (optional dependency)
    except Exception:
        return False
    for reg_name, ours in (("BACKBONES", BACKBONES), ("PREPROCESSOR", PREPROCESSOR), ("ENCODERS", ENCODERS),
                           ("DECODERS", DECODERS), ("CONVERTORS", CONVERTORS)):
        theirs = getattr(_b, reg_name, None)
        if theirs is None:
            continue
        for name, cls in ours.module_dict.items():
            theirs.register_module(name=name, force=force, module=cls)
    return True
