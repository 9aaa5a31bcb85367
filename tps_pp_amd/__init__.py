"""tps_pp_amd: the TPS++ rectification hot path of simplify23/TPS_PP, MI355X-native.

Only what the path needs: `csrc/` (HIP kernels + the C ABI of include/tpspp.h), `ops` (tensor-level
binding), and the host-side mirrors of the reference's module API (`TPSPreprocessor`, `TPS_PP`,
registries).  Importing the package does not touch the GPU or load the native library; the first op
does, and raises if libtpspp_hip.so is missing (no fallback).
"""
from .registry import (BACKBONES, PREPROCESSOR, build_backbone, build_preprocessor,  # noqa: F401
                       register_into_mmocr)
from .tps_preprocessor import TPSPreprocessor, LocalizationNetwork, GridGenerator  # noqa: F401
from .tps_pp import TPS_PP, Attention_Enhanced_TPS  # noqa: F401
from .resnet_v2_large import ResNetABI_v2_large, BasicBlock  # noqa: F401
from .nrtr_modality_transformer import NRTRModalityTransform  # noqa: F401

__all__ = ["BACKBONES", "PREPROCESSOR", "build_backbone", "build_preprocessor",
           "register_into_mmocr", "TPSPreprocessor", "LocalizationNetwork", "GridGenerator",
           "TPS_PP", "Attention_Enhanced_TPS", "ResNetABI_v2_large", "BasicBlock",
           "NRTRModalityTransform"]
