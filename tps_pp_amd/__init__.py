"""tps_pp_amd: the TPS++ rectification hot path of simplify23/TPS_PP, MI355X-native.

Only what the path needs: `csrc/` (HIP kernels + the C ABI of include/tpspp.h), `ops` (tensor-level
binding), and the host-side mirrors of the reference's module API (`TPSPreprocessor`, `TPS_PP`,
registries).  Importing the package does not touch the GPU or load the native library; the first op
does, and raises if libtpspp_hip.so is missing (no fallback).
"""
from .registry import (BACKBONES, PREPROCESSOR, ENCODERS, DECODERS, CONVERTORS, DETECTORS,  # noqa: F401
                       build_backbone, build_preprocessor, build_encoder, build_decoder, build_convertor,
                       build_detector, register_into_mmocr)
from .tps_preprocessor import TPSPreprocessor, LocalizationNetwork, GridGenerator  # noqa: F401
from .tps_pp import TPS_PP, Attention_Enhanced_TPS  # noqa: F401
from .resnet_v2_large import ResNetABI_v2_large, BasicBlock  # noqa: F401
from .nrtr_modality_transformer import NRTRModalityTransform  # noqa: F401
from .nrtr_head import (NRTREncoder, NRTRDecoder, AttnConvertor, BaseConvertor,  # noqa: F401
                        EncodeDecodeRecognizer, NRTR, TFEncoderLayer, TFDecoderLayer, MultiHeadAttention,
                        PositionwiseFeedForward, PositionalEncoding)

from .ocr_transforms import PIPELINES, ResizeOCR, NormalizeOCR, OCRBatchPreprocessor  # noqa: F401

__all__ = ["PIPELINES", "ResizeOCR", "NormalizeOCR", "OCRBatchPreprocessor",
           "BACKBONES", "PREPROCESSOR", "build_backbone", "build_preprocessor",
           "register_into_mmocr", "TPSPreprocessor", "LocalizationNetwork", "GridGenerator",
           "TPS_PP", "Attention_Enhanced_TPS", "ResNetABI_v2_large", "BasicBlock",
           "NRTRModalityTransform", "ENCODERS", "DECODERS", "CONVERTORS", "DETECTORS", "build_encoder",
           "build_decoder", "build_convertor", "build_detector", "NRTREncoder", "NRTRDecoder", "AttnConvertor",
           "BaseConvertor", "EncodeDecodeRecognizer", "NRTR"]
