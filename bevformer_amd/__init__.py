"""bevformer_amd — MI355X-native BEV-encoder hot path of BEVFormer.

Importing the package registers ``BEVFormerEncoder``, ``BEVFormerLayer``,
``TemporalSelfAttention``, ``SpatialCrossAttention`` and
``MSDeformableAttention3D`` under the reference's registry names; the sampling
operator lives in ``lib/libbevmsda.so`` (HIP, gfx950) behind the C ABI of
``include/bevmsda.h``.
"""
from . import registry
from .registry import build_transformer, build_transformer_layer_sequence
from . import modules  # noqa: F401  (registers the classes)
from .functions import (MultiScaleDeformableAttnFunction_fp16,
                        MultiScaleDeformableAttnFunction_fp32)

__all__ = ["registry", "modules", "build_transformer_layer_sequence", "build_transformer",
           "MultiScaleDeformableAttnFunction_fp32", "MultiScaleDeformableAttnFunction_fp16"]
__version__ = "0.1.0"
