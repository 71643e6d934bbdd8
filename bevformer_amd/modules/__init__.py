"""MI355X-native implementations of the reference's BEV-encoder plugin classes
(same names as projects/mmdet3d_plugin/bevformer/modules/__init__.py:3-5)."""
from .bricks import FFN
from .custom_base_transformer_layer import MyCustomBaseTransformerLayer
from .decoder import CustomMSDeformableAttention, DetectionTransformerDecoder
from .encoder import BEVFormerEncoder, BEVFormerLayer
from .spatial_cross_attention import MSDeformableAttention3D, SpatialCrossAttention
from .temporal_self_attention import TemporalSelfAttention
from .transformer import (PerceptionTransformer, PerceptionTransformerBEVEncoder, PerceptionTransformerV2,
                          ResNetFusion)

__all__ = ["BEVFormerEncoder", "BEVFormerLayer", "SpatialCrossAttention",
           "MSDeformableAttention3D", "TemporalSelfAttention", "MyCustomBaseTransformerLayer",
           "FFN", "PerceptionTransformer", "PerceptionTransformerBEVEncoder", "PerceptionTransformerV2", "ResNetFusion", "CustomMSDeformableAttention",
           "DetectionTransformerDecoder"]
