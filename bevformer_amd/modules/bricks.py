"""Feed-forward and normalisation bricks of the encoder layer.

With a real mmcv present the layer builds mmcv's own ``FFN`` through the
``FEEDFORWARD_NETWORK`` registry (as custom_base_transformer_layer.py:157-158
does); without it, the ``FFN`` below provides the same module with the same
parameter names (``layers.0.0.{weight,bias}``, ``layers.1.{weight,bias}``,
SURVEY.md §8a-K) so reference checkpoints load unchanged.
"""
import torch
import torch.nn as nn

from .. import ops
from ..registry import FEEDFORWARD_NETWORK, HAVE_MMCV, BaseModule, Sequential


class FFN(BaseModule):
    """``x + Drop(Linear(Drop(ReLU(Linear(x)))))`` — mmcv's FFN with
    ``add_identity=True``, called as ``ffn(query, identity=None)``
    (encoder.py:402-403)."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type="ReLU", inplace=True), ffn_drop=0.0, dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        if num_fcs < 2:
            raise ValueError(f"num_fcs should be no less than 2. got {num_fcs}.")
        if act_cfg.get("type", "ReLU") != "ReLU":
            raise NotImplementedError("only ReLU FFNs are used by the BEVFormer encoder")
        self.embed_dims = embed_dims
        self.feedforward_channels = feedforward_channels
        self.num_fcs = num_fcs
        layers, cin = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(Sequential(nn.Linear(cin, feedforward_channels), nn.ReLU(inplace=True),
                                     nn.Dropout(ffn_drop)))
            cin = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = Sequential(*layers)
        p = (dropout_layer or {}).get("drop_prob", 0.0) if dropout_layer else 0.0
        self.dropout_layer = nn.Dropout(p) if p else nn.Identity()
        self.add_identity = add_identity

    def _layers_inference(self, x, post_norm=None, identity=None):
        """Same math as ``self.layers`` with dropout inactive: the hidden
        Linear+ReLU pairs run with the bias+ReLU epilogue of the MFMA kernel
        (``ops.linear``; hipBLASLt's ``torch._addmm_activation`` in ``native`` GEMM mode)
        instead of a GEMM and a clamp launch."""
        lead = x.shape[:-1]
        h = x.reshape(-1, x.shape[-1])
        for blk in list(self.layers)[:-2]:
            fc = blk[0]
            y = ops.linear(h, fc.weight, fc.bias, relu=True, tag="ffn_fc1")
            h = y if y is not None else \
                torch._addmm_activation(fc.bias, h, fc.weight.t(), use_gelu=False)
        fc = self.layers[-2]
        if post_norm is not None:       # "+ identity" and the layer's norm in the epilogue of fc2
            y = ops.linear_layernorm(h, fc.weight, fc.bias, identity, post_norm, tag="ffn_fc2")
            if y is not None:
                return ops.Normed(y.view(*lead, fc.out_features))
        y = ops.linear(h, fc.weight, fc.bias, tag="ffn_fc2")
        if y is None:
            y = torch.addmm(fc.bias, h, fc.weight.t())
        return y.view(*lead, fc.out_features)

    def _dropout_inactive(self):
        return not self.training or all(m.p == 0 for m in self.layers.modules() if isinstance(m, nn.Dropout))

    def _layers_autograd(self, x):
        """``self.layers`` with dropout inactive, Linear layers through ``ops.linear_or_torch``
        (MFMA kernel inside an autograd Function; fc1's ReLU in its epilogue)."""
        h = x
        for blk in list(self.layers)[:-2]:
            h = ops.linear_or_torch(h, blk[0].weight, blk[0].bias, relu=True, tag="ffn_fc1")
        fc = self.layers[-2]
        return ops.linear_or_torch(h, fc.weight, fc.bias, tag="ffn_fc2")

    def forward(self, x, identity=None, defer_residual=False, post_norm=None):
        """``post_norm`` (inference): the LayerNorm that follows this step in the layer — the result is
        then ``ops.Normed(norm(identity + ffn(x)))`` when the fused kernel covers the shape."""
        if not self.training and not torch.is_grad_enabled():
            fuse = post_norm is not None and self.add_identity
            out = self._layers_inference(x, post_norm if fuse else None, x if identity is None else identity)
            if isinstance(out, ops.Normed):
                return out
        elif self._dropout_inactive() and x.is_cuda:
            out = self._layers_autograd(x)
        else:
            out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        if defer_residual:
            return self.dropout_layer(out), identity    # the layer fuses "+ identity" into its LayerNorm
        return identity + self.dropout_layer(out)


if not HAVE_MMCV:
    FEEDFORWARD_NETWORK.register_module(name="FFN", module=FFN)


def build_norm_layer(cfg, num_features):
    """Only ``dict(type='LN')`` occurs on this path (bevformer_base.py:78-105)."""
    if HAVE_MMCV:
        from mmcv.cnn import build_norm_layer as _b
        return _b(cfg, num_features)
    if cfg.get("type") != "LN":
        raise NotImplementedError(f"norm type {cfg.get('type')} is not used by the BEV encoder")
    return "ln", nn.LayerNorm(num_features, eps=cfg.get("eps", 1e-5))
