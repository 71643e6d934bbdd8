"""Layer container: builds attentions / FFNs / norms from the config dict.

Same constructor contract as the reference's ``MyCustomBaseTransformerLayer``
(projects/mmdet3d_plugin/bevformer/modules/custom_base_transformer_layer.py:72-163):
``attn_cfgs`` list (``batch_first`` injected into each), deprecated
``feedforward_channels / ffn_dropout / ffn_num_fcs`` folded into ``ffn_cfgs``,
one LayerNorm per 'norm' in ``operation_order``; attributes ``attentions``,
``ffns``, ``norms``, ``pre_norm``, ``embed_dims``, ``num_attn`` are what
``BEVFormerLayer.forward`` and the callers read.
"""
import copy
import warnings

from ..registry import (TRANSFORMER_LAYER, BaseModule, ModuleList, build_attention,
                        build_feedforward_network)
from .bricks import build_norm_layer

_OPS = ("self_attn", "norm", "ffn", "cross_attn")
_DEPRECATED = dict(feedforward_channels="feedforward_channels", ffn_dropout="ffn_drop",
                   ffn_num_fcs="num_fcs")


@TRANSFORMER_LAYER.register_module(force=True)
class MyCustomBaseTransformerLayer(BaseModule):

    def __init__(self, attn_cfgs=None, ffn_cfgs=None, operation_order=None,
                 norm_cfg=dict(type="LN"), init_cfg=None, batch_first=True, **kwargs):
        if ffn_cfgs is None:
            ffn_cfgs = dict(type="FFN", embed_dims=256, feedforward_channels=1024, num_fcs=2,
                            ffn_drop=0.0, act_cfg=dict(type="ReLU", inplace=True))
        else:
            ffn_cfgs = copy.deepcopy(ffn_cfgs)
        for old, new in _DEPRECATED.items():
            if old in kwargs:
                warnings.warn(f"The arguments `{old}` in BaseTransformerLayer has been deprecated, "
                              f"now you should set `{new}` and other FFN related arguments to a "
                              "dict named `ffn_cfgs`. ")
                ffn_cfgs[new] = kwargs[old]
        super().__init__(init_cfg)
        self.batch_first = batch_first
        unknown = set(operation_order) - set(_OPS)
        assert not unknown, (f"The operation_order of {self.__class__.__name__} should contains "
                             f"all four operation type {list(_OPS)}")
        num_attn = operation_order.count("self_attn") + operation_order.count("cross_attn")
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        else:
            assert num_attn == len(attn_cfgs), (
                f"The length of attn_cfg {num_attn} is not consistent with the number of "
                f"attentionin operation_order {operation_order}.")
            attn_cfgs = [copy.deepcopy(c) for c in attn_cfgs]
        self.num_attn = num_attn
        self.operation_order = operation_order
        self.norm_cfg = norm_cfg
        self.pre_norm = operation_order[0] == "norm"

        self.attentions = ModuleList()
        index = 0
        for name in operation_order:
            if name in ("self_attn", "cross_attn"):
                if "batch_first" in attn_cfgs[index]:
                    assert self.batch_first == attn_cfgs[index]["batch_first"]
                else:
                    attn_cfgs[index]["batch_first"] = self.batch_first
                attention = build_attention(attn_cfgs[index])
                attention.operation_name = name
                self.attentions.append(attention)
                index += 1
        self.embed_dims = self.attentions[0].embed_dims

        self.ffns = ModuleList()
        num_ffns = operation_order.count("ffn")
        if isinstance(ffn_cfgs, dict):
            ffn_cfgs = [copy.deepcopy(ffn_cfgs) for _ in range(num_ffns)]
        assert len(ffn_cfgs) == num_ffns
        for cfg in ffn_cfgs:
            if "embed_dims" not in cfg:
                cfg["embed_dims"] = self.embed_dims
            else:
                assert cfg["embed_dims"] == self.embed_dims
            self.ffns.append(build_feedforward_network(cfg))

        self.norms = ModuleList()
        for _ in range(operation_order.count("norm")):
            self.norms.append(build_norm_layer(norm_cfg, self.embed_dims)[1])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        """Generic op-order interpreter (custom_base_transformer_layer.py:165-260)."""
        norm_i = attn_i = ffn_i = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None] * self.num_attn
        for op in self.operation_order:
            if op == "self_attn":
                query = self.attentions[attn_i](
                    query, query, query, identity if self.pre_norm else None,
                    query_pos=query_pos, key_pos=query_pos, attn_mask=attn_masks[attn_i],
                    key_padding_mask=query_key_padding_mask, **kwargs)
                attn_i += 1
                identity = query
            elif op == "norm":
                query = self.norms[norm_i](query)
                norm_i += 1
            elif op == "cross_attn":
                query = self.attentions[attn_i](
                    query, key, value, identity if self.pre_norm else None,
                    query_pos=query_pos, key_pos=key_pos, attn_mask=attn_masks[attn_i],
                    key_padding_mask=key_padding_mask, **kwargs)
                attn_i += 1
                identity = query
            elif op == "ffn":
                query = self.ffns[ffn_i](query, identity if self.pre_norm else None)
                ffn_i += 1
        return query
