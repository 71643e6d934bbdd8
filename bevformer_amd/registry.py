"""Registry glue for the reference's plugin API.

The reference builds the encoder purely from config dicts through mmcv's
registries (``ATTENTION`` / ``TRANSFORMER_LAYER`` / ``TRANSFORMER_LAYER_SEQUENCE``
/ ``FEEDFORWARD_NETWORK``: spatial_cross_attention.py:14-17,31,178;
temporal_self_attention.py:25; encoder.py:24,242; transformer.py:53).  When a
real mmcv is importable the classes of this package register into *its*
registries (``force=True``: they deliberately take over the plugin's names so
that ``projects/mmdet3d_plugin`` configs resolve to the MI355X implementation);
otherwise a minimal registry with the same ``register_module()`` / ``build()``
surface is used, so configs written for the reference work unchanged here.
"""
import copy

import torch
import torch.nn as nn


class Registry:
    """Subset of ``mmcv.utils.Registry``: name -> class, ``register_module``
    decorator, ``build(cfg)``."""

    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def __contains__(self, key):
        return key in self._module_dict

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if not force and key in self._module_dict:
                raise KeyError(f"{key} is already registered in {self._name}")
            self._module_dict[key] = cls
            return cls
        return _register(module) if module is not None else _register

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or "type" not in cfg:
        raise TypeError(f"cfg must be a dict with a 'type' key, got {cfg!r}")
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop("type")
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError(f"{typ} is not in the {registry.name} registry")
    return cls(**args)


try:  # a real mmcv (not the test stub) is present: plug into it
    import mmcv as _mmcv
    if getattr(_mmcv, "__bevformer_amd_stub__", False):
        raise ImportError
    from mmcv.cnn.bricks.registry import (ATTENTION, FEEDFORWARD_NETWORK, TRANSFORMER_LAYER,
                                          TRANSFORMER_LAYER_SEQUENCE)
    from mmcv.runner import auto_fp16, force_fp32
    from mmcv.runner.base_module import BaseModule, ModuleList, Sequential
    HAVE_MMCV = True
except Exception:  # no mmcv in this environment (or only the oracle's stub)
    HAVE_MMCV = False
    ATTENTION = Registry("attention")
    FEEDFORWARD_NETWORK = Registry("feed-forward Network")
    TRANSFORMER_LAYER = Registry("transformerLayer")
    TRANSFORMER_LAYER_SEQUENCE = Registry("transformer-layers sequence")

    def _cast_floating(x, src, dst):
        """Tensors of dtype ``src`` inside ``x`` (nested lists / tuples / dicts) -> ``dst``; everything else as it is
        (integer level tensors, numpy camera matrices, strings)."""
        if isinstance(x, torch.Tensor):
            return x.to(dst) if x.dtype == src else x
        if isinstance(x, (str, bytes, nn.Module)):
            return x
        if isinstance(x, dict):
            return type(x)({k: _cast_floating(v, src, dst) for k, v in x.items()})
        if isinstance(x, (list, tuple)):
            return type(x)(_cast_floating(v, src, dst) for v in x)
        return x

    def _precision_wrapper(src, dst, autocast_on, out_attr):
        """``mmcv.runner.auto_fp16`` / ``force_fp32`` restated [third party: mmcv-full 1.4.0, docs/install.md:27; not on
        disk — mmcv/runner/fp16_utils.py as published]: a method decorator that is the identity while the module's
        ``fp16_enabled`` is False (the default; ``wrap_fp16_model`` sets it); otherwise the arguments NAMED in
        ``apply_to`` (default: the method's named positional parameters — keyword-only parameters are not in
        ``getfullargspec().args`` and stay as they are) are cast ``src`` -> ``dst`` and the method runs with CUDA
        autocast switched ``autocast_on`` (torch >= 1.6 branch of mmcv)."""
        import functools
        import inspect

        def factory(apply_to=None, **opts):
            out_cast = bool(opts.get(out_attr, False))

            def deco(fn):
                spec = inspect.getfullargspec(fn)

                @functools.wraps(fn)
                def wrapper(*args, **kwargs):
                    if not isinstance(args[0], nn.Module):
                        raise TypeError("@auto_fp16 / @force_fp32 can only decorate the methods of an nn.Module")
                    if not getattr(args[0], "fp16_enabled", False):
                        return fn(*args, **kwargs)
                    names = spec.args if apply_to is None else apply_to
                    arg_names = spec.args[:len(args)]
                    new_args = [_cast_floating(a, src, dst) if i < len(arg_names) and arg_names[i] in names else a
                                for i, a in enumerate(args)]
                    new_kwargs = {k: (_cast_floating(v, src, dst) if k in names else v) for k, v in kwargs.items()}
                    if torch.cuda.is_available():
                        with torch.autocast("cuda", enabled=autocast_on):
                            out = fn(*new_args, **new_kwargs)
                    else:
                        out = fn(*new_args, **new_kwargs)
                    return _cast_floating(out, dst, src) if out_cast else out
                return wrapper
            return deco
        return factory

    auto_fp16 = _precision_wrapper(torch.float, torch.half, True, "out_fp32")
    force_fp32 = _precision_wrapper(torch.half, torch.float, False, "out_fp16")

    class BaseModule(nn.Module):
        """``mmcv.runner.BaseModule`` surface used by this path."""

        def __init__(self, init_cfg=None):
            super().__init__()
            self._is_init = False
            self.init_cfg = copy.deepcopy(init_cfg)

        def init_weights(self):
            for m in self.children():
                if hasattr(m, "init_weights"):
                    m.init_weights()
            self._is_init = True

    class ModuleList(BaseModule, nn.ModuleList):
        def __init__(self, modules=None, init_cfg=None):
            BaseModule.__init__(self, init_cfg)
            nn.ModuleList.__init__(self, modules)

    class Sequential(BaseModule, nn.Sequential):
        def __init__(self, *args, init_cfg=None):
            BaseModule.__init__(self, init_cfg)
            nn.Sequential.__init__(self, *args)


try:  # mmdet's registry of whole transformers (modules/transformer.py:14,26)
    if not HAVE_MMCV:
        raise ImportError
    from mmdet.models.utils.builder import TRANSFORMER
except Exception:
    TRANSFORMER = Registry("Transformer")


def _build(cfg, registry, default_args=None):
    if HAVE_MMCV:
        from mmcv.utils import build_from_cfg as _bfc
        return _bfc(cfg, registry, default_args)
    return build_from_cfg(cfg, registry, default_args)


def build_attention(cfg, default_args=None):
    return _build(cfg, ATTENTION, default_args)


def build_feedforward_network(cfg, default_args=None):
    return _build(cfg, FEEDFORWARD_NETWORK, default_args)


def build_transformer_layer(cfg, default_args=None):
    return _build(cfg, TRANSFORMER_LAYER, default_args)


def build_transformer_layer_sequence(cfg, default_args=None):
    return _build(cfg, TRANSFORMER_LAYER_SEQUENCE, default_args)


def build_transformer(cfg, default_args=None):
    return _build(cfg, TRANSFORMER, default_args)


def wrap_fp16_model(model):
    """``mmcv.runner.wrap_fp16_model`` on torch >= 1.6 [third party, restated]: the parameters stay fp32, every
    submodule that has an ``fp16_enabled`` attribute gets it set — the ``@auto_fp16`` / ``@force_fp32`` methods then
    cast their inputs (reference call site: tools/fp16/train.py:224-226)."""
    for m in model.modules():
        if hasattr(m, "fp16_enabled"):
            m.fp16_enabled = True
    return model


def xavier_uniform_(module, bias=0.0):
    """mmcv ``xavier_init(m, distribution='uniform', bias=0.)``; a module
    without a weight (e.g. ``None``) is skipped like mmcv does."""
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.xavier_uniform_(module.weight)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def constant_(module, val, bias=0.0):
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)
