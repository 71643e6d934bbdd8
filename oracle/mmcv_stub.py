"""TEST INFRASTRUCTURE — not part of the product path.

A minimal stand-in for the parts of ``mmcv-full==1.4.0`` (reference pin:
docs/install.md:27) that the reference's five BEV-encoder files import, plus a
loader that executes those files *unmodified* from ``/root/reference`` on CPU
(recipe: SURVEY.md §8c).  mmcv itself is not installed here and its source is
not on disk, so the three third-party pieces the reference leans on are
restated from their published behaviour:

* ``multi_scale_deformable_attn_pytorch`` -> ``oracle.bevformer_cpu.msda_gridsample``
* ``FFN``                                -> Linear-ReLU-Drop-Linear-Drop + identity
* ``TransformerLayerSequence``           -> deep-copies the layer cfg ``num_layers`` times

This module is only usable where ``/root/reference`` exists (the build
container).  It is used by ``oracle/make_golden.py`` to produce the fixtures in
``tests/golden/`` and by the CPU tests that cross-check the restatement in
``oracle/bevformer_cpu.py`` against the reference's own code.  Nothing on the
GPU box imports it.
"""
import copy
import importlib
import os
import sys
import types
import warnings

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"
_MODULES_DIR = os.path.join(REFERENCE_ROOT, "projects/mmdet3d_plugin/bevformer/modules")


def reference_available():
    return os.path.isdir(_MODULES_DIR)


class _Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, **default_args):
        return _build_from_cfg(cfg, self, default_args or None)


class _ConfigDict(dict):
    # missing keys must raise AttributeError or copy.deepcopy probes break
    # (custom_base_transformer_layer.py:147-149 deep-copies a ConfigDict)
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop("type")
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError(f"{typ} is not in the {registry.name} registry")
    return cls(**args)


def _passthrough_decorator_factory(*a, **k):
    def deco(fn):
        return fn
    return deco


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = copy.deepcopy(init_cfg)

    def init_weights(self):
        for m in self.children():
            if hasattr(m, "init_weights"):
                m.init_weights()


class _ModuleList(_BaseModule, nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        _BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


class _Sequential(_BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        _BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


def _xavier_init(module, gain=1, bias=0, distribution="normal"):
    if hasattr(module, "weight") and module.weight is not None:
        if distribution == "uniform":
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def _constant_init(module, val, bias=0):
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def _digit_version(v):
    out = []
    for p in str(v).split("+")[0].split(".")[:3]:
        num = "".join(ch for ch in p if ch.isdigit())
        out.append(int(num) if num else 0)
    return tuple(out)


def install(ext_module=None):
    """Put the stub ``mmcv`` (and empty ``projects.*`` packages whose
    ``bevformer.modules`` path points at the reference directory) into
    ``sys.modules``.  ``ext_module`` is what ``ext_loader.load_ext`` returns —
    pass an ``_ext``-shaped object to make the reference's autograd Function
    call it (multi_scale_deformable_attn_function.py:10-12)."""
    if "mmcv" in sys.modules and getattr(sys.modules["mmcv"], "__bevformer_amd_stub__", False):
        if ext_module is not None:
            sys.modules["mmcv"].utils.ext_loader._ext = ext_module
        return sys.modules["mmcv"]
    if not reference_available():
        raise RuntimeError("the reference tree is not present on this machine")
    from . import bevformer_cpu

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    regs = {n: _Registry(n) for n in ("ATTENTION", "TRANSFORMER_LAYER", "TRANSFORMER_LAYER_SEQUENCE",
                                      "FEEDFORWARD_NETWORK", "POSITIONAL_ENCODING")}
    transformer_registry = _Registry("Transformer")

    def build_attention(cfg, default_args=None):
        return _build_from_cfg(cfg, regs["ATTENTION"], default_args)

    def build_feedforward_network(cfg, default_args=None):
        return _build_from_cfg(cfg, regs["FEEDFORWARD_NETWORK"], default_args)

    def build_transformer_layer(cfg, default_args=None):
        return _build_from_cfg(cfg, regs["TRANSFORMER_LAYER"], default_args)

    def build_transformer_layer_sequence(cfg, default_args=None):
        return _build_from_cfg(cfg, regs["TRANSFORMER_LAYER_SEQUENCE"], default_args)

    @regs["FEEDFORWARD_NETWORK"].register_module()
    class FFN(_BaseModule):
        # mmcv.cnn.bricks.transformer.FFN restated; parameter names
        # layers.0.0.{weight,bias}, layers.1.{weight,bias}
        def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                     act_cfg=dict(type="ReLU", inplace=True), ffn_drop=0.0,
                     dropout_layer=None, add_identity=True, init_cfg=None, **kwargs):
            super().__init__(init_cfg)
            assert num_fcs >= 2
            self.embed_dims = embed_dims
            layers, cin = [], embed_dims
            for _ in range(num_fcs - 1):
                layers.append(_Sequential(nn.Linear(cin, feedforward_channels),
                                          nn.ReLU(inplace=True), nn.Dropout(ffn_drop)))
                cin = feedforward_channels
            layers.append(nn.Linear(feedforward_channels, embed_dims))
            layers.append(nn.Dropout(ffn_drop))
            self.layers = _Sequential(*layers)
            self.dropout_layer = nn.Identity()
            self.add_identity = add_identity

        def forward(self, x, identity=None):
            out = self.layers(x)
            if not self.add_identity:
                return self.dropout_layer(out)
            if identity is None:
                identity = x
            return identity + self.dropout_layer(out)

    class TransformerLayerSequence(_BaseModule):
        def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
            super().__init__(init_cfg)
            if isinstance(transformerlayers, dict):
                transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
            self.num_layers = num_layers
            self.layers = _ModuleList()
            for i in range(num_layers):
                self.layers.append(build_transformer_layer(transformerlayers[i]))
            self.embed_dims = self.layers[0].embed_dims
            self.pre_norm = self.layers[0].pre_norm

    class _ExtLoader:
        _ext = ext_module

        @classmethod
        def load_ext(cls, name, funcs):
            class _Proxy:
                def __getattr__(self, item):
                    if cls._ext is None:
                        raise RuntimeError("no _ext module installed in the mmcv stub")
                    return getattr(cls._ext, item)
            return _Proxy()

    def build_norm_layer(cfg, num_features, postfix=""):
        assert cfg["type"] == "LN"
        return "ln" + str(postfix), nn.LayerNorm(num_features, eps=cfg.get("eps", 1e-5))

    def build_activation_layer(cfg):
        assert cfg["type"] == "ReLU"
        return nn.ReLU(inplace=cfg.get("inplace", False))

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    class MultiScaleDeformableAttention(_BaseModule):  # import target only
        pass

    mmcv = mod("mmcv", __bevformer_amd_stub__=True, ConfigDict=_ConfigDict,
               deprecated_api_warning=_passthrough_decorator_factory, __path__=[])
    mmcv.utils = mod("mmcv.utils", ext_loader=_ExtLoader, ConfigDict=_ConfigDict,
                     build_from_cfg=_build_from_cfg,
                     deprecated_api_warning=_passthrough_decorator_factory, to_2tuple=to_2tuple,
                     TORCH_VERSION=torch.__version__, digit_version=_digit_version,
                     Registry=_Registry)
    mmcv.ops = mod("mmcv.ops", __path__=[])
    mmcv.ops.multi_scale_deform_attn = mod(
        "mmcv.ops.multi_scale_deform_attn",
        multi_scale_deformable_attn_pytorch=bevformer_cpu.msda_gridsample,
        MultiScaleDeformableAttention=MultiScaleDeformableAttention)
    mmcv.cnn = mod("mmcv.cnn", xavier_init=_xavier_init, constant_init=_constant_init,
                   Linear=nn.Linear, build_activation_layer=build_activation_layer,
                   build_norm_layer=build_norm_layer, __path__=[])
    mmcv.cnn.bricks = mod("mmcv.cnn.bricks", __path__=[])
    mmcv.cnn.bricks.registry = mod("mmcv.cnn.bricks.registry", **regs)
    mmcv.cnn.bricks.transformer = mod(
        "mmcv.cnn.bricks.transformer", build_attention=build_attention,
        build_feedforward_network=build_feedforward_network,
        build_transformer_layer=build_transformer_layer,
        build_transformer_layer_sequence=build_transformer_layer_sequence,
        TransformerLayerSequence=TransformerLayerSequence, FFN=FFN)
    mmcv.runner = mod("mmcv.runner", force_fp32=_passthrough_decorator_factory,
                      auto_fp16=_passthrough_decorator_factory, __path__=[])
    mmcv.runner.base_module = mod("mmcv.runner.base_module", BaseModule=_BaseModule,
                                  ModuleList=_ModuleList, Sequential=_Sequential)

    def run_time(name):
        return lambda fn: fn

    for pkg in ("projects", "projects.mmdet3d_plugin", "projects.mmdet3d_plugin.models",
                "projects.mmdet3d_plugin.models.utils", "projects.mmdet3d_plugin.bevformer"):
        mod(pkg, __path__=[])
    mod("projects.mmdet3d_plugin.models.utils.bricks", run_time=run_time)
    mod("projects.mmdet3d_plugin.models.utils.visual", save_tensor=lambda *a, **k: None)
    mod("projects.mmdet3d_plugin.bevformer.modules", __path__=[_MODULES_DIR])

    # what modules/transformer.py (the encoder's caller) and decoder.py
    # import on top of the above: mmdet's TRANSFORMER registry; torchvision's rotate and the
    # two plotting / image libraries decoder.py imports without using are bound only while
    # those files are being imported (_absent_third_party below), so that nothing else in the
    # process ever sees a fake torchvision / cv2
    for pkg in ("mmdet", "mmdet.models", "mmdet.models.utils"):
        mod(pkg, __path__=[])
    mod("mmdet.models.utils.builder", TRANSFORMER=transformer_registry)
    # transformerV2.py imports mmdet's ResNet blocks and mmcv's conv builder for ResNetFusion
    # (not on the encoder's path): import targets only
    mod("mmdet.models.backbones", __path__=[])
    class BasicBlock(nn.Module):
        """mmdet 2.14 ``mmdet.models.backbones.resnet.BasicBlock`` restated [third party, not on
        disk: unpinned]: conv3x3 - norm - ReLU - conv3x3 - norm, + identity (or ``downsample``),
        ReLU; attribute names ``conv1 / bn1 / conv2 / bn2 / downsample`` as in mmdet (its
        ``norm1_name`` is ``bn1`` for BN-type configs)."""

        def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style="pytorch",
                     with_cp=False, conv_cfg=None, norm_cfg=dict(type="BN"), dcn=None, plugins=None,
                     init_cfg=None):
            super().__init__()
            self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=dilation, dilation=dilation,
                                   bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
            self.relu = nn.ReLU(inplace=True)
            self.downsample = downsample

        def forward(self, x):
            identity = x
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            if self.downsample is not None:
                identity = self.downsample(x)
            return self.relu(out + identity)

    mod("mmdet.models.backbones.resnet", Bottleneck=type("Bottleneck", (nn.Module,), {}),
        BasicBlock=BasicBlock)
    # mmcv's build_conv_layer(None, ...) = nn.Conv2d; build_norm_layer for BN / SyncBN = BatchNorm2d
    sys.modules["mmcv.cnn"].build_conv_layer = lambda cfg, *a, **k: nn.Conv2d(*a, **k)
    _ln_builder = sys.modules["mmcv.cnn"].build_norm_layer

    def _norm_builder(cfg, num_features, postfix=""):
        if cfg is not None and cfg.get("type") in ("BN", "SyncBN", "BN2d"):
            return "bn" + str(postfix), nn.BatchNorm2d(num_features)
        return _ln_builder(cfg, num_features, postfix) if postfix != "" else _ln_builder(cfg, num_features)
    sys.modules["mmcv.cnn"].build_norm_layer = _norm_builder
    @regs["TRANSFORMER_LAYER_SEQUENCE"].register_module()
    class NullDecoder(_BaseModule):
        """Placeholder for ``decoder=dict(type='NullDecoder')``: get_bev_features never
        touches the decoder, but PerceptionTransformer.__init__ builds one
        (transformer.py:54)."""
    return mmcv


class _absent_third_party:
    """While the reference's transformer.py / decoder.py are imported: ``torchvision``'s
    ``rotate`` = the oracle's restatement (torchvision is not installed; bound at import by
    ``from torchvision.transforms.functional import rotate``, transformer.py:18), and empty
    ``cv2`` / ``matplotlib.pyplot`` import targets when those are missing (decoder.py:9,12).
    Everything is removed from ``sys.modules`` again on exit."""

    def __enter__(self):
        from . import bevformer_cpu
        self.added = []

        def add(name, **attrs):
            if name not in sys.modules:
                m = types.ModuleType(name)
                m.__dict__.update(attrs)
                sys.modules[name] = m
                self.added.append(name)
        try:
            import torchvision.transforms.functional  # noqa: F401
        except ImportError:
            add("torchvision", __path__=[])
            add("torchvision.transforms", __path__=[])
            add("torchvision.transforms.functional", rotate=bevformer_cpu.rotate_nearest)
        for name in ("cv2",):
            try:
                importlib.import_module(name)
            except ImportError:
                add(name)
        try:
            importlib.import_module("matplotlib.pyplot")
        except ImportError:
            add("matplotlib", __path__=[])
            add("matplotlib.pyplot")
        return self

    def __exit__(self, *exc):
        for name in self.added:
            sys.modules.pop(name, None)
        return False


def load_reference(ext_module=None):
    """Import the reference's own hot-path files (unmodified) and return a
    namespace with its classes plus the stub's ``build_transformer_layer_sequence``."""
    install(ext_module)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        base = "projects.mmdet3d_plugin.bevformer.modules."
        fn = importlib.import_module(base + "multi_scale_deformable_attn_function")
        sca = importlib.import_module(base + "spatial_cross_attention")
        tsa = importlib.import_module(base + "temporal_self_attention")
        enc = importlib.import_module(base + "encoder")
    ns = types.SimpleNamespace(
        function_module=fn, sca_module=sca, tsa_module=tsa, encoder_module=enc,
        MultiScaleDeformableAttnFunction_fp32=fn.MultiScaleDeformableAttnFunction_fp32,
        SpatialCrossAttention=sca.SpatialCrossAttention,
        MSDeformableAttention3D=sca.MSDeformableAttention3D,
        TemporalSelfAttention=tsa.TemporalSelfAttention,
        BEVFormerEncoder=enc.BEVFormerEncoder, BEVFormerLayer=enc.BEVFormerLayer,
        build_transformer_layer_sequence=sys.modules["mmcv.cnn.bricks.transformer"]
        .build_transformer_layer_sequence)
    return ns


def build_reference_transformer(encoder_cfg, ext_module=None, **kwargs):
    """The reference's own ``PerceptionTransformer`` (modules/transformer.py, unmodified) around
    the reference encoder built from ``encoder_cfg``; ``torchvision...rotate`` is the oracle's
    restatement (torchvision is not installed here)."""
    load_reference(ext_module)
    with warnings.catch_warnings(), _absent_third_party():
        warnings.simplefilter("ignore")
        mod = importlib.import_module("projects.mmdet3d_plugin.bevformer.modules.transformer")
        t = mod.PerceptionTransformer(encoder=copy.deepcopy(encoder_cfg),
                                      decoder=dict(type="NullDecoder"), **kwargs)
    return t.eval()


def build_reference_bev_encoder_v2(encoder_cfg, ext_module=None, **kwargs):
    """The reference's ``PerceptionTransformerBEVEncoder`` (modules/transformerV2.py, unmodified)."""
    load_reference(ext_module)
    with warnings.catch_warnings(), _absent_third_party():
        warnings.simplefilter("ignore")
        mod = importlib.import_module("projects.mmdet3d_plugin.bevformer.modules.transformerV2")
        t = mod.PerceptionTransformerBEVEncoder(encoder=copy.deepcopy(encoder_cfg), **kwargs)
    return t.eval()


def load_reference_transformer_v2(ext_module=None):
    """The reference's modules/transformerV2.py (unmodified) as a module: ``ResNetFusion``,
    ``PerceptionTransformerV2``."""
    load_reference(ext_module)
    with warnings.catch_warnings(), _absent_third_party():
        warnings.simplefilter("ignore")
        return importlib.import_module("projects.mmdet3d_plugin.bevformer.modules.transformerV2")


def load_reference_decoder(ext_module=None):
    """The reference's own decoder.py (unmodified): namespace with
    ``CustomMSDeformableAttention``, ``DetectionTransformerDecoder``, ``inverse_sigmoid`` and the
    stub's builders (its layers can be the reference's ``MyCustomBaseTransformerLayer``)."""
    load_reference(ext_module)
    with warnings.catch_warnings(), _absent_third_party():
        warnings.simplefilter("ignore")
        base = "projects.mmdet3d_plugin.bevformer.modules."
        dec = importlib.import_module(base + "decoder")
        importlib.import_module(base + "custom_base_transformer_layer")
    return types.SimpleNamespace(
        module=dec, CustomMSDeformableAttention=dec.CustomMSDeformableAttention,
        DetectionTransformerDecoder=dec.DetectionTransformerDecoder,
        inverse_sigmoid=dec.inverse_sigmoid,
        build_transformer_layer_sequence=sys.modules["mmcv.cnn.bricks.transformer"]
        .build_transformer_layer_sequence)


def build_reference_encoder(cfg, ext_module=None):
    ns = load_reference(ext_module)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return ns.build_transformer_layer_sequence(copy.deepcopy(cfg)).eval()
