"""Thin functional front of the HIP operator for the modules of this package.

``msda`` is the reference's operator call (``MultiScaleDeformableAttnFunction_fp32
.apply`` at spatial_cross_attention.py:390-392 / temporal_self_attention.py:247-249);
``msda_ragged`` is the same math over a ragged batch (row -> value-batch table),
which is how this package runs SpatialCrossAttention without zero-padded rows.
Neither has a CPU implementation: CPU tensors raise ``RuntimeError``.

Round 5: the module is a package — ``_base`` (modes, cache keys, timers), ``sampling`` (the operators and the row kernels
around them), ``images`` (weight images of the MFMA kernels, their caches, merged / flattened parameters), ``gemm`` (the dense
projections behind ``linear``, ``KERNEL_SELECTION``, the autograd Function), ``chains`` (projection + LayerNorm and the seam
kernels of an encoder layer; round 6 took ``images`` and ``chains`` out of ``gemm``), ``prologue`` (rotation, flattening) —
re-exported here name for name: ``ops.linear``, ``ops.msda_fused`` ... are what the modules call and what the tests
substitute; calls BETWEEN operators go through this namespace too (``_pkg()`` in the submodules).
"""
from . import _base, sampling, images, gemm, chains, prologue

for _mod in (_base, sampling, images, gemm, chains, prologue):
    for _k, _v in vars(_mod).items():
        if not _k.startswith("__") and _k != "_pkg":
            globals()[_k] = _v
del _mod, _k, _v
