"""Shared test helpers (tests may use oracle/; the product package may not)."""
import contextlib

import torch

from bevformer_amd import ops
from oracle import bevformer_cpu as O


def _oracle_msda(value, shapes, start, loc, attn, im2col_step=64, tag=None):
    return O.msda_gridsample(value, shapes, loc, attn)


def _oracle_msda_ragged(value, shapes, start, loc, attn, row_batch, tag=None):
    """Ragged batch through the oracle: one call per value-batch entry."""
    R, M = loc.shape[:2]
    out = value.new_zeros(R, M * value.shape[-1])
    for n in range(value.shape[0]):
        sel = (row_batch == n).nonzero().squeeze(-1)
        if sel.numel():
            out[sel] = O.msda_gridsample(value[n:n + 1], shapes, loc[sel][None], attn[sel][None])[0]
    return out


@contextlib.contextmanager
def oracle_ops():
    """Route the package's operator calls through the CPU oracle so that the
    HOST logic of the modules (ragged rows, merged GEMMs, geometry, plans,
    tiling) can be parity-tested without a GPU.  Test-only: the product path
    itself has no CPU implementation."""
    saved = (ops.msda, ops.msda_ragged)
    ops.msda, ops.msda_ragged = _oracle_msda, _oracle_msda_ragged
    try:
        yield
    finally:
        ops.msda, ops.msda_ragged = saved


def build_pair(name, seed=3, device="cpu"):
    """(product encoder, reference-keyed state_dict with 'trained-like' weights)."""
    import bevformer_amd
    from bevformer_amd import synthetic as S
    torch.manual_seed(0)
    enc = bevformer_amd.build_transformer_layer_sequence(S.encoder_cfg(name)).eval()
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    S.trained_like_(sd, seed=seed)
    enc.load_state_dict(sd)
    return enc.to(device), sd
