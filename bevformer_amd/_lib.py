"""ctypes binding of ``libbevmsda.so`` (C ABI in include/bevmsda.h).

This is the only place the product path touches native code.  There is no CPU
fallback: if the shared object is missing or does not export the ABI the import
of anything that computes fails loudly.
"""
import ctypes
import os

from .build import LIB_PATH

_c_int = ctypes.c_int
_c_void_p = ctypes.c_void_p

ABI_VERSION = 3


class Tuning(ctypes.Structure):
    """Mirror of ``struct bevmsda_tuning``."""
    _fields_ = [("variant", ctypes.c_int32), ("qtile", ctypes.c_int32),
                ("xcd_remap", ctypes.c_int32), ("reserved", ctypes.c_int32 * 5)]


class FusedDesc(ctypes.Structure):
    """Mirror of ``struct bevmsda_fused_desc``."""
    _fields_ = [("R", ctypes.c_int64), ("proj_row", ctypes.c_int64),
                ("N", ctypes.c_int32), ("S", ctypes.c_int32), ("M", ctypes.c_int32),
                ("D", ctypes.c_int32), ("L", ctypes.c_int32), ("P", ctypes.c_int32),
                ("Q", ctypes.c_int32), ("K", ctypes.c_int32), ("A", ctypes.c_int32),
                ("ref_mode", ctypes.c_int32), ("off_head", ctypes.c_int32),
                ("off_k", ctypes.c_int32), ("lg_head", ctypes.c_int32), ("lg_k", ctypes.c_int32),
                ("vmul", ctypes.c_int32), ("vadd", ctypes.c_int32),
                ("reserved", ctypes.c_int32 * 6)]


class LocSource(ctypes.Structure):
    """Mirror of ``struct bevmsda_loc_source``."""
    _fields_ = [("offs", ctypes.c_void_p), ("ref", ctypes.c_void_p), ("row_src", ctypes.c_void_p),
                ("proj_row", ctypes.c_int64), ("off_head", ctypes.c_int32), ("A", ctypes.c_int32)]


class LinearDesc(ctypes.Structure):
    """Mirror of ``struct bevmsda_linear_desc``."""
    _fields_ = [("M", ctypes.c_int64), ("ldx0", ctypes.c_int64), ("lda0", ctypes.c_int64),
                ("ldx1", ctypes.c_int64), ("lda1", ctypes.c_int64), ("ldw", ctypes.c_int64),
                ("ldy", ctypes.c_int64), ("N", ctypes.c_int32), ("K0", ctypes.c_int32),
                ("K1", ctypes.c_int32), ("relu", ctypes.c_int32), ("precision", ctypes.c_int32),
                ("variant", ctypes.c_int32), ("group_cols", ctypes.c_int32),
                ("out_bf16", ctypes.c_int32), ("reserved", ctypes.c_int32 * 4)]


class LayerNormDesc(ctypes.Structure):
    """Mirror of ``struct bevmsda_layernorm_desc``."""
    _fields_ = [("res", ctypes.c_void_p), ("ldres", ctypes.c_int64), ("gamma", ctypes.c_void_p),
                ("beta", ctypes.c_void_p), ("eps", ctypes.c_float), ("reserved", ctypes.c_int32 * 3)]


class ChainDesc(ctypes.Structure):
    """Mirror of ``struct bevmsda_chain_desc``."""
    _fields_ = [("M", ctypes.c_int64), ("ld_rows", ctypes.c_int64), ("ld_res", ctypes.c_int64), ("ld_y", ctypes.c_int64),
                ("C", ctypes.c_int32), ("F", ctypes.c_int32), ("precision", ctypes.c_int32), ("eps0", ctypes.c_float),
                ("eps1", ctypes.c_float), ("reserved", ctypes.c_int32 * 5)]


class WgradProblem(ctypes.Structure):
    """Mirror of ``struct bevmsda_wgrad_problem``."""
    _fields_ = [("g", ctypes.c_void_p), ("ldg", ctypes.c_int64), ("x", ctypes.c_void_p), ("ldx", ctypes.c_int64),
                ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("grad_w", ctypes.c_void_p), ("ldgw", ctypes.c_int64),
                ("grad_b", ctypes.c_void_p)]


class PlanDesc(ctypes.Structure):
    """Mirror of ``struct bevmsda_plan_desc``."""
    _fields_ = [("B", ctypes.c_int32), ("Nc", ctypes.c_int32), ("Q", ctypes.c_int32),
                ("D", ctypes.c_int32), ("pc_range", ctypes.c_double * 6), ("img_w", ctypes.c_float),
                ("img_h", ctypes.c_float), ("q_lo", ctypes.c_int32), ("q_hi", ctypes.c_int32),
                ("row_capacity", ctypes.c_int32), ("reserved", ctypes.c_int32 * 5)]


ERR_UNSUPPORTED = -7
ERR_MISALIGNED = -4
ERR_TOO_LARGE = -3
_DIMS = [_c_int] * 7
# name -> argtypes; every symbol the header declares is listed (tests check it)
SIGNATURES = {
    "bevmsda_abi_version": ([], _c_int),
    "bevmsda_error_string": ([_c_int], ctypes.c_char_p),
    "bevmsda_forward_f32": ([_c_void_p] * 5 + _DIMS + [_c_void_p, _c_void_p], _c_int),
    "bevmsda_backward_f32": ([_c_void_p] * 6 + _DIMS + [_c_void_p] * 4, _c_int),
    "bevmsda_forward_bf16": ([_c_void_p] * 5 + _DIMS + [_c_void_p, _c_void_p], _c_int),
    "bevmsda_backward_bf16": ([_c_void_p] * 6 + _DIMS + [_c_void_p] * 4, _c_int),
    "bevmsda_forward_ragged_f32": ([_c_void_p] * 6 + _DIMS + [_c_void_p, _c_void_p], _c_int),
    "bevmsda_backward_ragged_f32": ([_c_void_p] * 7 + _DIMS + [_c_void_p] * 4, _c_int),
    "bevmsda_forward_ragged_bf16": ([_c_void_p] * 6 + _DIMS + [_c_void_p, _c_void_p], _c_int),
    "bevmsda_backward_ragged_bf16": ([_c_void_p] * 7 + _DIMS + [_c_void_p] * 4, _c_int),
    "bevmsda_backward_shared_f32": ([_c_void_p] * 6 + [ctypes.c_int64, ctypes.c_float] + _DIMS + [_c_void_p, ctypes.c_int64]
                                    + [_c_void_p] * 3, _c_int),
    "bevmsda_backward_shared_bf16": ([_c_void_p] * 6 + [ctypes.c_int64, ctypes.c_float] + _DIMS + [_c_void_p, ctypes.c_int64]
                                    + [_c_void_p] * 3, _c_int),
    "bevmsda_backward_rows_f32": ([_c_void_p] * 8 + _DIMS + [_c_void_p, ctypes.c_int64] + [_c_void_p] * 3, _c_int),
    "bevmsda_backward_rows_bf16": ([_c_void_p] * 8 + _DIMS + [_c_void_p, ctypes.c_int64] + [_c_void_p] * 3, _c_int),
    "bevmsda_backward_rows_offs_f32": ([_c_void_p] * 3 + [ctypes.POINTER(LocSource)] + [_c_void_p] * 4 + _DIMS
                                       + [_c_void_p, ctypes.c_int64] + [_c_void_p] * 3, _c_int),
    "bevmsda_backward_rows_offs_bf16": ([_c_void_p] * 3 + [ctypes.POINTER(LocSource)] + [_c_void_p] * 4 + _DIMS
                                        + [_c_void_p, ctypes.c_int64] + [_c_void_p] * 3, _c_int),
    "bevmsda_frontend_expand_rows_f32": ([_c_void_p] * 7 + [ctypes.POINTER(FusedDesc)] + [_c_void_p] * 4, _c_int),
    "bevmsda_cast_rows_bf16": ([_c_void_p, _c_void_p, ctypes.c_int64, _c_int, ctypes.c_float, _c_void_p, _c_void_p], _c_int),
    "bevmsda_rows_from_slots_f32": ([_c_void_p, ctypes.c_int64, _c_void_p, _c_void_p, _c_void_p, ctypes.c_int64, _c_int,
                                     _c_void_p, _c_void_p], _c_int),
    "bevmsda_proj_ln_proj_chain_backward_f32": ([_c_void_p, ctypes.c_int64] + [_c_void_p] * 5 + [ctypes.POINTER(ChainDesc)] + [_c_void_p] * 6,
                                                _c_int),
    "bevmsda_proj_ffn_chain_backward_f32": ([_c_void_p, ctypes.c_int64] + [_c_void_p] * 8 + [ctypes.POINTER(ChainDesc)] + [_c_void_p] * 6
                                            + [_c_void_p, _c_void_p, ctypes.c_float, _c_void_p, _c_void_p],
                                            _c_int),
    "bevmsda_proj_ffn_chain_train_f32": ([_c_void_p] * 14 + [ctypes.POINTER(ChainDesc)] + [_c_void_p] * 9, _c_int),
    "bevmsda_proj_ln_proj_chain_train_f32": ([_c_void_p] * 8 + [ctypes.POINTER(ChainDesc)] + [_c_void_p] * 5, _c_int),
    "bevmsda_forward_f32_ex": ([_c_void_p] * 5 + _DIMS + [_c_void_p, _c_void_p,
                                                            ctypes.POINTER(Tuning)], _c_int),
    "bevmsda_backward_f32_ex": ([_c_void_p] * 6 + _DIMS + [_c_void_p] * 4
                                + [ctypes.POINTER(Tuning)], _c_int),
    "bevmsda_fused_forward_f32": ([_c_void_p] * 8 + [ctypes.POINTER(FusedDesc), _c_void_p, _c_void_p],
                                  _c_int),
    "bevmsda_fused_forward_bf16": ([_c_void_p] * 8 + [ctypes.POINTER(FusedDesc), _c_void_p, _c_void_p],
                                   _c_int),
    "bevmsda_fused_forward_rows_f32": ([_c_void_p] * 9 + [ctypes.POINTER(FusedDesc), _c_void_p, _c_void_p],
                                       _c_int),
    "bevmsda_fused_forward_rows_bf16": ([_c_void_p] * 9 + [ctypes.POINTER(FusedDesc), _c_void_p, _c_void_p],
                                        _c_int),
    "bevmsda_fused_forward_rows_save_f32": ([_c_void_p] * 9 + [ctypes.POINTER(FusedDesc)] + [_c_void_p] * 4, _c_int),
    "bevmsda_fused_forward_rows_save_bf16": ([_c_void_p] * 9 + [ctypes.POINTER(FusedDesc)] + [_c_void_p] * 4, _c_int),
    "bevmsda_frontend_expand_f32": ([_c_void_p] * 6 + [ctypes.POINTER(FusedDesc)] + [_c_void_p] * 4, _c_int),
    "bevmsda_frontend_chain_f32": ([_c_void_p] * 5 + [ctypes.POINTER(FusedDesc)] + [_c_void_p] * 3, _c_int),
    "bevmsda_frontend_chain_gather_f32": ([_c_void_p] * 4 + [ctypes.c_int64, _c_int, _c_void_p, _c_void_p,
                                           ctypes.POINTER(FusedDesc)] + [_c_void_p] * 3, _c_int),
    "bevmsda_frame_plan_counters": ([_c_int, _c_int], ctypes.c_int64),
    "bevmsda_frame_plan_scratch": ([_c_int, _c_int], ctypes.c_int64),
    "bevmsda_frame_plan_f32": ([_c_void_p] * 3 + [ctypes.POINTER(PlanDesc)] + [_c_void_p] * 12, _c_int),
    "bevmsda_fold_extra_rows_f32": ([_c_void_p, ctypes.c_int64, _c_void_p, ctypes.c_int64, _c_int, _c_int,
                                     _c_void_p, _c_void_p], _c_int),
    "bevmsda_add_layernorm_f32": ([_c_void_p] * 4 + [ctypes.c_float, ctypes.c_int64, _c_int,
                                                     _c_void_p, _c_void_p], _c_int),
    "bevmsda_add_layernorm_backward_partials": ([ctypes.c_int64], ctypes.c_int64),
    "bevmsda_add_layernorm_backward_f32": ([_c_void_p] * 4 + [ctypes.c_float, ctypes.c_int64, _c_int] +
                                           [_c_void_p] * 4, _c_int),
    "bevmsda_gather_mean_f32": ([_c_void_p] * 3 + [ctypes.c_int64, _c_int, _c_int,
                                                   _c_void_p, _c_void_p], _c_int),
    "bevmsda_linear_f32": ([_c_void_p] * 6 + [ctypes.POINTER(LinearDesc), _c_void_p, _c_void_p], _c_int),
    "bevmsda_linear_packed_f32": ([_c_void_p] * 6 + [ctypes.POINTER(LinearDesc), _c_void_p, _c_void_p],
                                  _c_int),
    "bevmsda_linear_relu_backward_packed_f32": ([_c_void_p, _c_void_p, _c_void_p, ctypes.c_int64, ctypes.c_float,
                                                 ctypes.POINTER(LinearDesc), _c_void_p, _c_void_p], _c_int),
    "bevmsda_linear_gather_packed_f32": ([_c_void_p, ctypes.c_int64, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                          ctypes.POINTER(LinearDesc), _c_void_p, _c_void_p], _c_int),
    "bevmsda_linear_panel_packed_bytes": ([_c_int, _c_int], ctypes.c_int64),
    "bevmsda_linear_panel_pack_weight_f32": ([_c_void_p, ctypes.c_int64, _c_int, _c_int, _c_void_p, _c_void_p],
                                             _c_int),
    "bevmsda_linear_panel_pack_weight_t_f32": ([_c_void_p, ctypes.c_int64, _c_int, _c_int, _c_void_p, _c_void_p],
                                             _c_int),
    "bevmsda_linear_pack_job_blocks": ([_c_int, _c_int, _c_int], ctypes.c_int64),
    "bevmsda_linear_pack_weights_multi_f32": ([_c_void_p, _c_int, ctypes.c_int64, _c_void_p], _c_int),
    "bevmsda_linear_panel_f32": ([_c_void_p] * 8 + [ctypes.POINTER(LinearDesc), ctypes.POINTER(LayerNormDesc),
                                                  _c_void_p, _c_void_p], _c_int),
    "bevmsda_linear_panel_rows2_f32": ([_c_void_p, _c_void_p, ctypes.c_int64, _c_void_p, _c_void_p, ctypes.POINTER(LinearDesc), _c_void_p,
                                        _c_void_p], _c_int),
    "bevmsda_linear_panel_segments_f32": ([_c_void_p] * 3 + [ctypes.POINTER(LinearDesc), _c_void_p, ctypes.c_int64, _c_void_p,
                                           _c_int, _c_void_p, _c_void_p], _c_int),
    "bevmsda_proj_ffn_chain_f32": ([_c_void_p] * 14 + [ctypes.POINTER(ChainDesc), _c_void_p, _c_void_p], _c_int),
    "bevmsda_proj_ffn_chain_tail_f32": ([_c_void_p] * 14 + [ctypes.POINTER(ChainDesc), _c_void_p, _c_void_p, ctypes.c_int64,
                                         _c_void_p, ctypes.c_int64, _c_void_p, _c_void_p, _c_int, _c_void_p, ctypes.c_int64,
                                         _c_void_p], _c_int),
    "bevmsda_proj_ln_proj_chain_f32": ([_c_void_p] * 10 + [ctypes.POINTER(ChainDesc), _c_void_p, _c_void_p, _c_void_p], _c_int),
    "bevmsda_linear_wgrad_f32": ([_c_void_p, ctypes.c_int64, _c_void_p, ctypes.c_int64, ctypes.c_int64, _c_int, _c_int,
                                  _c_void_p, ctypes.c_int64, _c_void_p, _c_int, _c_void_p], _c_int),
    "bevmsda_linear_wgrad_multi_f32": ([ctypes.POINTER(WgradProblem), _c_int, ctypes.c_int64, _c_int, _c_int, _c_int, _c_void_p],
                                       _c_int),
    "bevmsda_add_layernorm_backward2_f32": ([_c_void_p] * 5 + [ctypes.c_float, ctypes.c_int64, _c_int] + [_c_void_p] * 4,
                                            _c_int),
    "bevmsda_linear_packed_bytes": ([_c_int, _c_int], ctypes.c_int64),
    "bevmsda_linear_pack_weight_f32": ([_c_void_p, ctypes.c_int64, _c_int, _c_int, _c_void_p, _c_void_p],
                                       _c_int),
    "bevmsda_linear_pack_weight_t_f32": ([_c_void_p, ctypes.c_int64, _c_int, _c_int, _c_void_p, _c_void_p],
                                         _c_int),
    "bevmsda_rotate_bev_f32": ([_c_void_p, ctypes.c_int64, _c_void_p, ctypes.c_int64, _c_int, _c_int, _c_int,
                                ctypes.POINTER(ctypes.c_float), _c_void_p], _c_int),
    "bevmsda_rotate_bev_dev_f32": ([_c_void_p, ctypes.c_int64, _c_void_p, ctypes.c_int64, _c_int, _c_int, _c_int,
                                _c_void_p, _c_void_p], _c_int),
    "bevmsda_flatten_feats_f32": ([_c_void_p] * 4 + [_c_int] * 6 + [_c_void_p], _c_int),
    "bevmsda_forward_bf16_ex": ([_c_void_p] * 5 + _DIMS + [_c_void_p, _c_void_p,
                                                             ctypes.POINTER(Tuning)], _c_int),
    "bevmsda_backward_bf16_ex": ([_c_void_p] * 6 + _DIMS + [_c_void_p] * 4
                                 + [ctypes.POINTER(Tuning)], _c_int),
}

_lib = None


class BevMsdaError(RuntimeError):
    pass


def load(path=None):
    """Load (once) and return the ctypes handle; raises if unavailable."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("BEVMSDA_LIBRARY", LIB_PATH)
    if not os.path.exists(p):
        raise BevMsdaError(
            f"{p} not found: the HIP library has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); "
            "there is no CPU fallback for this path.")
    lib = ctypes.CDLL(p)
    for name, (argtypes, restype) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise BevMsdaError(f"{p} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = restype
    if lib.bevmsda_abi_version() != ABI_VERSION:
        raise BevMsdaError(f"{p}: ABI version {lib.bevmsda_abi_version()} != {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().bevmsda_error_string(code).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")
