"""ctypes binding of libdmpfold_hip.so (C ABI declared in include/dmpfold_hip.h).

There is no fallback: if the shared library is missing or does not export the
expected symbols the import of the product path fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DMPFOLD_HIP_LIB") or os.path.join(_HERE, "libdmpfold_hip.so")

_vp, _i, _i64, _fp = C.c_void_p, C.c_int, C.c_int64, C.c_void_p   # device pointers travel as void*

# name -> (restype, argtypes); mirrors include/dmpfold_hip.h one to one
SIGNATURES = {
    "dmp_abi_version": (_i, []),
    "dmp_last_error": (C.c_char_p, []),
    "dmp_ctx_create": (_i, [_i, _i, _i, C.POINTER(_vp)]),
    "dmp_ctx_destroy": (None, [_vp]),
    "dmp_ctx_set_option": (_i, [_vp, C.c_char_p, _i]),
    "dmp_ctx_get_option": (_i, [_vp, C.c_char_p, C.POINTER(_i)]),
    "dmp_weights_set": (_i, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i]),
    "dmp_weights_finalize": (_i, [_vp]),
    "dmp_weights_share": (_i, [_vp, _vp]),
    "dmp_msa_encode": (_i, [_vp, _i64, _vp]),
    "dmp_msa_weights": (_i, [_vp, _fp, _i, _i, _fp, _vp]),
    "dmp_cov_build": (_i, [_vp, _fp, _fp, _i, _i, _fp, _vp]),
    "dmp_spd_inverse": (_i, [_vp, _fp, _i, _vp]),
    "dmp_dca_contacts": (_i, [_vp, _fp, _i, _fp, _vp]),
    "dmp_dca_features": (_i, [_vp, _fp, _i, _i, _fp, _vp]),
    "dmp_gru_vertical": (_i, [_vp, _fp, _i, _i, _fp, _vp]),
    "dmp_gru_vertical_group": (_i, [C.POINTER(_vp), _i, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), C.POINTER(_vp), _vp]),
    "dmp_gru_bidir": (_i, [_vp, _i, _fp, _i, _fp, _vp]),
    "dmp_stem_static": (_i, [_vp, _fp, _fp, _fp, _i, _fp, _vp]),
    "dmp_stem_update": (_i, [_vp, _fp, _fp, _i, _fp, _vp]),
    "dmp_block_conv5x5_maxout": (_i, [_vp, _i, _fp, _i, _fp, _fp, _vp]),
    "dmp_block_norm_scse_residual": (_i, [_vp, _i, _fp, _fp, _fp, _i, _fp, _vp]),
    "dmp_block_conv5x5_maxout_bwd": (_i, [_vp, _i, _fp, _fp, _vp, _i, _fp, _fp, _fp, _vp]),
    "dmp_block_conv5x5_maxout_winners": (_i, [_vp, _i, _fp, _i, _fp, _vp, _vp]),
    "dmp_block_norm_scse_residual_bwd": (_i, [_vp, _i, _fp, _fp, _i, _fp, _fp, _vp]),
    "dmp_head_conv_bwd": (_i, [_vp, _fp, _fp, _i, _fp, _fp, _vp]),
    "dmp_stem_maxout_winners": (_i, [_vp, _fp, _fp, _i, _fp, _vp, _vp]),
    "dmp_stem_bwd": (_i, [_vp, _fp, _vp, _fp, _fp, _fp, _i, _fp, _fp, _fp, _vp]),
    "dmp_head_gram": (_i, [_vp, _fp, _i, _fp, _fp, _vp]),
    "dmp_trunk_pass": (_i, [_vp, _fp, _fp, _i, _fp, _fp, _vp]),
    "dmp_eigh_top8": (_i, [_vp, _fp, _i, _fp, _vp]),
    "dmp_coords_from_mds": (_i, [_vp, _fp, _fp, _i, _fp, _vp]),
    "dmp_pair_distances": (_i, [_vp, _fp, _i, _i, _fp, _vp]),
    "dmp_refine_coords": (_i, [_vp, _fp, _i, _i, _vp]),
    "dmp_ca_to_backbone": (_i, [_vp, _fp, _fp, _i, _fp, _fp, _vp]),
    "dmp_predict": (_i, [_vp, _fp, _i, _i, _fp, _i, _i, _i, _fp, _fp, _vp]),
    "dmp_predict_end": (_i, [_vp, _fp, _fp, _vp]),
    "dmp_predict_next_unit": (_i, [_vp]),
    "dmp_predict_group_vgru": (_i, [C.POINTER(_vp), _i]),
    "dmp_predict_group_riders": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), C.POINTER(_vp)]),
    "dmp_predict_set_vgru_result": (_i, [_vp, _fp, _vp]),
    "dmp_predict_issue_unit": (_i, [_vp, _vp]),
    "dmp_ctx_pending": (_i, [_vp]),
    "dmp_predict_begin_units": (_i, [_vp, _fp, _i, _i, _fp, _i, _i, _i]),
    "dmp_ctx_share_lane": (_i, [_vp, _vp]),
    "dmp_sync_faults": (_i, [_vp, _vp, C.POINTER(_i)]),
    "dmp_debug_fetch": (_i64, [_vp, C.c_char_p, _fp, _i64, _vp]),
    "dmp_profile_enable": (_i, [_vp, _i, _i]),
    "dmp_profile_conv_intervals": (_i, [_vp, _vp, C.POINTER(C.c_float), C.POINTER(C.c_float), _i, C.POINTER(_i)]),
    "dmp_pipeline_create": (_i, [_i, _i, _i, _i, C.POINTER(_vp)]),
    "dmp_pipeline_create_on": (_i, [_i, _i, _i, _i, C.POINTER(_vp), C.POINTER(_vp)]),
    "dmp_pipeline_destroy": (None, [_vp]),
    "dmp_pipeline_engines": (_i, [_vp]),
    "dmp_pipeline_ctx": (_vp, [_vp, _i]),
    "dmp_pipeline_stream": (_vp, [_vp, _i]),
    "dmp_pipeline_weights_ready": (_i, [_vp]),
    "dmp_pipeline_set_option": (_i, [_vp, C.c_char_p, _i]),
    "dmp_pipeline_submit": (_i64, [_vp, _fp, _i, _i, _fp, _i, _i, _fp, _fp, _vp]),
    "dmp_pipeline_wait": (_i, [_vp, _i]),
    "dmp_pipeline_poll": (_i, [_vp, C.POINTER(_i64), _i, C.POINTER(_i)]),
    "dmp_pipeline_status": (_i, [_vp, _i64, C.POINTER(_i), C.POINTER(_i)]),
    "dmp_pipeline_release": (_i, [_vp, _i64]),
    "dmp_pipeline_backlog": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "dmp_pipeline_stats": (_i, [_vp, C.POINTER(C.c_longlong), _i]),
    "dmp_pipeline_pause": (_i, [_vp, _i]),
}

ABI_VERSION = 5      # include/dmpfold_hip.h DMP_ABI_VERSION

_lib = None


class DmpError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m dmpfold2_amd.build` "
            "(hipcc, gfx950). dmpfold2_amd has no CPU or PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.dmp_abi_version() != ABI_VERSION:
        raise ImportError(f"libdmpfold_hip.so ABI version {lib.dmp_abi_version()} != {ABI_VERSION} "
                          "(stale library: rebuild with `python -m dmpfold2_amd.build`)")
    _lib = lib
    return lib


def check(rc, exc=DmpError):
    if rc is not None and rc < 0:
        msg = load().dmp_last_error().decode("utf-8", "replace")
        raise exc(msg or f"libdmpfold_hip error {rc}")
    return rc
