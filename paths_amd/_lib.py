"""ctypes binding of ``libpaths_hip.so`` (C ABI in include/paths_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel launch fails this
module raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (or
``make -C paths_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PATHS_HIP_LIB") or os.path.join(_HERE, "libpaths_hip.so")

_i64, _i32, _f32, _vp, _u32, _u64 = C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_uint32, C.c_uint64

# name -> argtypes (mirrors include/paths_hip.h; tests/test_cpu_surface.py::test_library_exports_every_declared_symbol checks both against the .so exports)
SIGNATURES = {
    "paths_lstm_cell": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp,
                        _i32, _i32, _i32, _vp, _i32, _i32, _vp],
    "paths_pe_table": [_vp, _i32, _i32, _i32, _vp, _vp],
    "paths_importance_proj": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32,
                              _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "paths_gemm_nt_f32": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _i32, _vp],
    "paths_x6_pack_weights": [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _f32, _vp],
    "paths_x6_pack_weights_t": [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "paths_lstm_cell_x6": [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp,
                           _i32, _i32, _i32, _vp, _i32, _i32, _i32, _f32, _f32, _f32, _vp],
    "paths_importance_proj_x6": [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32,
                                 _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp],
    "paths_importance_qkv_x6": [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                _vp, _vp, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _f32, _f32, _vp, _i32, _i32,
                                _i32, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "paths_gemm_nt_x6": [_vp, _i64, _vp, _i32, _i32, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _i32,
                         _i32, _f32, _f32, _vp],
    "paths_gemm_add_nt_x6": [_vp, _i64, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _f32, _f32, _vp],
    "paths_gemm_tn_f32": [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "paths_attention_fp8": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "paths_gemm_tn_x6": [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp],
    "paths_colsum_f32": [_vp, _i64, _i32, _i32, _vp, _i32, _i32, _vp, _vp],
    "paths_transpose_f32": [_vp, _i64, _i32, _i32, _vp, _i64, _vp],
    "paths_attention_wide_fwd": [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _u64, _f32, _vp, _vp],
    "paths_attention_wide_bwd": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _u64, _f32, _vp, _vp],
    "paths_sibling_sum": [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _i64, _i32, _vp],
    "paths_scatter_kept_rows": [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp],
    "paths_adamw_multi": [_vp, _vp, _vp, _i32, _vp, _vp, _f32, _i32, _f32, _f32, _f32, _f32, _i32, _vp],
    "paths_lstm_bwd_a": [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _i64, _vp, _vp],
    "paths_lstm_bwd_b": [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _i32, _i64, _i32, _vp, _i64, _vp, _i64, _vp],
    "paths_importance_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp],
    "paths_layernorm_fwd_stats": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp],
    "paths_attention_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "paths_attention_token0_any": [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp],
    "paths_attention_token0_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _u64, _f32, _vp],
    "paths_attention_token0_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _u64, _f32, _vp],
    "paths_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "paths_layernorm_bwd_sums": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "paths_reduce_slabs_f32": [_vp, _i32, _i32, _vp, _i32, _vp],
    "paths_defer_reductions": [_i32],
    "paths_flush_reductions": [_vp, _vp],
    "paths_linear_f32": [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp],
    "paths_attention_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "paths_attention_x6": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp],
    "paths_tlayer_pack_h3": [_i32, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp],
    "paths_token_layer_h3": [_vp] * 16 + [_f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _vp, _vp],
    "paths_tlayer_pack_ws": [_i32, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _i32, _vp],
    "paths_token_layer_ws": [_vp] * 17 + [_f32, _f32, _f32, _f32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _i32, _vp],
    "paths_token_layer_ws_rows": [_vp] * 16 + [_f32, _f32, _f32, _f32, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp],
    "paths_token0_pack_ws": [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp],
    "paths_token0_pack_ws_d": [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _i32, _vp, _vp],
    "paths_token0_tail_ws": [_vp] * 16 + [_vp, _i64, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _vp],
    "paths_attention_any": [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp],
    "paths_attention_any_train": [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _u64, _f32, _vp],
    "paths_layernorm_fwd_stats_any": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp],
    "paths_layernorm_bwd_any": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "paths_layernorm_bwd_sums_any": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "paths_importance_bwd_any": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _i64, _vp, _vp, _vp, _vp],
    "paths_importance_rows_bwd_any": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp],
    "paths_attention_bwd_any": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _u64, _f32, _vp],
    "paths_fp8_scale": [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _i32, _vp],
    "paths_fp8_pack_weight": [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp],
    "paths_fp8_quantize": [_vp, _i64, _i32, _i32, _vp, _vp, _vp],
    "paths_gemm_nt_fp8_out8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "paths_gemm_nt_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp],
    "paths_attention_fp8_qkv": [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp],
    "paths_attention_h3_any": [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp],
    "paths_attention_h3_any_img": [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "paths_layernorm_rows": [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _f32, _vp],
    "paths_layernorm2_rows": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _f32, _vp],
    "paths_importance_rows": [_vp, _i64, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _i32, _vp],
    "paths_importance_tokens_rows": [_vp, _i64, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp],
    "paths_tokens_assemble": [_vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "paths_final_head_any": [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _f32, _vp],
    "paths_attention_h3_img": [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "paths_token_layer_f32": [_vp] * 21 + [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _vp],
    "paths_token0_tail": [_vp] * 20 + [_vp, _i64, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _vp],
    "paths_final_head": [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _f32, _vp],
    "paths_layernorm_f32": [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp],
    "paths_topk": [_vp, _i64, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp],
    "paths_topk_rows": [_vp, _i64, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp],
    "paths_gemm_rows_nt_x6": [_vp, _vp, _i32, _i32, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _f32, _vp],
    "paths_expand_children": [_vp, _i64, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "paths_gather_kept_rows": [_vp, _i64, _i64, _vp, _i64, _vp, _i32, _i32, _vp, _vp],
    "paths_gather_rows_bwd": [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _vp],
    "paths_fallback_all_cells": [_vp, _vp, _vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "paths_gather_rows": [_vp, _vp, _i32, _vp, _i64, _i64, _vp, _i32, _vp, _i32, _i64, _vp, _vp, _i32, _vp, _vp, _vp],
    "paths_level0_batch": [_vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp],
    "paths_scale_add_rows": [_vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _vp, _vp],
    "paths_importance_rows_bwd": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp],
    "paths_dropout_rows": [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _u64, _f32, _vp],
    "paths_dropout_mask": [_vp, _i64, _u64, _f32, _vp],
    "paths_attention_x6_dropout": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _u64, _f32, _vp],
    "paths_attention_bwd_f32_dropout": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _u64, _f32, _vp],
    "paths_attention_bwd_x6_dropout": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _u64, _f32, _vp],
    "paths_attention_bwd_x6_planes": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _u64, _f32, _i32, _vp],
    "paths_stream_wait": [_vp, _vp, _vp],
    "paths_set_stop_event": [_vp],
    "paths_flush_stop_event": [_vp],
    "paths_stream_wait_event": [_vp, _vp],
    "paths_record_event": [_vp, _vp],
    "paths_event_destroy": [_vp],
    "paths_memset_zero": [_vp, C.c_size_t, _vp],
    "paths_tissue_mask": [_vp, _i64, _i32, _vp, _vp],
    "paths_tissue_mask_absmax": [_vp, _i64, _i32, _vp, _vp, _vp],
    "paths_synth_grid": [_vp, _i32, _i32, _i32, _u32, _i32, _u64, _vp],
}
_PLAIN = {"paths_gemm_tn_workspace": (C.c_int64, [_i32, _i32, _i32]), "paths_x6_packed_bytes": (C.c_int64, [_i32, _i32, _i32]), "paths_tlayer_h3_image_bytes": (C.c_int64, [_i32]), "paths_tlayer_ws_image_bytes": (C.c_int64, [_i32, _i32]), "paths_token0_ws_image_bytes": (C.c_int64, []), "paths_token0_ws_partials": (C.c_int64, [_i32, _i32]), "paths_token0_ws_image_bytes_d": (C.c_int64, [_i32]), "paths_token0_ws_partials_d": (C.c_int64, [_i32, _i32, _i32]), "paths_token0_ws_supported": (_i32, [_i32, _i32, _i32, _i32]), "paths_attention_x6_workspace": (C.c_int64, [_i32, _i32, _i32, _i32, _i32]), "paths_attention_fp8_workspace": (C.c_int64, [_i32, _i32, _i32, _i32]), "paths_attention_bwd_x6_workspace": (C.c_int64, [_i32, _i32, _i32, _i32]), "paths_attention_token0_workspace": (C.c_int64, [_i32, _i32, _i32]), "paths_attention_h3_any_workspace": (C.c_int64, [_i32, _i32, _i32, _i32]), "paths_importance_proj_x6_workspace": (C.c_int64, [_i32]), "paths_last_error": (C.c_char_p, []), "paths_build_info": (C.c_char_p, []), "paths_abi_version": (_i32, []), "paths_stop_event_pending": (_i32, []), "paths_clear_stop_event": (_i32, []), "paths_adamw_chunk": (_i32, []), "paths_attention_wide_workspace": (C.c_int64, [_i32, _i32]), "paths_event_create": (_vp, []), "paths_stream_create_masked": (_vp, [_vp, _i32])}

ABI_VERSION = 2     # include/paths_hip.h: paths_abi_version() of the library this binding was written against
_lib: Optional[C.CDLL] = None


class PathsHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise PathsHipError(f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build()); "
                                "paths_amd has no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        lib.paths_abi_version.restype = _i32
        if lib.paths_abi_version() != ABI_VERSION:
            raise PathsHipError(f"{LIB_PATH} has ABI version {lib.paths_abi_version()}, this binding needs {ABI_VERSION}: rebuild it "
                                "(__graft_entry__.build())")
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, _i32
        for name, (res, args) in _PLAIN.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, res
        _lib = lib
    return _lib


def ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    return t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)
_FAST_STREAM = os.environ.get("PATHS_FAST_STREAM", "1") != "0"


def stream() -> int:
    """Handle of torch's current stream on the current device.  Called once per launch (~600 times per training step): the public
    ``torch.cuda.current_stream()`` builds a Stream object through four Python layers (5 ms per step measured under cProfile); the
    two C bindings below return the same handle directly."""
    if _RAW_STREAM is not None and _GET_DEVICE is not None and _FAST_STREAM:
        return _RAW_STREAM(_GET_DEVICE())
    return torch.cuda.current_stream().cuda_stream


TAPE = None        # a list while paths_amd.utils.TapedRecursion records: every launch is executed AND appended as (fn, args, name)
TAPE_EVENTS = None # the recorder's event pool: [events, next index]; events are created once per tape position and re-used when a
                   # tape is recorded again (a weight changed), so re-recording does not leak HIP events


def call(name: str, *args):
    lib = load()
    fn = getattr(lib, name)
    rc = fn(*args)
    if rc != 0:
        raise PathsHipError(f"{name} failed ({rc}): {lib.paths_last_error().decode()}")
    if TAPE is not None:
        TAPE.append((fn, args, name))
        fb = fork_behind.active
        if fb is not None and fb.taken_at is None and name != "paths_set_stop_event" and not lib.paths_stop_event_pending():
            fb.taken_at = len(TAPE) - 1                  # this call's kernel carries the block's stop event


def stream_wait(dst: "torch.cuda.Stream", src: "torch.cuda.Stream"):
    """``dst.wait_stream(src)``; recorded on the launch tape (as a paths_stream_wait with its own event) when one is being built."""
    dst.wait_stream(src)
    if TAPE is not None:
        lib = load()
        pool = TAPE_EVENTS if TAPE_EVENTS is not None else [[], 0]
        if pool[1] == len(pool[0]):
            ev = lib.paths_event_create()
            if not ev:
                raise PathsHipError("paths_event_create failed")
            pool[0].append(ev)
        ev = pool[0][pool[1]]
        pool[1] += 1
        TAPE.append((lib.paths_stream_wait, (dst.cuda_stream, src.cuda_stream, ev), "paths_stream_wait"))


STOP_EVENTS = __import__("os").environ.get("PATHS_STOP_EVENTS", "1") != "0"


# entries of a launch tape that are not kernel launches on the block's stream (fork_behind looks for the block's LAST launch)
_TAPE_PLUMBING = {"paths_set_stop_event", "paths_flush_stop_event", "paths_stream_wait_event", "paths_stream_wait", "paths_record_event",
                  "paths_clear_stop_event"}


class fork_behind:
    """``with fork_behind([dst...], src): <launches on src>`` - afterwards every ``dst`` waits for what ran on ``src``.  While a
    launch tape is recorded the join travels as a STOP EVENT of the block's stop-capable kernel (include/paths_hip.h:
    paths_set_stop_event; the finish kernel of the importance / projection GEMM, the top-K kernel) instead of an event record behind it;
    otherwise (eager launches, or PATHS_STOP_EVENTS=0) it is :func:`stream_wait` per destination.

    The stop event covers the block only if the kernel that took it is the block's LAST launch on ``src``.  That is checked, not
    assumed: the C call that owns the stop-capable kernel is the one during which ``paths_stop_event_pending()`` falls to 0
    (:func:`call` notes it); if any launch was recorded after it the event is recorded again behind the block the ordinary way
    (``paths_record_event``: waiters then wait for the later record), so a launch added behind the finish / top-K kernel cannot let
    ``dst`` start early on a replayed tape."""

    active = None        # the block being recorded (call() reports to it)

    def __init__(self, dsts, src):
        self.dsts, self.src = list(dsts), src
        self.ev = None
        self.taken_at = None       # tape index of the call whose kernel took the event

    def __enter__(self):
        if TAPE is not None and STOP_EVENTS:
            lib = load()
            pool = TAPE_EVENTS if TAPE_EVENTS is not None else [[], 0]
            if pool[1] == len(pool[0]):
                e = lib.paths_event_create()
                if not e:
                    raise PathsHipError("paths_event_create failed")
                pool[0].append(e)
            self.ev = pool[0][pool[1]]
            pool[1] += 1
            call("paths_set_stop_event", self.ev)
            self.prev, fork_behind.active = fork_behind.active, self
        return self

    def __exit__(self, et, ev, tb):
        if self.ev is not None:
            fork_behind.active = self.prev
            if et is not None:
                load().paths_clear_stop_event()         # an exception inside the block: leave nothing armed for an unrelated launch
                return False
            call("paths_flush_stop_event", self.src.cuda_stream)
            if self.taken_at is not None and any(nm not in _TAPE_PLUMBING for _, _, nm in TAPE[self.taken_at + 1:]):
                call("paths_record_event", self.ev, self.src.cuda_stream)     # launches followed the stop-capable kernel: cover them
            for d in self.dsts:
                call("paths_stream_wait_event", d.cuda_stream, self.ev)
        elif et is None:
            for d in self.dsts:
                stream_wait(d, self.src)
        return False


def zeros(shape, **kw) -> torch.Tensor:
    """``torch.zeros`` whose fill is repeated on every replay of a launch tape (status words, importance of padding rows)."""
    t = torch.zeros(shape, **kw)
    if TAPE is not None:
        TAPE.append((load().paths_memset_zero, (t.data_ptr(), t.numel() * t.element_size(), stream()), "paths_memset_zero"))
    return t


def require_cuda(*tensors: torch.Tensor):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise PathsHipError("paths_amd runs on the GPU only: got a CPU tensor (no CPU fallback)")


def float_from_bits(bits: int) -> float:
    """fp32 value of a bit pattern written by paths_tissue_mask_absmax; any NaN pattern (above +inf) reads as inf."""
    import struct
    bits &= 0xFFFFFFFF
    return float("inf") if bits >= 0x7F800000 else struct.unpack("<f", struct.pack("<I", bits))[0]


def absmax(t: torch.Tensor) -> float:
    """max|x| of a contiguous fp32 device tensor (inf if not finite) through the same kernel; one host sync."""
    require_cuda(t)
    assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() % 4 == 0
    bits = torch.zeros((1,), dtype=torch.int32, device=t.device)
    D = t.shape[-1] if (t.dim() > 1 and t.shape[-1] % 4 == 0) else 4
    call("paths_tissue_mask_absmax", t.data_ptr(), t.numel() // D, D, None, bits.data_ptr(), stream())
    return float_from_bits(int(bits.item()))
