"""ctypes binding of libldweaver_amd.so (the C ABI declared in include/ldweaver_amd.h).

There is NO fallback: if the shared library is missing or a call fails, an
exception is raised.  Nothing in this package computes MI, Hamming weights or
the ACGTN2num mask on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LDW_AMD_LIB") or os.path.join(_HERE, "libldweaver_amd.so")  # override for kernel A/B experiments

LDW_OK = 0
LDW_ERR_ARG, LDW_ERR_HIP, LDW_ERR_STATE, LDW_ERR_NOGPU, LDW_ERR_SIZE = 1, 2, 3, 4, 5
QUIRK_REFERENCE, QUIRK_INTENDED = 0, 1
MI_SR_ROWS_STAY = 1
ENGINE_MFMA, ENGINE_HIST, ENGINE_HIST_STATES = 0, 1, 2
COL_INT32, COL_INT64, COL_DOUBLE = 0, 1, 2


class LdwError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libldweaver_amd error {code}: {msg}")
        self.code = code


class MIParams(C.Structure):
    _fields_ = [("sr_dist", C.c_double), ("lr_retain_links", C.c_double), ("lr_links_approx", C.c_double),
                ("sr_only", C.c_int32), ("quirk_mode", C.c_int32), ("keep_sr", C.c_int32), ("flags", C.c_int32)]


_lib = None

_p = C.c_void_p
_i64 = C.c_int64
_SIGS = {
    "ldw_version": (C.c_int, []),
    "ldw_last_error": (C.c_char_p, []),
    "ldw_device_count": (C.c_int, []),
    "ldw_ctx_create": (C.c_int, [C.c_int, C.POINTER(_p)]),
    "ldw_ctx_destroy": (C.c_int, [_p]),
    "ldw_ctx_set_stream": (C.c_int, [_p, _p]),
    "ldw_ctx_sync": (C.c_int, [_p]),
    "ldw_ctx_last_timing": (C.c_int, [_p, _p]),
    "ldw_ctx_counters": (C.c_int, [_p, _p]),
    "ldw_acgtn2num": (C.c_int, [_p, _p, _p, _i64, C.c_int]),
    "ldw_acgtn2num_dev": (C.c_int, [_p, _p, _p, _i64]),
    "ldw_fast_hadamard": (C.c_int, [_p] + [_p] * 8 + [_i64, C.c_int]),
    "ldw_set_alignment": (C.c_int, [_p, _p, _i64, _i64, C.c_int]),
    "ldw_encode_alignment": (C.c_int, [_p, _p, _i64, _i64, _p, _i64, _p]),
    "ldw_alignment_scan": (C.c_int, [_p, _p, _i64, _i64, _p]),
    "ldw_state_counts": (C.c_int, [_p, _p]),
    "ldw_get_alignment": (C.c_int, [_p, _p]),
    "ldw_hamming_weights": (C.c_int, [_p, C.c_int32, _p, _p]),
    "ldw_hamming_counts": (C.c_int, [_p, C.c_int32, C.c_int32, C.c_int32, _p]),
    "ldw_hamming_stats": (C.c_int, [_p, _p]),
    "ldw_set_weights": (C.c_int, [_p, _p, _i64, C.c_int]),
    "ldw_set_snp_meta": (C.c_int, [_p, _p, _p, _p, _p, C.c_double]),
    "ldw_set_engine": (C.c_int, [_p, C.c_int]),
    "ldw_mi_block": (C.c_int, [_p, _p, _i64, _p, _i64, C.c_int, _p, C.c_int]),
    "ldw_joint_tables": (C.c_int, [_p, _p, _p, _i64, _p, _p, _p]),
    "ldw_mi_all_pairs": (C.c_int, [_p, _p, _i64, C.POINTER(MIParams), C.c_int]),
    "ldw_build_info": (C.c_int, []),
    "ldw_mi_all_pairs_multi": (C.c_int, [_p, C.c_int, _p, _i64, C.POINTER(MIParams), _p, _p]),
    "ldw_deal_blocks": (C.c_int, [_p, _i64, C.c_int, _p]),
    "ldw_hamming_weights_multi": (C.c_int, [_p, C.c_int, C.c_int32, _p]),
    "ldw_links_begin": (C.c_int, [_p, _i64]),
    "ldw_mi_block_links": (C.c_int, [_p, _p, _i64, _p, _i64, C.POINTER(MIParams)]),
    "ldw_links_end": (C.c_int, [_p]),
    "ldw_set_overlap": (C.c_int, [_p, C.c_int]),
    "ldw_set_fused": (C.c_int, [_p, C.c_int]),
    "ldw_set_mixed": (C.c_int, [_p, C.c_int]),
    "ldw_ctx_counters2": (C.c_int, [_p, _p]),
    "ldw_set_screen": (C.c_int, [_p, C.c_int]),
    "ldw_set_path": (C.c_int, [_p, C.c_int]),
    "ldw_set_select": (C.c_int, [_p, C.c_int]),
    "ldw_links_device_ptrs": (C.c_int, [_p, C.c_int, _p, _p, _p, _p]),
    "ldw_apx_info": (C.c_int, [_p, _p]),
    "ldw_links_count": (C.c_int, [_p, C.c_int, C.POINTER(_i64)]),
    "ldw_links_fetch": (C.c_int, [_p, C.c_int, _p, _p, _p, _i64, C.c_int]),
    "ldw_block_stats": (C.c_int, [_p, _i64, _p, _p, _p, _p]),
    "ldw_links_import": (C.c_int, [_p, C.c_int, _p, _p, _p, _i64, C.c_int]),
    "ldw_aracne": (C.c_int, [_p, _p, _p, _p, _i64, _p, _p, _p, _i64, _p]),
    "ldw_sr_len_quantiles": (C.c_int, [_p, C.c_int, C.c_double, C.c_double, C.c_int32, _p, _p, _p]),
    "ldw_sr_excess_stats": (C.c_int, [_p, C.c_int, C.c_int32, _p, _p]),
    "ldw_sr_pvalues": (C.c_int, [_p, C.c_int, C.c_int32, _p, _p, C.c_double, _p, _p, _p]),
    "ldw_sr_reduced_fetch": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _p]),
    "ldw_sr_pool_fetch": (C.c_int, [_p, _i64, _p, _p, _p]),
    "ldw_aracne_device": (C.c_int, [_p, _i64, _p]),
    "ldw_sr_tail_extract": (C.c_int, [_p, C.c_int, C.c_int32, _p, _p, _p, _i64, C.c_int, _p]),
    "ldw_sr_quantiles_merge": (C.c_int, [_p, C.c_int, C.c_int32, C.c_double, C.c_int, _p, _p, _p, C.c_int, _p, _p, _p]),
    "ldw_sr_excess_stats_blocks": (C.c_int, [_p, C.c_int, C.c_int32, _p, _i64, _p, _p]),
    "ldw_sr_pool_build": (C.c_int, [_p, C.c_double, _p]),
    "ldw_sr_reduced_import": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _p, _p, _p]),
    "ldw_sr_len_quantiles_multi": (C.c_int, [_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int32, _p, _p, _p]),
    "ldw_sr_excess_stats_multi": (C.c_int, [_p, C.c_int, C.c_int, C.c_int32, _p, _p]),
    "ldw_sr_pvalues_multi": (C.c_int, [_p, C.c_int, C.c_int, C.c_int32, _p, _p, C.c_double, _p, _p, _p]),
    "ldw_gemm_stats": (C.c_int, [_p, _p, C.c_int]),
    "ldw_lr_tukey": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _p, _p, _p, _p, _p]),
    "ldw_lr_reduced_fetch": (C.c_int, [_p, _i64, _p, _p, _p, _p]),
    "ldw_ldmap": (C.c_int, [_p, C.c_int32, C.c_int32, C.c_int32, _p, _p, _p, _p, _i64]),
    "ldw_reset_speculation": (C.c_int, [_p]),
    "ldw_path_report": (C.c_int, [_p, _p, C.c_char_p, C.c_int]),
    "ldw_set_prune": (C.c_int, [_p, C.c_int]),
    "ldw_prune_report": (C.c_int, [_p, _p]),
    "ldw_ctx_reserve": (C.c_int, [_p, C.c_int64, C.c_int64, C.c_int64]),
    "ldw_sr_pairs_fill": (C.c_int, [_p, _p, C.c_int64, C.c_double, _p, _p, C.c_int64, _p]),
    "ldw_set_span": (C.c_int, [_p, C.c_int, C.c_int]),
    "ldw_span_report": (C.c_int, [_p, _p]),
    "ldw_overflow_report": (C.c_int, [_p, _p]),
    "ldw_set_pair_cap": (C.c_int, [C.c_uint32]),
    "ldw_snp_bounds": (C.c_int, [_p, _p, C.c_int64]),
    "ldw_debug_violations": (C.c_int, [_p, _p]),
    "ldw_debug_tab11": (C.c_int, [_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _p, _p]),
    "ldw_debug_apx_params": (C.c_int, [_p, _p, _p, _p, _i64]),
    "ldw_debug_rows": (C.c_int, [_p, _p, _p, _i64]),
    "ldw_debug_apx_gemm": (C.c_int, [_p, _p, C.c_int, _p, C.c_int, _p]),
    "ldw_debug_screen_bound": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "ldw_format_number": (C.c_int, [C.c_double, C.c_char_p, C.c_int]),
    "ldw_r_sample": (C.c_int, [C.c_uint32, C.c_int64, C.c_int64, _p]),
    "ldw_write_table_tsv": (C.c_int, [C.c_char_p, C.c_int, _i64, C.c_int, _p, _p, C.c_int, C.POINTER(_i64)]),
    "ldw_write_links_tsv": (C.c_int, [_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "ldw_write_links_tsv_begin": (C.c_int, [_p, C.c_int, C.c_char_p, C.c_int, C.c_int]),
    "ldw_write_links_tsv_end": (C.c_int, [_p, C.POINTER(_i64), C.POINTER(_i64)]),
    "ldw_tsv_join": (C.c_int, [_p]),
    "ldw_host_trim": (C.c_int, [_p, C.POINTER(_i64)]),
    "ldw_lr_stream_begin": (C.c_int, [_p, C.c_char_p, C.c_int, C.c_int]),
    "ldw_lr_stream_end": (C.c_int, [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "ldw_compare_to_row": (C.c_int, [_p, _i64, _i64, _p, _i64, _p]),
    "ldw_vec_pos_match": (C.c_int, [_p, _i64, _p, _i64, _p]),
    "ldw_compare_triplet": (C.c_int, [_p, _p, _i64, C.c_double, C.POINTER(C.c_int)]),
    "ldw_fast_intersect": (C.c_int, [_p, _i64, _p, _i64, _p, C.POINTER(_i64)]),
}


def declared_symbols():
    return sorted(_SIGS)


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} not found: build it with `make -C ldweaver_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'` — there is no CPU fallback")
        # Load order (r05, found by tests/test_bench_flags.py running first): torch bundles its own HIP runtime; if THIS library — and with it
        # the system's libamdhip64 — is the first of the two in the process, torch's later lazy initialisation finds "No HIP GPUs".  The Python
        # side uses torch for device memory anyway (engine.py), so torch's runtime goes in first.  (A host without torch, R, has one runtime.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def has_experiments() -> bool:
    """The loaded library was built with -DLDW_EXPERIMENTS (make EXPERIMENTS=1; LDW_AMD_LIB selects it)."""
    return bool(lib().ldw_build_info() & 1)


def check(code: int):
    if code != LDW_OK:
        raise LdwError(code, lib().ldw_last_error().decode("utf-8", "replace"))


def ptr(a):
    """Device or host pointer of a numpy array / torch tensor / int / None."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):  # torch tensor
        return C.c_void_p(a.data_ptr())
    raise TypeError(f"cannot take a pointer of {type(a)}")


def as_c(a, dtype, name="array"):
    """C-contiguous numpy array of the given dtype (copy only if needed)."""
    out = np.ascontiguousarray(a, dtype=dtype)
    return out
