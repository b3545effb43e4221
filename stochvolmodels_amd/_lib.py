"""
ctypes binding of libsvmc.so (C ABI: include/svmc.h).  The product path has NO CPU fallback: if the
library is missing or cannot be loaded, importing any compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
# SVMC_LIB: an alternative build of the same ABI (A/B runs of tools/ubench/build_variants.sh); default: the in-tree library
LIB_PATH = os.environ.get("SVMC_LIB") or os.path.join(_PKG, "libsvmc.so")

OK, ERR_INVALID_ARGUMENT, ERR_HIP, ERR_UNKNOWN_PAYOFF, ERR_UNSUPPORTED_VARIABLE, ERR_WORKSPACE, ERR_RCCL = range(7)

# svmc_all_reduce_fn (include/svmc.h): int fn(void *user, double *device_buf, size_t n, svmc_stream_t stream)
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

_lock = threading.Lock()
_lib = None


class SvmcError(RuntimeError):
    """a libsvmc call failed (HIP error, bad argument, workspace)."""


def _declare(L: C.CDLL) -> None:
    vp, sz, i32, u32, u64, f64 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_uint64, C.c_double
    pi32, psz, pvp, pf = C.POINTER(i32), C.POINTER(sz), C.POINTER(vp), C.POINTER(C.c_float)
    pf64, pi8 = C.POINTER(f64), C.POINTER(C.c_int8)
    sig = {
        "svmc_version": ([], i32),
        "svmc_rng_stream_version": ([], i32),
        "svmc_last_error": ([], C.c_char_p),
        "svmc_device_count": ([pi32], i32),
        "svmc_set_device": ([i32], i32),
        "svmc_get_device": ([pi32], i32),
        "svmc_device_info": ([i32, C.c_char_p, sz, pi32, pi32, psz], i32),
        "svmc_malloc": ([pvp, sz], i32),
        "svmc_free": ([vp], i32),
        "svmc_host_alloc": ([pvp, sz], i32),
        "svmc_host_free": ([vp], i32),
        "svmc_memset": ([vp, i32, sz, vp], i32),
        "svmc_memcpy_h2d": ([vp, vp, sz, vp], i32),
        "svmc_memcpy_d2h": ([vp, vp, sz, vp], i32),
        "svmc_memcpy_d2d": ([vp, vp, sz, vp], i32),
        "svmc_memcpy2d_h2d": ([vp, sz, vp, sz, sz, sz, vp], i32),
        "svmc_stream_create": ([pvp], i32),
        "svmc_stream_destroy": ([vp], i32),
        "svmc_stream_synchronize": ([vp], i32),
        "svmc_event_create": ([pvp], i32),
        "svmc_event_destroy": ([vp], i32),
        "svmc_event_record": ([vp, vp], i32),
        "svmc_event_synchronize": ([vp], i32),
        "svmc_event_elapsed_ms": ([vp, vp, pf], i32),
        "svmc_row_power_sums": ([vp, sz, sz, sz, f64, i32, vp, vp, sz, vp], i32),
        "svmc_expanding_mean_squares": ([vp, sz, sz, sz, vp, sz, vp], i32),
        "svmc_clock_probe_arm": ([i32], i32),
        "svmc_clock_probe_read": ([C.POINTER(u64), vp], i32),
        "svmc_fill_state": ([vp, vp, vp, sz, f64, f64, f64, vp], i32),
        "svmc_fill_normals": ([vp, vp, sz, sz, i32, u64, u32, u64, u32, vp], i32),
        "svmc_fill_uniforms": ([vp, sz, sz, i32, u64, u32, u64, u32, vp], i32),
        "svmc_logsv_terminal_rng": ([vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, f64, f64, i32, u64, u32, u64, u32,
                                     vp], i32),
        "svmc_logsv_slice_rng": ([vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, f64, f64, i32, u64, u32, u64, u32,
                                  f64, vp, vp, vp, vp, sz, vp], i32),
        "svmc_logsv_terminal_w": ([vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, f64, f64, i32, vp, vp, sz, vp], i32),
        "svmc_logsv_vol_paths": ([vp, sz, sz, i32, f64, f64, f64, f64, f64, f64, f64, i32, vp, sz, u64, u32, u64, vp], i32),
        "svmc_black_implied_vols": ([pf64, pf64, C.POINTER(C.c_int8), sz, f64, f64, f64, f64, f64, pf64], i32),
        "svmc_logsv_chain_rng": ([vp, vp, vp, sz, i32, C.POINTER(i32), pf64, pf64, pf64, f64, f64, f64, f64, f64, i32, u64, u32,
                                  u64, u32, vp, vp, vp, vp, sz, vp], i32),
        "svmc_heston_chain_rng": ([vp, vp, vp, sz, i32, C.POINTER(i32), pf64, pf64, f64, f64, f64, f64, i32, u64, u32, u64, u32,
                                   vp, vp, vp, vp, sz, vp], i32),
        "svmc_logsv_slice_w": ([vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, f64, f64, i32, vp, vp, sz, f64, vp, vp, vp, vp,
                                sz, vp], i32),
        "svmc_rough_logsv_slice": ([vp, vp, vp, sz, i32, f64, i32, pf64, pf64, pf64, f64, f64, f64, f64, f64, vp, vp, sz,
                                    u64, u32, u64, u32, i32, f64, vp, vp, vp, vp, sz, vp], i32),
        "svmc_rough_logsv_chain": ([vp, vp, vp, sz, i32, C.POINTER(i32), pf64, pf64, i32, pf64, pf64, pf64, f64, f64, f64, f64, f64,
                                    vp, vp, sz, u64, u32, u64, vp, vp, vp, vp, sz, vp], i32),
        "svmc_rough_logsv_terminal": ([vp, vp, vp, sz, i32, f64, i32, pf64, pf64, pf64, f64, f64, f64, f64, f64, vp, vp, sz,
                                       u64, u32, u64, u32, i32, vp], i32),
        "svmc_heston_terminal_rng": ([vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, i32, u64, u32, u64, u32, vp], i32),
        "svmc_logsv_slice_rng_from": ([f64, f64, f64, vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, f64, f64, i32, u64, u32, u64,
                                       u32, f64, vp, vp, vp, vp, sz, vp], i32),
        "svmc_logsv_chain_rng_from": ([f64, f64, f64, vp, vp, vp, sz, i32, C.POINTER(i32), pf64, pf64, pf64, f64, f64, f64, f64,
                                       f64, i32, u64, u32, u64, u32, vp, vp, vp, vp, sz, vp], i32),
        "svmc_heston_slice_rng_from": ([f64, f64, f64, vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, i32, u64, u32, u64, u32,
                                        f64, vp, vp, vp, vp, sz, vp], i32),
        "svmc_heston_chain_rng_from": ([f64, f64, f64, vp, vp, vp, sz, i32, C.POINTER(i32), pf64, pf64, f64, f64, f64, f64, i32,
                                        u64, u32, u64, u32, vp, vp, vp, vp, sz, vp], i32),
        "svmc_heston_slice_rng": ([vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, i32, u64, u32, u64, u32,
                                   f64, vp, vp, vp, vp, sz, vp], i32),
        "svmc_heston_terminal_w": ([vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, vp, vp, sz, vp], i32),
        "svmc_heston_qe_terminal_w": ([vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, vp, vp, vp, sz, vp], i32),
        "svmc_payoff_workspace_bytes": ([psz], i32),
        "svmc_slice_workspace_bytes": ([sz, psz], i32),
        "svmc_spot_sums": ([vp, sz, f64, vp, vp, sz, vp], i32),
        "svmc_payoff_sums_chain": ([C.POINTER(vp), C.POINTER(vp), sz, pf64, pf64, vp, i32, pf64, pi8, pf64, psz, i32, vp, vp,
                                    sz, vp], i32),
        "svmc_payoff_sums": ([vp, vp, sz, f64, f64, vp, pf64, pi8, pf64, sz, i32, vp, vp, sz, vp], i32),
        "svmc_logsv_mgf_grid": ([vp, vp, sz, f64, f64, f64, f64, f64, f64, f64, i32, i32, f64, vp, vp, f64, f64, vp], i32),
        "svmc_logsv_mgf_grid_batch": ([vp, vp, sz, i32, f64, pf64, i32, i32, vp, vp, f64, f64, vp], i32),
        "svmc_mgf_vanilla_slice_batch": ([vp, vp, sz, i32, f64, pf64, sz, vp, vp], i32),
        "svmc_heston_mgf_grid": ([vp, vp, sz, f64, f64, f64, f64, f64, f64, vp, vp, i32, vp, vp], i32),
        "svmc_mgf_qvar_slice": ([vp, vp, sz, f64, pf64, sz, vp, vp], i32),
        "svmc_mgf_vanilla_slice": ([vp, vp, sz, f64, pf64, sz, vp, vp], i32),
        "svmc_session_create": ([pvp, sz, i32, sz], i32),
        "svmc_session_create_on": ([pvp, sz, i32, sz, vp, vp, vp, u64, vp], i32),
        "svmc_session_destroy": ([vp], i32),
        "svmc_session_time_stepping": ([vp, i32], i32),
        "svmc_session_last_stepping_ms": ([vp, C.POINTER(C.c_float)], i32),
        "svmc_session_state": ([vp, vp, vp, vp], i32),
        "svmc_logsv_chain_price": ([vp, pf64, pf64, pf64, pf64, i32, pf64, pi8, psz, f64, f64, f64, f64, f64, f64, i32,
                                    i32, i32, u64, u32, pf64, pf64], i32),
        "svmc_logsv_chain_price_fixed": ([vp, pf64, pf64, pf64, pf64, i32, pf64, pi8, psz, f64, f64, f64, f64, f64, f64, i32,
                                          i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), pf64, sz, pf64, pf64], i32),
        "svmc_logsv_chain_price_fixed_iv": ([vp, pf64, pf64, pf64, pf64, i32, pf64, pi8, psz, f64, f64, f64, f64, f64, f64, i32,
                                             i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), pf64, sz, pf64, pf64, pf64],
                                            i32),
        "svmc_logsv_chain_price_fixed_sets": ([vp, pf64, pf64, pf64, i32, pf64, pi8, psz, i32, pf64, i32, i32, C.POINTER(vp),
                                               C.POINTER(vp), C.POINTER(i32), pf64, sz, pf64, pf64, pf64], i32),
        "svmc_logsv_chain_price_frozen_sets": ([vp, pf64, pf64, pf64, i32, pf64, pi8, psz, i32, pf64, i32, i32, C.POINTER(i32),
                                                pf64, u64, u32, pf64, pf64, pf64], i32),
        "svmc_session_use_graphs": ([vp, i32], i32),
        "svmc_session_graph_launches": ([vp, psz], i32),
        "svmc_heston_chain_price": ([vp, pf64, pf64, pf64, i32, pf64, pi8, psz, f64, f64, f64, f64, f64, i32, i32, i32,
                                     u64, u32, pf64, pf64], i32),
        "svmc_payoff_finalize": ([pf64, pf64, sz, f64, f64, pf64, pf64], i32),
        "svmc_payoff_finalize_chain": ([pf64, pf64, pf64, sz, f64, pf64, pf64], i32),
        "svmc_rccl_available": ([], i32),
        "svmc_rccl_origin": ([], C.c_char_p),
        "svmc_rccl_unique_id": ([vp, sz], i32),
        "svmc_rccl_comm_create": ([pvp, vp, sz, i32, i32], i32),
        "svmc_rccl_comm_destroy": ([vp], i32),
        "svmc_rccl_comm_count": ([vp, pi32, pi32], i32),
        "svmc_rccl_all_reduce_sum": ([vp, vp, sz, vp], i32),
        "svmc_session_set_comm": ([vp, vp, i32, i32, u64, u64], i32),
        "svmc_session_set_reducer": ([vp, ALL_REDUCE_FN, vp, i32, i32, u64, u64], i32),
        "svmc_multi_create": ([pvp, i32, pi32, u64, i32, sz, i32], i32),
        "svmc_multi_destroy": ([vp], i32),
        "svmc_multi_info": ([vp, pi32, pi32, pi32, pi32], i32),
        "svmc_multi_shard_info": ([vp, i32, pi32, C.POINTER(u64), C.POINTER(u64), pf64], i32),
        "svmc_multi_logsv_chain_price": ([vp, pf64, pf64, pf64, pf64, i32, pf64, pi8, psz, f64, f64, f64, f64, f64, f64, i32,
                                          i32, i32, u64, u32, pf64, pf64], i32),
        "svmc_multi_heston_chain_price": ([vp, pf64, pf64, pf64, i32, pf64, pi8, psz, f64, f64, f64, f64, f64, i32, i32, i32,
                                           u64, u32, pf64, pf64], i32),
        "svmc_multi_state": ([vp, pf64, pf64, pf64], i32),
    }
    for name, (argtypes, restype) in sig.items():
        if os.environ.get("SVMC_ALLOW_OLD_ABI") == "1" and not hasattr(L, name):
            continue                   # an A/B build of an OLDER ABI (tools/ubench sets this): its newer entry points are absent
        fn = getattr(L, name)          # AttributeError here = the .so does not match include/svmc.h
        fn.argtypes = argtypes
        fn.restype = restype
    L._svmc_symbols = tuple(sig)


def load() -> C.CDLL:
    """load libsvmc.so once; raise loudly when it is not there (no fallback exists)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise SvmcError(
                    f"{LIB_PATH} is not built: run `python -m stochvolmodels_amd.build` "
                    "(needs hipcc; the Monte Carlo path has no CPU fallback)")
            # One HIP runtime per process: PyTorch-ROCm links its bundled runtime under the un-versioned
            # name, so if torch were imported AFTER this library a second copy would be mapped.  Importing
            # torch first makes the loader resolve our libamdhip64.so.7 dependency to the copy torch loaded.
            try:
                import torch  # noqa: F401
            except Exception:  # torch absent: stand-alone ROCm runtime via the library's RUNPATH
                pass
            L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
            _declare(L)
            _lib = L
    return _lib


def check(rc: int) -> None:
    """map an svmc status to the reference's exception types (utils/mc_payoffs.py:69-70,84)."""
    if rc == OK:
        return
    msg = load().svmc_last_error().decode("utf-8", "replace")
    if rc == ERR_UNKNOWN_PAYOFF:
        raise ValueError("unknown option payoff code")
    if rc == ERR_UNSUPPORTED_VARIABLE:
        raise NotImplementedError(msg)
    if rc == ERR_INVALID_ARGUMENT:
        raise ValueError(msg)
    if rc == ERR_RCCL:
        raise SvmcError(f"RCCL: {msg}")
    raise SvmcError(f"svmc status {rc}: {msg}")
