"""
Several GPUs from ONE Python process: the host mirror of the C ABI's multi-sessions (include/svmc.h svmc_multi_*).

The reference's users run one interpreter (logsv_mc_chain_pricer, pricers/logsv_pricer.py:806-867, is a single NumPy
loop); `logsv_mc_chain_pricer(..., devices=8)` / `heston_mc_chain_pricer(..., devices=[0, 1, 2, 3])` shard the job's paths
by global path id over that many devices inside libsvmc -- one host thread and one session per device, the two sum
all-reduces of compute_mc_vars_payoff (utils/mc_payoffs.py:61-63, :85-86) over RCCL (ncclCommInitAll) or through pinned
host memory -- with no launcher, no rendezvous and no second interpreter.  The process-per-GPU route (dist.py,
torch.distributed / RcclComm) stays what a multi-node or torchrun deployment uses; both give the same numbers.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from .engine import option_type_codes

REDUCE_AUTO, REDUCE_HOST, REDUCE_RCCL = 0, 1, 2
_REDUCE = {"auto": REDUCE_AUTO, "host": REDUCE_HOST, "rccl": REDUCE_RCCL, None: REDUCE_AUTO}


def resolve_devices(devices: Union[int, Sequence[int]]) -> Tuple[int, ...]:
    """devices=8 -> (0, .., 7); a sequence is taken as given (a device may appear twice: shards then share it)"""
    if isinstance(devices, (int, np.integer)):
        if devices < 1:
            raise ValueError("devices must be >= 1")
        return tuple(range(int(devices)))
    out = tuple(int(d) for d in devices)
    if not out:
        raise ValueError("devices must name at least one device")
    return out


class MultiDeviceSession:
    """n_path_total paths of chain pricing sharded over `devices` in this process (svmc_multi_create)."""

    def __init__(self, devices: Union[int, Sequence[int]], n_path_total: int, max_expiries: int, max_strikes_total: int,
                 reduce: Optional[str] = None):
        self.lib = _lib.load()
        self.devices = resolve_devices(devices)
        self.n_path_total = int(n_path_total)
        self.max_expiries, self.max_strikes = int(max_expiries), int(max(max_strikes_total, 1))
        if reduce not in _REDUCE:
            raise ValueError("reduce must be 'auto', 'host' or 'rccl'")
        devs = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        _lib.check(self.lib.svmc_multi_create(C.byref(h), len(self.devices), devs, self.n_path_total, self.max_expiries,
                                              self.max_strikes, _REDUCE[reduce]))
        self.handle = h

    # ---- diagnostics ---------------------------------------------------------------------------------------------
    def info(self) -> dict:
        n, mode, seen, agree = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(self.lib.svmc_multi_info(self.handle, C.byref(n), C.byref(mode), C.byref(seen), C.byref(agree)))
        shards = []
        for r in range(n.value):
            dev, off, cnt, ms = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_double()
            _lib.check(self.lib.svmc_multi_shard_info(self.handle, r, C.byref(dev), C.byref(off), C.byref(cnt), C.byref(ms)))
            shards.append({"device": dev.value, "path_offset": off.value, "n_path": cnt.value, "last_call_ms": ms.value})
        return {"n_shards": n.value, "reduce": {REDUCE_HOST: "host", REDUCE_RCCL: "rccl"}[mode.value],
                "rccl_ranks_seen": seen.value, "shards_agree": bool(agree.value), "shards": shards}

    # ---- pricing -------------------------------------------------------------------------------------------------
    def _chain_args(self, ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms):
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)      # noqa: E731
        m = len(ttms)
        strikes = [f64(np.asarray(k, dtype=np.float64)).ravel() for k in strikes_ttms]
        codes = [option_type_codes(np.asarray(t).ravel()) for t in optiontypes_ttms]          # ValueError on unknown codes
        offs = np.concatenate([[0], np.cumsum([k.size for k in strikes])]).astype(np.uintp)
        total = int(offs[-1])
        if m > self.max_expiries or total > self.max_strikes:
            raise ValueError("the chain exceeds the multi-session's sizes")
        k_all = f64(np.concatenate(strikes)) if total else np.zeros(1)
        c_all = np.ascontiguousarray(np.concatenate(codes), dtype=np.int8) if total else np.zeros(1, dtype=np.int8)
        return m, f64(ttms), f64(forwards), f64(discfactors), k_all, c_all, offs, total

    @staticmethod
    def _split(flat, offs, strikes_ttms):
        return [flat[offs[i]:offs[i + 1]].copy().reshape(np.shape(strikes_ttms[i])) for i in range(len(strikes_ttms))]

    def price_logsv_chain(self, ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms, v0, theta, kappa1, kappa2, beta,
                          volvol, vol_backbone_etas, is_spot_measure: bool, nb_steps_per_year: int, variable_type: int,
                          seed: int, call_id: int) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        m, tt, fw, df, k_all, c_all, offs, total = self._chain_args(ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms)
        etas = np.ascontiguousarray(vol_backbone_etas, dtype=np.float64)
        prices, stderrs = np.empty(max(total, 1)), np.empty(max(total, 1))
        dp = C.POINTER(C.c_double)
        _lib.check(self.lib.svmc_multi_logsv_chain_price(
            self.handle, tt.ctypes.data_as(dp), fw.ctypes.data_as(dp), df.ctypes.data_as(dp), etas.ctypes.data_as(dp), m,
            k_all.ctypes.data_as(dp), c_all.ctypes.data_as(C.POINTER(C.c_int8)), offs.ctypes.data_as(C.POINTER(C.c_size_t)),
            float(v0), float(theta), float(kappa1), float(kappa2), float(beta), float(volvol), int(bool(is_spot_measure)),
            int(nb_steps_per_year), int(variable_type), int(seed), int(call_id), prices.ctypes.data_as(dp),
            stderrs.ctypes.data_as(dp)))
        return self._split(prices, offs, strikes_ttms), self._split(stderrs, offs, strikes_ttms)

    def price_heston_chain(self, ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms, v0, theta, kappa, rho, volvol,
                           scheme: int, nb_steps_per_year: int, variable_type: int, seed: int, call_id: int
                           ) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        m, tt, fw, df, k_all, c_all, offs, total = self._chain_args(ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms)
        prices, stderrs = np.empty(max(total, 1)), np.empty(max(total, 1))
        dp = C.POINTER(C.c_double)
        _lib.check(self.lib.svmc_multi_heston_chain_price(
            self.handle, tt.ctypes.data_as(dp), fw.ctypes.data_as(dp), df.ctypes.data_as(dp), m, k_all.ctypes.data_as(dp),
            c_all.ctypes.data_as(C.POINTER(C.c_int8)), offs.ctypes.data_as(C.POINTER(C.c_size_t)), float(v0), float(theta),
            float(kappa), float(rho), float(volvol), int(scheme), int(nb_steps_per_year), int(variable_type), int(seed),
            int(call_id), prices.ctypes.data_as(dp), stderrs.ctypes.data_as(dp)))
        return self._split(prices, offs, strikes_ttms), self._split(stderrs, offs, strikes_ttms)

    def get_state(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """terminal (x, sigma | variance, qvar) of the whole job, in global path order"""
        x, v, q = (np.empty(self.n_path_total) for _ in range(3))
        dp = C.POINTER(C.c_double)
        _lib.check(self.lib.svmc_multi_state(self.handle, x.ctypes.data_as(dp), v.ctypes.data_as(dp), q.ctypes.data_as(dp)))
        return x, v, q

    def close(self) -> None:
        if getattr(self, "handle", None) is not None:
            self.lib.svmc_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# One resident multi-session per (devices, paths, transport, thread): a chain priced again re-uses its buffers, threads and
# communicators (creating them costs milliseconds to seconds -- RCCL builds its rings).  Grown when a larger chain arrives.
# At most MAX_CACHED_MULTI per thread; a thread's sessions are released when it asks for one more, or by another thread's
# request once the owner has ended.
MAX_CACHED_MULTI = 2
_CACHE = {}
_CACHE_LOCK = threading.Lock()


def get_multi_session(devices, n_path_total: int, n_expiries: int, n_strikes_total: int, reduce: Optional[str] = None
                      ) -> MultiDeviceSession:
    devs = resolve_devices(devices)
    key = (devs, int(n_path_total), reduce, threading.get_ident())
    with _CACHE_LOCK:
        ms = _CACHE.pop(key, None)
        if ms is not None and (ms.handle is None or ms.max_expiries < n_expiries or ms.max_strikes < n_strikes_total):
            ms.close()
            ms = None
        if ms is None:
            # make room -- but only among sessions no one can be pricing on right now: this thread's own (it is here, not inside
            # a pricing call) and those of threads that have ended.  A session of another LIVE thread may be in the middle of
            # svmc_multi_*_chain_price, outside this lock; destroying it would be a use-after-free (round-5 advisor finding)
            me = threading.get_ident()
            alive = {t.ident for t in threading.enumerate()}
            mine = [k for k in _CACHE if k[3] == me or k[3] not in alive]
            while mine and sum(1 for k in _CACHE if k[3] == me or k[3] not in alive) >= MAX_CACHED_MULTI:
                _CACHE.pop(mine.pop(0)).close()
            ms = MultiDeviceSession(devs, n_path_total, max(n_expiries, 8), max(n_strikes_total, 256), reduce=reduce)
        _CACHE[key] = ms                                    # most recently used last
        return ms


def close_all() -> None:
    with _CACHE_LOCK:
        for ms in _CACHE.values():
            ms.close()
        _CACHE.clear()
