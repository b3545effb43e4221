#!/usr/bin/env python
"""
End-to-end wall time of the bulk-output calls: simulate_vol_paths returning the reference's NumPy array,
at 2^20 x 1024 (8.6 GB) and at the reference's own caller's size 4 x 10^5 x 361 (1.2 GB; papers/.../moments_vol_qvar.py runs
10^5 x 541), against (a) the box's pinned device-to-host rate measured with one plain copy into page-locked memory, (b) the
round-4 route (one synchronous copy into pageable memory), (c) the resident forms: return_device=True and vol_path_moments().
Also simulate_terminal_values (24 bytes per path).  One JSON object per line.

    python tools/r05/bulk_outputs.py
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402
from stochvolmodels_amd import _lib, engine  # noqa: E402


def pinned_rate_gbs(n_bytes=1 << 30, reps=3):
    L = _lib.load()
    d, h = C.c_void_p(), C.c_void_p()
    _lib.check(L.svmc_malloc(C.byref(d), n_bytes))
    _lib.check(L.svmc_host_alloc(C.byref(h), n_bytes))
    _lib.check(L.svmc_memset(d, 1, n_bytes, None))
    _lib.check(L.svmc_stream_synchronize(None))
    best = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        _lib.check(L.svmc_memcpy_d2h(h, d, n_bytes, None))
        _lib.check(L.svmc_stream_synchronize(None))
        best = max(best, n_bytes / (time.perf_counter() - t0) / 1e9)
    L.svmc_free(d)
    L.svmc_host_free(h)
    return best


def pageable_copy_s(dev_ptr, n_doubles):
    """the round-4 route: one synchronous hipMemcpy into a fresh pageable NumPy array"""
    L = _lib.load()
    t0 = time.perf_counter()
    out = np.empty(n_doubles)
    _lib.check(L.svmc_memcpy_d2h(out.ctypes.data, dev_ptr, 8 * n_doubles, None))
    _lib.check(L.svmc_stream_synchronize(None))
    return time.perf_counter() - t0


def main():
    p = sv.LOGSV_BTC_PARAMS
    rate = pinned_rate_gbs()
    print(json.dumps({"pinned_d2h_GBps_one_plain_copy_of_1GiB": rate, "host_cores": os.cpu_count(),
                      "pipeline": {"chunk_MiB": engine.PIPELINE_CHUNK_BYTES >> 20, "slots": engine.PIPELINE_SLOTS,
                                   "threads": engine.PIPELINE_THREADS}}), flush=True)
    pricer = sv.LogSVPricer()
    for n, ttm, spy, tag in ((1 << 20, 1.0, 1023, "2^20 x 1024"), (400_000, 1.0, 360, "4e5 x 361")):
        kw = dict(params=p, ttm=ttm, nb_path=n, nb_steps=spy, seed=5)
        pricer.simulate_vol_paths(**dict(kw, nb_path=4096))                     # library warm
        rows = None
        res = {"config": tag}
        for rep in range(2):                                                    # second repeat: destination pages already mapped once
            t0 = time.perf_counter()
            sig, grid = pricer.simulate_vol_paths(**kw)
            res[f"numpy_return_s_rep{rep}"] = time.perf_counter() - t0
            rows = sig.shape[0]
            digest = float(sig[-1, :1000].sum())
            del sig
        nbytes = 8.0 * rows * n
        res["bytes"] = nbytes
        res["GBps_of_the_call"] = nbytes / min(res["numpy_return_s_rep0"], res["numpy_return_s_rep1"]) / 1e9
        res["fraction_of_pinned_rate"] = res["GBps_of_the_call"] / rate
        out = np.empty((rows, n))
        out[:] = 0.0                                                            # a caller-owned, already mapped destination
        t0 = time.perf_counter()
        pricer.simulate_vol_paths(out=out, **kw)
        res["into_caller_array_s"] = time.perf_counter() - t0
        res["into_caller_array_fraction_of_pinned_rate"] = nbytes / res["into_caller_array_s"] / 1e9 / rate
        assert float(out[-1, :1000].sum()) == digest
        del out
        t0 = time.perf_counter()
        dev, _ = pricer.simulate_vol_paths(return_device=True, **kw)
        dev.synchronize()
        res["return_device_s"] = time.perf_counter() - t0
        res["round4_pageable_copy_s"] = pageable_copy_s(dev.ptr, rows * n)
        dev.free()
        for rep in range(3):               # rep 0 grows the engine's cached bulk buffers (an 8.6 GB hipMalloc maps VRAM: 0.2-0.7 s)
            t0 = time.perf_counter()
            mom = pricer.vol_path_moments(p, ttm=ttm, nb_path=n, nb_steps=spy, seed=5, with_qvar=True)
            res[f"vol_path_moments_with_qvar_s_rep{rep}"] = time.perf_counter() - t0
        res["moment_rows"] = int(mom["mean"].shape[0])
        print(json.dumps(res), flush=True)
    for n in (1 << 20, 1 << 24):
        pricer.simulate_terminal_values(p, nb_path=n, seed=3)
        t0 = time.perf_counter()
        x, s, q = pricer.simulate_terminal_values(p, nb_path=n, seed=3)
        dt = time.perf_counter() - t0
        print(json.dumps({"config": f"simulate_terminal_values {n} paths x 361 steps", "s": dt, "state_bytes": 24.0 * n,
                          "state_GBps_incl_kernel": 24.0 * n / dt / 1e9}), flush=True)


if __name__ == "__main__":
    main()
