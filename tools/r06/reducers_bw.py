#!/usr/bin/env python
"""
The two streaming reducers of the vol-paths array (row_power_sums_kernel<K>, expanding_mean_sq_kernel) on 2^20 paths x 1025 rows
(8.6 GB) by HIP events around repeated C-ABI calls on the resident array -- kernel time, not the Python caller's wall time (which
also holds two allocations and a download).  GB/s = algorithmic bytes (8 B per element read; the expanding mean also writes 8), beside
the runtime's device-to-device copy of the same array (the roof of a read-and-write stream).

    python tools/r06/reducers_bw.py [reps]                one JSON line
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402
from stochvolmodels_amd import _lib  # noqa: E402
from stochvolmodels_amd.engine import DeviceBuffer  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    L = _lib.load()
    p = sv.LOGSV_BTC_PARAMS
    pr = sv.LogSVPricer()
    dev, _ = pr.simulate_vol_paths(p, ttm=1.0, nb_path=1 << 20, nb_steps=1023, seed=5, return_device=True)
    dev.synchronize()
    rows, cols = dev.shape
    out = {"rows": rows, "cols": cols, "reps": reps}

    def timed(fn):
        e0, e1 = C.c_void_p(), C.c_void_p()
        _lib.check(L.svmc_event_create(C.byref(e0)))
        _lib.check(L.svmc_event_create(C.byref(e1)))
        fn()
        _lib.check(L.svmc_stream_synchronize(None))
        _lib.check(L.svmc_event_record(e0, None))
        for _ in range(reps):
            fn()
        _lib.check(L.svmc_event_record(e1, None))
        _lib.check(L.svmc_stream_synchronize(None))
        ms = C.c_float()
        _lib.check(L.svmc_event_elapsed_ms(e0, e1, C.byref(ms)))
        L.svmc_event_destroy(e0)
        L.svmc_event_destroy(e1)
        return ms.value / reps

    for k in (1, 4):
        sums, ws = DeviceBuffer(rows * 2 * k), DeviceBuffer(rows * 4 * 2 * k)
        ms = timed(lambda: _lib.check(L.svmc_row_power_sums(dev.ptr, cols, rows, cols, float(p.theta), k, sums.ptr, ws.ptr, ws.nbytes, None)))
        out[f"row_power_sums_{k}_ms"] = round(ms, 4)
        out[f"row_power_sums_{k}_TBps"] = round(8.0 * rows * cols / ms / 1e9, 3)
        sums.free()
        ws.free()
    q = DeviceBuffer(rows * cols)
    ms = timed(lambda: _lib.check(L.svmc_expanding_mean_squares(dev.ptr, cols, rows, cols, q.ptr, cols, None)))
    out["expanding_mean_sq_ms"] = round(ms, 4)
    out["expanding_mean_sq_TBps"] = round(16.0 * rows * cols / ms / 1e9, 3)
    # the roof of a read-and-write stream on this box: the runtime's own device-to-device copy of the same array
    ms = timed(lambda: _lib.check(L.svmc_memcpy_d2d(q.ptr, dev.ptr, 8 * rows * cols, None)))
    out["memcpy_d2d_ms"] = round(ms, 4)
    out["memcpy_d2d_TBps_read_plus_write"] = round(16.0 * rows * cols / ms / 1e9, 3)
    q.free()
    dev.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
