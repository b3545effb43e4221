"""HBM write rate of logsv_vol_paths_kernel: 2^20 paths, 360 and 1024 steps, device RNG and supplied
brownians, HIP events around the C-ABI call."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.getcwd())
from stochvolmodels_amd import _lib  # noqa: E402
from stochvolmodels_amd.engine import DeviceBuffer, get_engine  # noqa: E402

L = _lib.load()


def ev():
    e = C.c_void_p()
    _lib.check(L.svmc_event_create(C.byref(e)))
    return e


n = 1 << 20
eng = get_engine(n)
for nb in (360, 1024):
    out = DeviceBuffer((nb + 1) * n)
    w0, _ = eng.fill_normals(nb, 3)
    for mode, b in (("device_rng", None), ("supplied", w0)):
        def f():
            _lib.check(L.svmc_logsv_vol_paths(out.ptr, n, n, nb, 1.0 / 360, 0.8, 1.0, 3.0, 3.0, 0.15, 1.8, 1, b, n, 5, 0, 0, None))
        f()
        e0, e1 = ev(), ev()
        L.svmc_event_record(e0, None)
        for _ in range(5):
            f()
        L.svmc_event_record(e1, None)
        eng.synchronize()
        ms = C.c_float()
        L.svmc_event_elapsed_ms(e0, e1, C.byref(ms))
        t = ms.value / 5
        wr = 8.0 * (nb + 1) * n
        rd = 8.0 * nb * n if b else 0.0
        print(json.dumps(dict(kernel="logsv_vol_paths_kernel", mode=mode, paths=n, steps=nb, ms=t,
                              write_GBps=wr / t / 1e6, total_GBps=(wr + rd) / t / 1e6)))
    out.free()
