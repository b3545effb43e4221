"""A/B of logsv_vol_paths_kernel between builds of libsvmc on the SAME box, raw ctypes, one library per process:

    python tools/ubench/ab_vol_paths.py <lib.so> [tag]

2^20 paths x {1024, 360} steps, device RNG and supplied brownians; HIP events around 6 launches after 2 warm-ups (mean and
min); a digest of the output (sum and sum of squares of every 4097th element + the last row's sum) so that variants that
must not change the bits can be compared.  Only symbols every build since round 2 exports are used."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch  # BEFORE the library: one HIP runtime per process (torch links its bundled copy under the un-versioned name; loaded
              # after libsvmc's libamdhip64.so.7 it would map a second one and see no device) -- stochvolmodels_amd/_lib.py does the same

L = C.CDLL(os.path.abspath(sys.argv[1]))
tag = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(sys.argv[1])
vp, f64, sz, i32, u64, u32 = C.c_void_p, C.c_double, C.c_size_t, C.c_int, C.c_uint64, C.c_uint32
L.svmc_malloc.argtypes = [C.POINTER(vp), sz]
L.svmc_free.argtypes = [vp]
L.svmc_fill_normals.argtypes = [vp, vp, sz, sz, i32, u64, u32, u64, u32, vp]
L.svmc_logsv_vol_paths.argtypes = [vp, sz, sz, i32, f64, f64, f64, f64, f64, f64, f64, i32, vp, sz, u64, u32, u64, vp]
L.svmc_event_create.argtypes = [C.POINTER(vp)]
L.svmc_event_record.argtypes = [vp, vp]
L.svmc_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
L.svmc_stream_synchronize.argtypes = [vp]
L.svmc_memcpy_d2h.argtypes = [vp, vp, sz, vp]


def malloc(n):
    p = vp()
    assert L.svmc_malloc(C.byref(p), 8 * n) == 0
    return p


n = 1 << 20
res = {"lib": tag}
for nb in (1024, 360):
    out = malloc((nb + 1) * n)
    # supplied brownians are the reference's SCALED increments sqrt(dt) N(0,1) (pricers/logsv_pricer.py:925).  Round 4 passed
    # svmc_fill_normals' UNSCALED N(0,1) here: sigma overflowed within tens of steps, the whole output was NaN (the committed
    # supplied_360_digest of profiles/r04_vol_paths.json) and NaN operands toggle less, draw less power and clock higher than
    # real data -- the "supplied" figures of round 4 were measured on that.  Now: scaled, and the digest must be finite.
    bt = torch.randn(nb, n, dtype=torch.float64, device="cuda") * (1.0 / 360) ** 0.5
    torch.cuda.synchronize()
    for mode, b in (("rng", None), ("supplied", vp(bt.data_ptr()))):
        def launch():
            return L.svmc_logsv_vol_paths(out, n, n, nb, 1.0 / 360, 0.8, 1.0, 3.0, 3.0, 0.15, 1.8, 1, b, n, 5, 0, 0, None)
        for _ in range(2):
            assert launch() == 0
        L.svmc_stream_synchronize(None)
        ts = []
        for _ in range(6):
            e0, e1 = vp(), vp()
            L.svmc_event_create(C.byref(e0)); L.svmc_event_create(C.byref(e1))
            L.svmc_event_record(e0, None)
            assert launch() == 0
            L.svmc_event_record(e1, None)
            L.svmc_stream_synchronize(None)
            ms = C.c_float(); L.svmc_event_elapsed_ms(e0, e1, C.byref(ms)); ts.append(ms.value)
        t = float(np.mean(ts))
        wr = 8.0 * (nb + 1) * n
        rd = 8.0 * nb * n if b else 0.0
        key = f"{mode}_{nb}"
        if hasattr(L, "svmc_clock_probe_read"):         # round-4+ builds: the shader clock inside the last launch
            st = (C.c_uint64 * 8)()
            L.svmc_clock_probe_read.argtypes = [C.POINTER(C.c_uint64), vp]
            if hasattr(L, "svmc_clock_probe_arm"):      # round-5 builds: the probe is off unless the thread armed it
                L.svmc_clock_probe_arm.argtypes = [i32]
                L.svmc_clock_probe_arm(1)
                launch()
            L.svmc_clock_probe_read(st, None)
            mhz = [100.0 * (st[i + 2] - st[i]) / (st[i + 3] - st[i + 1]) for i in (0, 4) if st[i + 3] > st[i + 1]]
            res[key + "_clock_mhz"] = [round(v, 1) for v in mhz]
        res[key + "_ms"] = round(t, 4)
        res[key + "_min_ms"] = round(float(min(ts)), 4)
        res[key + "_write_TBps"] = round(wr / t / 1e9, 3)
        res[key + "_total_TBps"] = round((wr + rd) / t / 1e9, 3)
        if nb == 360:                                   # digest of the output (device RNG / supplied)
            host = np.empty((nb + 1) * n)
            L.svmc_memcpy_d2h(host.ctypes.data, out, 8 * host.size, None)
            L.svmc_stream_synchronize(None)
            samp = host[::4097]
            res[key + "_digest"] = [float(samp.sum()).hex(), float((samp * samp).sum()).hex(), float(host[-n:].sum()).hex()]
            assert np.all(np.isfinite(host)), f"{key}: the output holds non-finite values -- the measurement ran on NaN data"
            res[key + "_finite"] = True
    L.svmc_free(out)
    del bt
print(json.dumps(res))
