"""Per-call time of C2 through (a) LogSVPricer.model_mc_price_chain (Python chain driver, one ctypes call per launch)
and (b) ONE call of the C++ driver svmc_logsv_chain_price: how much of the whole-call time is host overhead.

Round 4: the first version timed 62 Python-route calls and then 62 C-route calls as two averaged
blocks with the garbage collector live -- one generation-2 collection (53-67 ms, profiles/r03_gc_stall.json) landing in
either block moved its mean by +1 ms, and the committed numbers were bimodal (1.62 / 2.69 / 3.72 ms).  Now: gc.collect() +
gc.freeze() after the warm-up, the two routes INTERLEAVED call by call (clock drift and box noise hit both alike), every
call timed on its own, median / mean / max / the four slowest calls per route, 200 calls per route, and a gc.callbacks hook
that reports any collection that still happens inside the timed region.

    python tools/ubench/fused_driver_overhead.py [calls] > profiles/r04_fused_driver_overhead.json
"""
import ctypes as C
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np

import stochvolmodels_amd as sv
from stochvolmodels_amd import _lib

CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
# --as-round-3: reproduce the first version's protocol (collector live, one block of calls per route) with per-call timing and
# the gc hook, to show WHERE its extra milliseconds were
LEGACY = "--as-round-3" in sys.argv
P = sv.LOGSV_BTC_PARAMS
n, spy = 1 << 20, 1023
kk = np.linspace(0.5, 1.5, 21)
ty = np.where(kk >= 1.0, "C", "P")
chain = sv.OptionChain.slice_to_chain(ttm=1.0, forward=1.0, strikes=kk, optiontypes=ty)
pr = sv.LogSVPricer()


def py_call(i):
    return pr.model_mc_price_chain(chain, P, nb_path=n, nb_steps=spy, seed=100 + i)


lib = _lib.load()
sess = C.c_void_p()
_lib.check(lib.svmc_session_create(C.byref(sess), n, 1, 21))
dp = C.POINTER(C.c_double)
arr = lambda v: np.ascontiguousarray(v, dtype=np.float64)                                            # noqa: E731
ttms, fw, df, eta, kk64 = arr([1.0]), arr([1.0]), arr([1.0]), arr([1.0]), arr(kk)
codes = np.ascontiguousarray(ty != "C", dtype=np.int8)
offs = np.array([0, 21], dtype=np.uintp)
prices, errs = np.empty(21), np.empty(21)
c_args = (ttms.ctypes.data_as(dp), fw.ctypes.data_as(dp), df.ctypes.data_as(dp), eta.ctypes.data_as(dp), 1,
          kk64.ctypes.data_as(dp), codes.ctypes.data_as(C.POINTER(C.c_int8)), offs.ctypes.data_as(C.POINTER(C.c_size_t)))
c_out = (prices.ctypes.data_as(dp), errs.ctypes.data_as(dp))


def c_call(i):
    _lib.check(lib.svmc_logsv_chain_price(sess, *c_args, P.sigma0, P.theta, P.kappa1, P.kappa2, P.beta, P.volvol, 1, spy, 1,
                                          100 + i, 0, *c_out))
    return prices, errs


for i in range(12):                                  # clock ramp, first-call work of both routes
    py_call(-i)
    c_call(-i)
if not LEGACY:
    gc.collect()
    gc.freeze()
collections = []


def on_gc(phase, info, _t=[0.0]):
    if phase == "start":
        _t[0] = time.perf_counter()
    elif info["generation"] == 2 or LEGACY is False:
        collections.append({"generation": info["generation"], "ms": round(1e3 * (time.perf_counter() - _t[0]), 3)})


gc.callbacks.append(on_gc)
t_py, t_c = np.empty(CALLS), np.empty(CALLS)
equal = True
if LEGACY:
    for i in range(CALLS):
        t0 = time.perf_counter()
        a = py_call(i)
        t_py[i] = time.perf_counter() - t0
    n_gc_python_block = len(collections)
    for i in range(CALLS):
        t0 = time.perf_counter()
        b = c_call(i)
        t_c[i] = time.perf_counter() - t0
    equal = bool(np.array_equal(a[0][0], b[0]))
else:
    for i in range(CALLS):
        t0 = time.perf_counter()
        a = py_call(i)
        t1 = time.perf_counter()
        b = c_call(i)
        t2 = time.perf_counter()
        t_py[i], t_c[i] = t1 - t0, t2 - t1
        equal = equal and bool(np.array_equal(a[0][0], b[0])) and bool(np.array_equal(a[1][0], b[1]))
gc.callbacks.remove(on_gc)


def stats(t):
    t = 1e3 * t
    worst = np.argsort(t)[::-1][:4]
    return {"median_ms": float(np.median(t)), "mean_ms": float(t.mean()), "min_ms": float(t.min()), "max_ms": float(t.max()),
            "p99_ms": float(np.percentile(t, 99)), "slowest": [[int(j), round(float(t[j]), 3)] for j in worst]}


print(json.dumps({"calls_per_route": CALLS, "interleaved": not LEGACY, "gc_frozen": not LEGACY,
                  "protocol": "round 3 (collector live, a block of calls per route)" if LEGACY else "round 4",
                  "python_chain_driver": stats(t_py), "fused_c_driver": stats(t_c),
                  "fused_minus_python_median_ms": float(np.median(1e3 * t_c) - np.median(1e3 * t_py)),
                  "gc_collections_in_timed_region": collections,
                  "gen2_collections_in_timed_region": sum(1 for c in collections if c["generation"] == 2),
                  "prices_bit_equal_every_call": equal}))
