"""Per-call time of C2 through (a) LogSVPricer.model_mc_price_chain (Python chain driver, one ctypes call per launch)
and (b) ONE call of the C++ driver svmc_logsv_chain_price: how much of the whole-call time is host overhead."""
import ctypes as C, os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import stochvolmodels_amd as sv
from stochvolmodels_amd import _lib
P = sv.LOGSV_BTC_PARAMS
n, spy = 1 << 20, 1023
kk = np.linspace(0.5, 1.5, 21); ty = np.where(kk >= 1.0, "C", "P")
chain = sv.OptionChain.slice_to_chain(ttm=1.0, forward=1.0, strikes=kk, optiontypes=ty)
pr = sv.LogSVPricer()
def py_call(i): return pr.model_mc_price_chain(chain, P, nb_path=n, nb_steps=spy, seed=100 + i)
for i in range(12): py_call(-i)
t0 = time.perf_counter()
for i in range(50): a = py_call(i)
t_py = (time.perf_counter() - t0) / 50
lib = _lib.load()
sess = C.c_void_p()
_lib.check(lib.svmc_session_create(C.byref(sess), n, 1, 21))
dp = C.POINTER(C.c_double)
arr = lambda v: np.ascontiguousarray(v, dtype=np.float64)
ttms, fw, df, eta = arr([1.0]), arr([1.0]), arr([1.0]), arr([1.0])
codes = np.ascontiguousarray(ty != "C", dtype=np.int8)
offs = np.array([0, 21], dtype=np.uint64)
prices, errs = np.empty(21), np.empty(21)
def c_call(i):
    _lib.check(lib.svmc_logsv_chain_price(sess, ttms.ctypes.data_as(dp), fw.ctypes.data_as(dp), df.ctypes.data_as(dp),
               eta.ctypes.data_as(dp), 1, arr(kk).ctypes.data_as(dp), codes.ctypes.data_as(C.POINTER(C.c_int8)),
               offs.ctypes.data_as(C.POINTER(C.c_size_t)), P.sigma0, P.theta, P.kappa1, P.kappa2, P.beta, P.volvol, 1, spy, 1,
               100 + i, 0, prices.ctypes.data_as(dp), errs.ctypes.data_as(dp)))
for i in range(12): c_call(-i)
t0 = time.perf_counter()
for i in range(50): c_call(i)
t_c = (time.perf_counter() - t0) / 50
print(json.dumps({"python_chain_driver_ms": 1e3 * t_py, "fused_c_driver_ms": 1e3 * t_c, "prices_equal": bool(np.allclose(prices, a[0][0], rtol=1e-12))}))
