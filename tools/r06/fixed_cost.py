#!/usr/bin/env python
"""Round 6: the fixed cost of one svmc_logsv_chain_price call (C ABI through ctypes): the 4 x 13 chain at 364, 8 and 2 steps and
64 / 4096 / 65536 paths, beside one trivial kernel + hipStreamSynchronize -- profiles/r06_fixed_cost.txt.

    python tools/r06/fixed_cost.py
"""
import ctypes as C, time, numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import stochvolmodels_amd as sv
from stochvolmodels_amd import _lib
from stochvolmodels_amd.engine import option_type_codes
L=_lib.load(); p=sv.LOGSV_BTC_PARAMS
k=np.linspace(0.7,1.3,13); ty=np.where(k>=1.0,"C","P")
for label, ttms, spy in (("4 expiries x 13, 364 steps", np.array([1/12,0.25,0.5,1.0]), 360), ("4 expiries x 13, 8 steps", np.array([1/12,0.25,0.5,1.0]), 4), ("1 expiry x 13, 2 steps", np.array([1.0]), 1)):
    m=len(ttms)
    for n in (64, 4096, 65536):
        sess=C.c_void_p(); kk=np.ascontiguousarray(np.concatenate([k]*m)); codes=np.ascontiguousarray(np.concatenate([option_type_codes(ty)]*m).astype(np.int8))
        offs=(C.c_size_t*(m+1))(*[13*i for i in range(m+1)])
        _lib.check(L.svmc_session_create(C.byref(sess), n, m, kk.size))
        prices,errs,etas,fw,df=np.empty(kk.size),np.empty(kk.size),np.ones(m),np.ones(m),np.ones(m)
        dp=lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        def fn(seed):
            _lib.check(L.svmc_logsv_chain_price(sess, dp(ttms), dp(fw), dp(df), dp(etas), m, dp(kk), codes.ctypes.data_as(C.POINTER(C.c_int8)), offs, p.sigma0,p.theta,p.kappa1,p.kappa2,p.beta,p.volvol,1,spy,1,seed,0,dp(prices),dp(errs)))
        for i in range(30): fn(i)
        ts=[]
        for i in range(400):
            t0=time.perf_counter(); fn(100+i); ts.append(time.perf_counter()-t0)
        print(label, "n", n, "wall us median %.1f min %.1f" % (1e6*np.median(ts), 1e6*np.min(ts)))
        L.svmc_session_destroy(sess)
# floor: one trivial kernel + sync
x=C.c_void_p(); L.svmc_malloc.argtypes=[C.POINTER(C.c_void_p), C.c_size_t]; L.svmc_malloc(C.byref(x), 8*64*3)
L.svmc_fill_state.argtypes=[C.c_void_p]*3+[C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_void_p]
L.svmc_stream_synchronize.argtypes=[C.c_void_p]
def floor():
    L.svmc_fill_state(x, C.c_void_p(x.value+512), C.c_void_p(x.value+1024), 64, 0.0, 1.0, 0.0, None); L.svmc_stream_synchronize(None)
for i in range(30): floor()
ts=[]
for i in range(400):
    t0=time.perf_counter(); floor(); ts.append(time.perf_counter()-t0)
print("one trivial kernel + stream synchronize: us median %.1f min %.1f" % (1e6*np.median(ts), 1e6*np.min(ts)))
