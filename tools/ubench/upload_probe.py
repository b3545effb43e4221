import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import ctypes as C
from stochvolmodels_amd import _lib
from stochvolmodels_amd.engine import DeviceBuffer, get_engine
L = _lib.load()
eng = get_engine(1024)   # runtime init
n, nb = 100000, 182
rng = np.random.RandomState(1)
t=time.perf_counter(); W = rng.normal(0,1,size=(nb,n)); print("draw", time.perf_counter()-t)
t=time.perf_counter(); buf = DeviceBuffer(nb*n); print("malloc", time.perf_counter()-t)
t=time.perf_counter(); _lib.check(L.svmc_memcpy2d_h2d(buf.ptr, 8*n, W.ctypes.data, 8*n, 8*n, nb, None)); _lib.check(L.svmc_stream_synchronize(None)); print("h2d first", time.perf_counter()-t, W.nbytes/1e6,"MB")
t=time.perf_counter(); _lib.check(L.svmc_memcpy2d_h2d(buf.ptr, 8*n, W.ctypes.data, 8*n, 8*n, nb, None)); _lib.check(L.svmc_stream_synchronize(None)); print("h2d second", time.perf_counter()-t)
W2 = rng.normal(0,1,size=(nb,n))
t=time.perf_counter(); _lib.check(L.svmc_memcpy_h2d(buf.ptr, W2.ctypes.data, W2.nbytes, None)); _lib.check(L.svmc_stream_synchronize(None)); print("h2d fresh array linear", time.perf_counter()-t)
W3 = np.ones((nb,n))
t=time.perf_counter(); _lib.check(L.svmc_memcpy_h2d(buf.ptr, W3.ctypes.data, W3.nbytes, None)); _lib.check(L.svmc_stream_synchronize(None)); print("h2d np.ones array", time.perf_counter()-t)
t=time.perf_counter(); buf2 = DeviceBuffer(nb*n); print("malloc2", time.perf_counter()-t)
t=time.perf_counter(); _lib.check(L.svmc_memcpy_h2d(buf2.ptr, W3.ctypes.data, W3.nbytes, None)); _lib.check(L.svmc_stream_synchronize(None)); print("h2d into FRESH device buffer", time.perf_counter()-t)
t=time.perf_counter(); _lib.check(L.svmc_memcpy_h2d(buf2.ptr, W3.ctypes.data, W3.nbytes, None)); _lib.check(L.svmc_stream_synchronize(None)); print("h2d into it again", time.perf_counter()-t)
W4 = rng.normal(0,1,size=(nb,n))
t=time.perf_counter(); _lib.check(L.svmc_memcpy_h2d(buf2.ptr, W4.ctypes.data, W4.nbytes, None)); _lib.check(L.svmc_stream_synchronize(None)); print("fresh host array into used device buffer", time.perf_counter()-t)
buf3 = DeviceBuffer(nb*n); W5 = rng.normal(0,1,size=(nb,n))
t=time.perf_counter(); _lib.check(L.svmc_memcpy_h2d(buf3.ptr, W5.ctypes.data, W5.nbytes, None)); _lib.check(L.svmc_stream_synchronize(None)); print("fresh host array into fresh device buffer", time.perf_counter()-t)
t=time.perf_counter(); buf2.free(); print("free", time.perf_counter()-t)
