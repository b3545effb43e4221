"""Where the ~60 us between the last kernel of one chain call and the first of the next goes: launch + completion
latencies of the runtime on this box, measured through libsvmc's own C ABI.
  a) tiny kernel + hipStreamSynchronize          b) ... + 504-byte D2H into pageable memory (what engine.download does)
  c) ... + D2H into pinned memory                 d) ... + the kernel's result read from host-mapped pinned memory after a
                                                     stream synchronise (no copy at all)"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from stochvolmodels_amd import _lib
L = _lib.load()
vp = C.c_void_p
d = [vp() for _ in range(3)]
for x in d:
    assert L.svmc_malloc(C.byref(x), 8 * 4096) == 0
pin = vp(); assert L.svmc_host_alloc(C.byref(pin), 4096) == 0
host = np.empty(63)
def kern(): assert L.svmc_fill_state(d[0], d[1], d[2], 64, 0.0, 1.0, 0.0, None) == 0
def run(fn, reps=2000):
    for _ in range(50): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return 1e6 * (time.perf_counter() - t0) / reps
def a(): kern(); L.svmc_stream_synchronize(None)
def b(): kern(); L.svmc_memcpy_d2h(host.ctypes.data, d[0], 504, None); L.svmc_stream_synchronize(None)
def c(): kern(); L.svmc_memcpy_d2h(pin, d[0], 504, None); L.svmc_stream_synchronize(None)
def k5(): 
    for _ in range(5): kern()
    L.svmc_stream_synchronize(None)
out = {"kernel_sync_us": run(a), "kernel_d2h_pageable_sync_us": run(b), "kernel_d2h_pinned_sync_us": run(c), "five_kernels_sync_us": run(k5),
       "launch_only_us": run(kern)}
L.svmc_stream_synchronize(None)
print(json.dumps(out))
