"""A/B of the stepping kernel between two builds of libsvmc on the SAME box: python ab_kernel.py <lib.so> (raw ctypes,
only symbols both builds export)"""
import ctypes as C, json, sys, os
import torch  # noqa: F401  one HIP runtime per process
L = C.CDLL(os.path.abspath(sys.argv[1]))
vp, f64, sz, i32, u64, u32 = C.c_void_p, C.c_double, C.c_size_t, C.c_int, C.c_uint64, C.c_uint32
L.svmc_malloc.argtypes = [C.POINTER(vp), sz]
L.svmc_fill_state.argtypes = [vp, vp, vp, sz, f64, f64, f64, vp]
L.svmc_logsv_terminal_rng.argtypes = [vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, f64, f64, i32, u64, u32, u64, u32, vp]
L.svmc_event_create.argtypes = [C.POINTER(vp)]
L.svmc_event_record.argtypes = [vp, vp]
L.svmc_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
L.svmc_stream_synchronize.argtypes = [vp]
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
b = [vp() for _ in range(3)]
for x in b:
    assert L.svmc_malloc(C.byref(x), 8 * n) == 0
def run():
    L.svmc_fill_state(b[0], b[1], b[2], n, 0.0, 0.8376, 0.0, None)
    assert L.svmc_logsv_terminal_rng(b[0], b[1], b[2], n, 1024, 1 / 1024, 1.0413, 3.1844, 3.058, 0.1514, 1.8458, 1.0, 1, 7, 0, 0, 0, None) == 0
for _ in range(3):
    run()
L.svmc_stream_synchronize(None)
ts = []
for _ in range(20):
    L.svmc_fill_state(b[0], b[1], b[2], n, 0.0, 0.8376, 0.0, None)
    e0, e1 = vp(), vp()
    L.svmc_event_create(C.byref(e0)); L.svmc_event_create(C.byref(e1))
    L.svmc_event_record(e0, None)
    L.svmc_logsv_terminal_rng(b[0], b[1], b[2], n, 1024, 1 / 1024, 1.0413, 3.1844, 3.058, 0.1514, 1.8458, 1.0, 1, 7, 0, 0, 0, None)
    L.svmc_event_record(e1, None)
    ms = C.c_float(); L.svmc_event_elapsed_ms(e0, e1, C.byref(ms)); ts.append(ms.value)
print(json.dumps(dict(lib=os.path.basename(sys.argv[1]), log2_paths=n.bit_length() - 1, mean_ms=sum(ts) / len(ts), min_ms=min(ts), path_steps_per_s=n * 1024 / (sum(ts) / len(ts)) * 1e3)))
