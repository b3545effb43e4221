"""kernel rate of the LogSV stepping kernel vs launch size (the launch tail): 2^18 .. 2^23 paths x 1024 steps"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.getcwd())
L = C.CDLL(os.path.abspath(sys.argv[1])) if len(sys.argv) > 1 else __import__("stochvolmodels_amd._lib", fromlist=["_lib"]).load()
vp = C.c_void_p
if len(sys.argv) > 1:
    f64, sz, i32, u64, u32 = C.c_double, C.c_size_t, C.c_int, C.c_uint64, C.c_uint32
    L.svmc_malloc.argtypes = [C.POINTER(vp), sz]; L.svmc_free.argtypes = [vp]
    L.svmc_fill_state.argtypes = [vp, vp, vp, sz, f64, f64, f64, vp]
    L.svmc_logsv_terminal_rng.argtypes = [vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, f64, f64, i32, u64, u32, u64, u32, vp]
    L.svmc_event_create.argtypes = [C.POINTER(vp)]; L.svmc_event_record.argtypes = [vp, vp]
    L.svmc_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]; L.svmc_stream_synchronize.argtypes = [vp]
out = {}
for lg in (20, 21):
    n = 1 << lg
    b = [vp() for _ in range(3)]
    for x in b:
        assert L.svmc_malloc(C.byref(x), 8 * n) == 0
    ts = []
    for rep in range(8):
        L.svmc_fill_state(b[0], b[1], b[2], n, 0.0, 0.8376, 0.0, None)
        e0, e1 = vp(), vp()
        L.svmc_event_create(C.byref(e0)); L.svmc_event_create(C.byref(e1))
        L.svmc_event_record(e0, None)
        assert L.svmc_logsv_terminal_rng(b[0], b[1], b[2], n, 1024, 1 / 1024, 1.0413, 3.1844, 3.058, 0.1514, 1.8458, 1.0, 1, 7, 0, 0, 0, None) == 0
        L.svmc_event_record(e1, None)
        L.svmc_stream_synchronize(None)
        ms = C.c_float(); L.svmc_event_elapsed_ms(e0, e1, C.byref(ms)); ts.append(ms.value)
    for x in b:
        L.svmc_free(x)
    t = sorted(ts[2:])[len(ts[2:]) // 2]
    out[f"2^{lg}"] = {"ms": round(t, 4), "path_steps_per_s": round(n * 1024 / (t * 1e-3) / 1e11, 3)}
print(json.dumps({"lib": sys.argv[1] if len(sys.argv) > 1 else "in-tree", **out}))
