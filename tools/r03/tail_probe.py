"""Per-wave start / end stamps of one C2 launch of the LogSV stepping kernel (a library built with -DSVMC_TAIL_PROBE):
how many waves are running at each moment, when the rounds change over, how long the last waves run alone."""
import ctypes as C, json, os, sys
import numpy as np
L = C.CDLL(os.path.abspath(sys.argv[1]))
n = 1 << int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
vp, f64, sz, i32, u64, u32 = C.c_void_p, C.c_double, C.c_size_t, C.c_int, C.c_uint64, C.c_uint32
L.svmc_malloc.argtypes = [C.POINTER(vp), sz]
L.svmc_fill_state.argtypes = [vp, vp, vp, sz, f64, f64, f64, vp]
L.svmc_logsv_slice_rng.argtypes = [vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, f64, f64, i32, u64, u32, u64, u32, f64, vp, vp, vp, vp, sz, vp]
L.svmc_slice_workspace_bytes.argtypes = [sz, C.POINTER(sz)]
L.svmc_memcpy_d2h.argtypes = [vp, vp, sz, vp]
L.svmc_stream_synchronize.argtypes = [vp]
b = [vp() for _ in range(6)]
wsb = sz(); L.svmc_slice_workspace_bytes(n, C.byref(wsb))
for x, nb in zip(b, (8 * n, 8 * n, 8 * n, 8 * n, 8 * n, wsb.value)):
    assert L.svmc_malloc(C.byref(x), max(nb, 64)) == 0
spot = vp(); L.svmc_malloc(C.byref(spot), 64)
res = {}
for rep in range(4):
    L.svmc_fill_state(b[0], b[1], b[2], n, 0.0, 0.8376, 0.0, None)
    assert L.svmc_logsv_slice_rng(b[0], b[1], b[2], n, 1024, 1 / 1024, 1.0413, 3.1844, 3.058, 0.1514, 1.8458, 1.0, 1, 7, 0, 0, 0, 1.0, b[3], b[4], spot, b[5], wsb.value, None) == 0
    L.svmc_stream_synchronize(None)
w = n // 64
st = np.empty(2 * w)
L.svmc_memcpy_d2h(st.ctypes.data, b[4], 16 * w, None); L.svmc_stream_synchronize(None)
t0, t1 = st[0::2], st[1::2]
base = t0.min()
t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0          # microseconds
span = t1.max()
life = t1 - t0
grid = np.linspace(0, span, 41)
running = [(int(((t0 <= g) & (t1 > g)).sum())) for g in grid]
res = {"paths": n, "waves": w, "span_us": round(float(span), 1), "wave_life_us": {"min": round(float(life.min()), 1), "median": round(float(np.median(life)), 1), "max": round(float(life.max()), 1)},
       "start_us_percentiles": [round(float(v), 1) for v in np.percentile(t0, [0, 25, 50, 75, 100])],
       "end_us_percentiles": [round(float(v), 1) for v in np.percentile(t1, [0, 25, 50, 75, 90, 99, 100])],
       "running_waves_at_40ths_of_span": running, "sum_of_lives_over_span_x_slots": round(float(life.sum() / (span * 8192)), 4)}
print(json.dumps(res))
