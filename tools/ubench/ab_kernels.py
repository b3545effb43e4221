"""A/B of the generator kernels between builds of libsvmc on the SAME box, raw ctypes, one library per process:

    python tools/ubench/ab_kernels.py <lib.so> [tag]

Times (HIP events, 12 launches after 3 warm-ups): LogSV on-device RNG at C2 (2^20 x 1024), Heston Euler and QE at C3's
shape (2^22 x 512, both parameter sets), and the chain-wide payoff pass at C3's shape (2^22 paths x 4 x 21 strikes).
Only symbols every build since round 1 exports are used."""
import ctypes as C, json, os, sys
L = C.CDLL(os.path.abspath(sys.argv[1]))
tag = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(sys.argv[1])
vp, f64, sz, i32, u64, u32 = C.c_void_p, C.c_double, C.c_size_t, C.c_int, C.c_uint64, C.c_uint32
L.svmc_malloc.argtypes = [C.POINTER(vp), sz]
L.svmc_fill_state.argtypes = [vp, vp, vp, sz, f64, f64, f64, vp]
L.svmc_logsv_terminal_rng.argtypes = [vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, f64, f64, i32, u64, u32, u64, u32, vp]
L.svmc_heston_terminal_rng.argtypes = [vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, i32, u64, u32, u64, u32, vp]
L.svmc_event_create.argtypes = [C.POINTER(vp)]
L.svmc_event_record.argtypes = [vp, vp]
L.svmc_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
L.svmc_stream_synchronize.argtypes = [vp]


def bufs(n, k=3):
    out = [vp() for _ in range(k)]
    for x in out:
        assert L.svmc_malloc(C.byref(x), 8 * n) == 0
    return out


def timed(launch, prep, reps=12, warm=3):
    for _ in range(warm):
        prep(); launch()
    L.svmc_stream_synchronize(None)
    ts = []
    for _ in range(reps):
        prep()
        e0, e1 = vp(), vp()
        L.svmc_event_create(C.byref(e0)); L.svmc_event_create(C.byref(e1))
        L.svmc_event_record(e0, None)
        assert launch() == 0
        L.svmc_event_record(e1, None)
        L.svmc_stream_synchronize(None)
        ms = C.c_float(); L.svmc_event_elapsed_ms(e0, e1, C.byref(ms)); ts.append(ms.value)
    return sum(ts) / len(ts), min(ts)


res = {"lib": tag}
n = 1 << 20
b = bufs(n)
m, lo = timed(lambda: L.svmc_logsv_terminal_rng(b[0], b[1], b[2], n, 1024, 1 / 1024, 1.0413, 3.1844, 3.058, 0.1514, 1.8458, 1.0, 1, 7, 0, 0, 0, None),
              lambda: L.svmc_fill_state(b[0], b[1], b[2], n, 0.0, 0.8376, 0.0, None))
res["logsv_c2_ms"], res["logsv_c2_min_ms"] = round(m, 4), round(lo, 4)
# the whole-chain kernel at C4's rank share: 2^21 paths, 8 x 128 steps (svmc_logsv_chain_rng, fused epilogues)
import numpy as np
n = 1 << 21
cb = bufs(n)
pd_ = C.POINTER(C.c_double)
L.svmc_slice_workspace_bytes.argtypes = [sz, C.POINTER(C.c_size_t)]
L.svmc_logsv_chain_rng.argtypes = [vp, vp, vp, sz, i32, C.POINTER(i32), pd_, pd_, pd_, f64, f64, f64, f64, f64, i32, u64, u32,
                                   u64, u32, vp, vp, vp, vp, sz, vp]
wsb_c = C.c_size_t()
assert L.svmc_slice_workspace_bytes(n, C.byref(wsb_c)) == 0
ws_c, spot_c, snap_c = vp(), vp(), vp()
assert L.svmc_malloc(C.byref(ws_c), wsb_c.value) == 0 and L.svmc_malloc(C.byref(spot_c), 256) == 0
assert L.svmc_malloc(C.byref(snap_c), 8 * 8 * n) == 0
nbs = (i32 * 8)(*([128] * 8))
dts = np.full(8, 1.0 / 1024); etas = np.ones(8); fws = 67000.0 * np.exp(0.05 * np.arange(1, 9) / 8)
m, lo = timed(lambda: L.svmc_logsv_chain_rng(cb[0], cb[1], cb[2], n, 8, nbs, dts.ctypes.data_as(pd_), etas.ctypes.data_as(pd_),
                                             fws.ctypes.data_as(pd_), 1.0413, 3.1844, 3.058, 0.1514, 1.8458, 1, 7, 0, 0, 0,
                                             snap_c, None, spot_c, ws_c, wsb_c.value, None),
              lambda: L.svmc_fill_state(cb[0], cb[1], cb[2], n, 0.0, 0.8376, 0.0, None), reps=8, warm=2)
res["logsv_chain_c4_ms"], res["logsv_chain_c4_min_ms"] = round(m, 4), round(lo, 4)
n = 1 << 22
h = bufs(n)
for name, (v0, th, ka, rho, vv) in (("base", (0.04, 0.04, 4.0, -0.5, 0.4)), ("btc", (0.8, 1.0, 2.0, 0.0, 2.0))):
    for sname, scheme in (("euler", 0), ("qe", 1)):
        m, lo = timed(lambda: L.svmc_heston_terminal_rng(h[0], h[1], h[2], n, 512, 1 / 512, th, ka, rho, vv, scheme, 7, 0, 0, 0, None),
                      lambda: L.svmc_fill_state(h[0], h[1], h[2], n, 0.0, v0, 0.0, None), reps=6, warm=2)
        res[f"heston_{name}_{sname}_ms"] = round(m, 4)
# the whole-chain Heston kernel at C3's shape (4 x 128 steps, fused epilogues), when the build has it
if hasattr(L, "svmc_heston_chain_rng"):
    L.svmc_heston_chain_rng.argtypes = [vp, vp, vp, sz, i32, C.POINTER(i32), pd_, pd_, f64, f64, f64, f64, i32, u64, u32, u64, u32,
                                        vp, vp, vp, vp, sz, vp]
    wsb_h = C.c_size_t()
    assert L.svmc_slice_workspace_bytes(n, C.byref(wsb_h)) == 0
    ws_h, spot_h, snap_h = vp(), vp(), vp()
    assert L.svmc_malloc(C.byref(ws_h), wsb_h.value) == 0 and L.svmc_malloc(C.byref(spot_h), 256) == 0
    assert L.svmc_malloc(C.byref(snap_h), 8 * 4 * n) == 0
    nbs4 = (i32 * 4)(*([128] * 4))
    dts4 = np.full(4, 1.0 / 512); fws4 = np.ones(4)
    for name, (v0, th, ka, rho, vv) in (("base", (0.04, 0.04, 4.0, -0.5, 0.4)), ("btc", (0.8, 1.0, 2.0, 0.0, 2.0))):
        for sname, scheme in (("euler", 0), ("qe", 1)):
            m, lo = timed(lambda: L.svmc_heston_chain_rng(h[0], h[1], h[2], n, 4, nbs4, dts4.ctypes.data_as(pd_), fws4.ctypes.data_as(pd_),
                                                          th, ka, rho, vv, scheme, 7, 0, 0, 0, snap_h, None, spot_h, ws_h, wsb_h.value, None),
                          lambda: L.svmc_fill_state(h[0], h[1], h[2], n, 0.0, v0, 0.0, None), reps=6, warm=2)
            res[f"heston_chain_{name}_{sname}_ms"] = round(m, 4)
# ---- the chain-wide payoff pass at C3's shape: 4 expiries x 21 strikes over 2^22 paths each (P below 1, C at/above) ----
pd, pi8, psz = C.POINTER(C.c_double), C.POINTER(C.c_int8), C.POINTER(C.c_size_t)
L.svmc_slice_workspace_bytes.argtypes = [sz, psz]
L.svmc_spot_sums.argtypes = [vp, sz, f64, vp, vp, sz, vp]
L.svmc_memcpy_d2d.argtypes = [vp, vp, sz, vp]
L.svmc_payoff_sums_chain.argtypes = [C.POINTER(vp), C.POINTER(vp), sz, pd, pd, vp, i32, pd, pi8, pd, psz, i32, vp, vp, sz, vp]
wsb = C.c_size_t()
assert L.svmc_slice_workspace_bytes(n, C.byref(wsb)) == 0
ws, spot, sums = vp(), vp(), vp()
assert L.svmc_malloc(C.byref(ws), wsb.value) == 0 and L.svmc_malloc(C.byref(spot), 64) == 0 and L.svmc_malloc(C.byref(sums), 8 * 3 * 84) == 0
L.svmc_fill_state(h[0], h[1], h[2], n, 0.0, 0.04, 0.0, None)
L.svmc_heston_terminal_rng(h[0], h[1], h[2], n, 64, 1 / 256, 0.04, 4.0, -0.5, 0.4, 0, 7, 0, 0, 0, None)
snaps = bufs(n, 4)
for i in range(4):
    L.svmc_memcpy_d2d(snaps[i], h[0], 8 * n, None)
    assert L.svmc_spot_sums(snaps[i], n, 1.0, vp(spot.value + 16 * i), ws, wsb.value, None) == 0
kk = np.tile(np.linspace(0.5, 1.5, 21), 4)
ty = np.where(kk >= 1.0, 0, 1).astype(np.int8)
sh = np.where(ty == 0, np.maximum(1.0 - kk, 0), np.maximum(kk - 1.0, 0))
offs = (C.c_size_t * 5)(0, 21, 42, 63, 84)
fw, tt = np.ones(4), np.array([0.25, 0.5, 0.75, 1.0])
xs = (vp * 4)(*[b_.value for b_ in snaps])
m, lo = timed(lambda: L.svmc_payoff_sums_chain(xs, None, n, fw.ctypes.data_as(pd), tt.ctypes.data_as(pd), spot, 4,
                                               kk.ctypes.data_as(pd), ty.ctypes.data_as(pi8), sh.ctypes.data_as(pd), offs, 1,
                                               sums, ws, wsb.value, None), lambda: None, reps=20, warm=3)
res["payoff_c3_us"], res["payoff_c3_min_us"] = round(1e3 * m, 2), round(1e3 * lo, 2)


def payoff_x16():
    rc = 0
    for _ in range(16):
        rc |= L.svmc_payoff_sums_chain(xs, None, n, fw.ctypes.data_as(pd), tt.ctypes.data_as(pd), spot, 4, kk.ctypes.data_as(pd),
                                       ty.ctypes.data_as(pi8), sh.ctypes.data_as(pd), offs, 1, sums, ws, wsb.value, None)
    return rc


# the same pass 16 times back to back: in a chain pricing it follows milliseconds of stepping, not an idle queue
m, lo = timed(payoff_x16, lambda: None, reps=8, warm=2)
res["payoff_c3_back_to_back_us"] = round(1e3 * m / 16, 2)
print(json.dumps(res), flush=True)
