"""The chain-wide payoff pass alone at C3's shape (2^22 paths x 4 expiries x 21 strikes), a few passes: the target of the
counter runs of profiles/r02_payoff_pmc.txt.

    python tools/ubench/payoff_probe.py [lib.so] [passes]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "stochvolmodels_amd", "libsvmc.so")
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
L = C.CDLL(os.path.abspath(lib))
vp, f64, sz, i32, u64, u32 = C.c_void_p, C.c_double, C.c_size_t, C.c_int, C.c_uint64, C.c_uint32
pd, pi8, psz = C.POINTER(C.c_double), C.POINTER(C.c_int8), C.POINTER(C.c_size_t)
L.svmc_malloc.argtypes = [C.POINTER(vp), sz]
L.svmc_fill_state.argtypes = [vp, vp, vp, sz, f64, f64, f64, vp]
L.svmc_heston_terminal_rng.argtypes = [vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, i32, u64, u32, u64, u32, vp]
L.svmc_stream_synchronize.argtypes = [vp]
L.svmc_slice_workspace_bytes.argtypes = [sz, psz]
L.svmc_spot_sums.argtypes = [vp, sz, f64, vp, vp, sz, vp]
L.svmc_memcpy_d2d.argtypes = [vp, vp, sz, vp]
L.svmc_payoff_sums_chain.argtypes = [C.POINTER(vp), C.POINTER(vp), sz, pd, pd, vp, i32, pd, pi8, pd, psz, i32, vp, vp, sz, vp]


def bufs(n, k):
    out = [vp() for _ in range(k)]
    for x in out:
        assert L.svmc_malloc(C.byref(x), 8 * n) == 0
    return out


n = 1 << 22
h = bufs(n, 3)
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
for _ in range(passes):
    assert L.svmc_payoff_sums_chain(xs, None, n, fw.ctypes.data_as(pd), tt.ctypes.data_as(pd), spot, 4, kk.ctypes.data_as(pd),
                                    ty.ctypes.data_as(pi8), sh.ctypes.data_as(pd), offs, 1, sums, ws, wsb.value, None) == 0
L.svmc_stream_synchronize(None)
print("ok")
