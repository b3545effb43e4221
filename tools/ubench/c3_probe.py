"""C3's kernels alone (for rocprofv3 counter passes): Heston Euler / QE, both parameter sets, 2^22 paths x 512 steps each"""
import ctypes as C, os, sys
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "stochvolmodels_amd", "libsvmc.so"))
vp, f64, sz, i32, u64, u32 = C.c_void_p, C.c_double, C.c_size_t, C.c_int, C.c_uint64, C.c_uint32
L.svmc_malloc.argtypes = [C.POINTER(vp), sz]
L.svmc_fill_state.argtypes = [vp, vp, vp, sz, f64, f64, f64, vp]
L.svmc_heston_terminal_rng.argtypes = [vp, vp, vp, sz, i32, f64, f64, f64, f64, f64, i32, u64, u32, u64, u32, vp]
L.svmc_stream_synchronize.argtypes = [vp]
n = 1 << 22
h = [vp() for _ in range(3)]
for b in h:
    assert L.svmc_malloc(C.byref(b), 8 * n) == 0
for name, (v0, th, ka, rho, vv) in (("base", (0.04, 0.04, 4.0, -0.5, 0.4)), ("btc", (0.8, 1.0, 2.0, 0.0, 2.0))):
    for scheme in (0, 1):
        for rep in range(3):
            L.svmc_fill_state(h[0], h[1], h[2], n, 0.0, v0, 0.0, None)
            assert L.svmc_heston_terminal_rng(h[0], h[1], h[2], n, 512, 1 / 512, th, ka, rho, vv, scheme, 7, 0, 0, 0, None) == 0
        L.svmc_stream_synchronize(None)
print("order of heston_rng_kernel dispatches: base euler x3, base qe x3, btc euler x3, btc qe x3")
