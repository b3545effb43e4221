import sys, os, ctypes as C, json
sys.path.insert(0, os.getcwd())
import numpy as np
from stochvolmodels_amd.engine import get_engine, option_type_codes, payoff_shifts
from stochvolmodels_amd import _lib
L = _lib.load()
def ev():
    e = C.c_void_p(); _lib.check(L.svmc_event_create(C.byref(e))); return e
for lg in (16, 20, 22):
    n = 1 << lg
    eng = get_engine(n)
    eng.fill_state(0.0, 0.5, 0.02)
    eng.logsv_rng(16, 1/360, 1.0, 3.0, 3.0, 0.15, 1.8, 1.0, True, 3, 0, 0)
    eng.reserve_snapshots(1); eng.snapshot(0, "x")
    sp, _ = eng.alloc_sums(2, "spot"); eng.spot_sums(eng.snapshot_ptr(0), 1.0, sp)
    for ns in (1, 8, 9, 21, 32, 64):
        k = np.linspace(0.6, 1.4, ns); codes = option_type_codes(np.where(k >= 1, "C", "P")); sh = payoff_shifts(k, codes, 1.0, 1)
        out, _ = eng.alloc_sums(3 * ns + 1, "pay")
        f = lambda: eng.payoff_sums(eng.snapshot_ptr(0), None, 1.0, 0.25, sp, k, codes, sh, 1, out)
        for _ in range(3): f()
        e0, e1 = ev(), ev()
        L.svmc_event_record(e0, None)
        for _ in range(20): f()
        L.svmc_event_record(e1, None); eng.synchronize()
        ms = C.c_float(); L.svmc_event_elapsed_ms(e0, e1, C.byref(ms))
        print(json.dumps(dict(log2_paths=lg, strikes=ns, us_per_call=ms.value * 50)))
