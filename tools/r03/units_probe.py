"""(goes with commit e53a51a, where the dynamic-units launch lived: SVMC_UNIT_* are read by that build only)
A/B of the dynamic-units launch of the LogSV generator (SVMC_UNIT_STEPS=0: one-round kernel): kernel time by HIP events
and a digest of the outputs (x, sigma, qvar, snapshot, spot sums) -- the two launch forms must agree to the bit.
usage: [SVMC_UNIT_STEPS=...] units_probe.py [log2 sizes ...] [--steps N]"""
import ctypes as C, hashlib, json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from stochvolmodels_amd import _lib
if os.environ.get("SVMC_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SVMC_PROBE_LIB"])
L = _lib.load()
vp = C.c_void_p
args = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 1024
if "--steps" in sys.argv:
    args.remove(str(steps))
sizes = [int(a) for a in args] or [20, 21]
out = {"unit_steps": os.environ.get("SVMC_UNIT_STEPS", "default"), "steps": steps}
for lg in sizes:
    n = (1 << lg) if lg < 64 else lg
    ws = C.c_size_t()
    _lib.check(L.svmc_slice_workspace_bytes(n, C.byref(ws)))
    b = [vp() for _ in range(6)]
    for x, nbytes in zip(b, (8 * n, 8 * n, 8 * n, 8 * n, 16, ws.value)):
        _lib.check(L.svmc_malloc(C.byref(x), nbytes))
    ts = []
    for rep in range(8):
        e0, e1 = vp(), vp()
        L.svmc_event_create(C.byref(e0)); L.svmc_event_create(C.byref(e1))
        L.svmc_event_record(e0, None)
        _lib.check(L.svmc_logsv_slice_rng_from(0.0, 0.8376, 0.0, b[0], b[1], b[2], n, steps, 1.0 / steps, 1.0413, 3.1844, 3.058,
                                               0.1514, 1.8458, 1.0, 1, 7, 3, 0, 0, 1.0, b[3], None, b[4], b[5], ws.value, None))
        L.svmc_event_record(e1, None)
        _lib.check(L.svmc_stream_synchronize(None))
        ms = C.c_float(); L.svmc_event_elapsed_ms(e0, e1, C.byref(ms)); ts.append(ms.value)
    def fetch():
        res = []
        for x, cnt in zip(b[:5], (n, n, n, n, 2)):
            a = np.empty(cnt)
            _lib.check(L.svmc_memcpy_d2h(a.ctypes.data, x, 8 * cnt, None)); _lib.check(L.svmc_stream_synchronize(None))
            res.append(a)
        return res
    got = fetch()
    h = hashlib.sha256()
    for a in got:
        h.update(a.tobytes())
    sums = got[4].tolist()
    # the one-round kernel in the same process: a launch that starts from the arrays never runs as dynamic units
    _lib.check(L.svmc_fill_state(b[0], b[1], b[2], n, 0.0, 0.8376, 0.0, None))
    _lib.check(L.svmc_logsv_slice_rng(b[0], b[1], b[2], n, steps, 1.0 / steps, 1.0413, 3.1844, 3.058, 0.1514, 1.8458, 1.0, 1, 7, 3,
                                      0, 0, 1.0, b[3], None, b[4], b[5], ws.value, None))
    ref = fetch()
    diff = {}
    for name, a, r in zip(("x", "sigma", "qvar", "snap", "sums"), got, ref):
        bad = np.flatnonzero(a.view(np.uint64) != r.view(np.uint64))
        if bad.size:
            i = int(bad[0])
            diff[name] = {"n": int(bad.size), "first": i, "got": float(a[i]), "ref": float(r[i]), "max_rel": float(np.max(np.abs(a[bad] / r[bad] - 1)))}
    for x in b:
        L.svmc_free(x)
    t = sorted(ts[2:])
    out[f"{lg}"] = {"ms_median": round(t[len(t) // 2], 4), "ms_min": round(t[0], 4),
                    "path_steps_per_s_e11": round(n * steps / (t[len(t) // 2] * 1e-3) / 1e11, 3), "digest": h.hexdigest()[:16], "sums": sums, "differs_from_one_round": diff}
print(json.dumps(out))
