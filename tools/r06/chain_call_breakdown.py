#!/usr/bin/env python
"""
Where one logsv_mc_chain_pricer call goes at the reference's path counts (4 x 13 chain x 364 steps): wall time per call through
the Python host and through ONE svmc_logsv_chain_price call of the C ABI, and -- from a rocprofv3 kernel trace of the same loop --
the average duration of every kernel of a call, the idle time between consecutive kernels of a call and the host time between
the last kernel of a call and the first of the next.

    python tools/r06/chain_call_breakdown.py            # drives everything (spawns itself under rocprofv3), prints one JSON line
    python tools/r06/chain_call_breakdown.py --child py|c N CALLS
"""
import ctypes as C
import json
import os
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SIZES = (65536, 100000, 200000, 400000)


def chain4():
    import numpy as np
    k = np.linspace(0.7, 1.3, 13)
    ty = np.where(k >= 1.0, "C", "P")
    ttms = np.array([1 / 12, 0.25, 0.5, 1.0])
    return ttms, np.ones(4), np.ones(4), (k,) * 4, (ty,) * 4


def child(route, n, calls):
    import gc
    import numpy as np
    import stochvolmodels_amd as sv
    from stochvolmodels_amd import _lib
    from stochvolmodels_amd.engine import option_type_codes
    p = sv.LOGSV_BTC_PARAMS
    ttms, fw, df, ks, tys = chain4()
    if route == "py":
        fn = lambda seed: sv.logsv_mc_chain_pricer(ttms=ttms, forwards=fw, discfactors=df, strikes_ttms=ks, optiontypes_ttms=tys,  # noqa: E731
                                                   v0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta,
                                                   volvol=p.volvol, vol_backbone_etas=np.ones(4), nb_path=n, nb_steps_per_year=360,
                                                   seed=seed)
    else:
        L = _lib.load()
        sess = C.c_void_p()
        kk = np.ascontiguousarray(np.concatenate(ks))
        codes = np.ascontiguousarray(np.concatenate([option_type_codes(t) for t in tys]).astype(np.int8))
        offs = (C.c_size_t * 5)(0, 13, 26, 39, 52)
        _lib.check(L.svmc_session_create(C.byref(sess), n, 4, kk.size))
        prices, errs, etas = np.empty(kk.size), np.empty(kk.size), np.ones(4)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))        # noqa: E731

        def fn(seed):
            _lib.check(L.svmc_logsv_chain_price(sess, dp(ttms), dp(fw), dp(df), dp(etas), 4, dp(kk), codes.ctypes.data_as(C.POINTER(C.c_int8)),
                                                offs, p.sigma0, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, 1, 360, 1, seed, 0,
                                                dp(prices), dp(errs)))
            return prices
    for i in range(20):
        fn(i)
    gc.collect()
    gc.freeze()
    ts = []
    for i in range(calls):
        t0 = time.perf_counter()
        fn(100 + i)
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"route": route, "n": n, "wall_ms_median": round(1e3 * float(np.median(ts)), 4),
                      "wall_ms_min": round(1e3 * float(np.min(ts)), 4)}), flush=True)


def timeline(db_path, calls):
    """per-kernel average duration over the LAST `calls` calls, the gaps inside a call and between calls"""
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else ("kernel_name" if "kernel_name" in cols else cols[0])
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    # a call starts at each stepping kernel (name contains 'rng')
    starts = [i for i, r in enumerate(rows) if "rng" in r[0] and "kernel" in r[0]]
    starts = starts[-calls:]
    per_kernel, inner_gap, outer_gap, span = {}, [], [], []
    for a, b in zip(starts[:-1], starts[1:]):
        seq = rows[a:b]
        span.append(seq[-1][2] - seq[0][1])
        outer_gap.append(rows[b][1] - seq[-1][2])
        for j, (nm, s, e) in enumerate(seq):
            short = nm.split("(")[0][:48]
            per_kernel.setdefault((j, short), []).append(e - s)
            if j > 0:
                inner_gap.append(s - seq[j - 1][2])
    avg = lambda v: round(sum(v) / max(len(v), 1) / 1e3, 2)        # noqa: E731
    return {"kernels_us": [{"k": k[1], "avg_us": avg(v)} for k, v in sorted(per_kernel.items())],
            "first_kernel_to_last_kernel_end_us": avg(span), "idle_between_kernels_of_a_call_us_each": avg(inner_gap),
            "last_kernel_end_to_next_call_first_kernel_us": avg(outer_gap)}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    calls = 200
    out = {}
    for n in SIZES:
        for route in ("py", "c"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", route, str(n), str(calls)], capture_output=True,
                               text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            rec = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
            with tempfile.TemporaryDirectory(prefix="svmc_kt_", dir="/tmp") as tmp:
                env = dict(os.environ, TMPDIR="/tmp")
                r = subprocess.run(["rocprofv3", "--kernel-trace", "-d", tmp, "-o", "t", "--", sys.executable, os.path.abspath(__file__),
                                    "--child", route, str(n), str(calls)], capture_output=True, text=True, timeout=900, cwd="/tmp", env=env)
                dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp) for f in fs if f.endswith(".db")]
                if dbs:
                    try:
                        rec["trace"] = timeline(dbs[0], calls)
                    except Exception as exc:                     # noqa: BLE001
                        rec["trace_error"] = repr(exc)
                else:
                    rec["trace_error"] = r.stderr[-300:]
            out[f"{route}_{n}"] = rec
    print(json.dumps(out))


if __name__ == "__main__":
    main()
