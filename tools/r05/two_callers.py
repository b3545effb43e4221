#!/usr/bin/env python
"""C2's call (2^20 paths x 1024 steps, 21 strikes) from ONE host thread in a loop, and from TWO threads at once (each its own
engine: its own buffers, the launches of both queue on the device): aggregate path-steps/s.  With two callers the device always
has the other caller's stepping kernel queued while one caller's payoff pass, download and host work run -- what a throughput-
oriented host gets, against the latency-oriented loop the bench line times.  One JSON line."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402

P = sv.LOGSV_BTC_PARAMS
kk = np.linspace(0.6, 1.6, 21)
ty = np.where(kk >= 1.0, "C", "P")
wl = dict(ttms=np.array([1.0]), forwards=np.ones(1), discfactors=np.ones(1), strikes_ttms=(kk,), optiontypes_ttms=(ty,))


def call(seed):
    return sv.logsv_mc_chain_pricer(v0=P.sigma0, theta=P.theta, kappa1=P.kappa1, kappa2=P.kappa2, beta=P.beta, volvol=P.volvol,
                                    vol_backbone_etas=np.ones(1), nb_path=1 << 20, nb_steps_per_year=1023, seed=seed, **wl)


def loop(n, seed0, out, i):
    for j in range(10):
        call(seed0 + j)
    t0 = time.perf_counter()
    for j in range(n):
        call(seed0 + 100 + j)
    out[i] = (t0, time.perf_counter())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    res = {}
    out = [None]
    loop(n, 1, out, 0)
    res["one_caller_ms_per_call"] = 1e3 * (out[0][1] - out[0][0]) / n
    res["one_caller_psps"] = (1 << 30) / (res["one_caller_ms_per_call"] * 1e-3)
    for k in (2, 3):
        out = [None] * k
        th = [threading.Thread(target=loop, args=(n, 1000 * (i + 1), out, i)) for i in range(k)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        wall = max(o[1] for o in out) - min(o[0] for o in out)
        res[f"{k}_callers_aggregate_ms_per_call"] = 1e3 * wall / (k * n)
        res[f"{k}_callers_psps"] = k * n * (1 << 30) / wall
    a, b = call(77), call(77)
    res["deterministic"] = bool(all(np.array_equal(x, y) for x, y in zip(a[0] + a[1], b[0] + b[1])))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
