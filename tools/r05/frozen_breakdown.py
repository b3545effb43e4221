#!/usr/bin/env python
"""
Where one frozen-randoms objective evaluation spends its time (4 x 13 chain, 10^5 paths x 364 steps, 1 and 7 sets):
  python_ms   the public function (logsv_mc_chain_pricer_fixed_randoms / _batch), host clock around the call
  c_call_ms   svmc_logsv_chain_price_frozen_sets alone, measured INSIDE those calls (the ctypes function wrapped)
  c_loop_ms   the same entry point called back to back from a loop with prebuilt arguments -- what a C host sees
Medians of `calls`; one JSON line.

    python tools/r05/frozen_breakdown.py [nb_path] [calls]
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402
from stochvolmodels_amd import _lib  # noqa: E402


def main():
    nb_path = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    ttms = np.array([1 / 12, 0.25, 0.5, 1.0])
    k = np.linspace(0.7, 1.3, 13)
    ty = np.where(k >= 1.0, "C", "P")
    chain = dict(ttms=ttms, forwards=np.ones(4), discfactors=np.ones(4), strikes_ttms=(k,) * 4, optiontypes_ttms=(ty,) * 4)
    p = sv.LOGSV_BTC_PARAMS
    sets = [sv.LogSvParams(sigma0=p.sigma0 + 1e-3 * j, theta=p.theta, kappa1=p.kappa1 + 1e-2 * j, kappa2=p.kappa2, beta=p.beta,
                           volvol=p.volvol - 1e-2 * j) for j in range(8)]
    L = _lib.load()
    res = sv.draw_fixed_randoms_on_device(ttms, nb_path=nb_path, nb_steps_per_year=360, seed=10)
    inner, last_args = [], {}
    real = L.svmc_logsv_chain_price_frozen_sets

    class Wrapped:                       # times the C call and keeps its arguments for the back-to-back loop
        def __call__(self, *a):
            t0 = time.perf_counter()
            rc = real(*a)
            inner.append(time.perf_counter() - t0)
            last_args["a"] = a
            return rc
    L.svmc_logsv_chain_price_frozen_sets = Wrapped()
    out = {"nb_path": nb_path, "steps": int(sum(res.nb_steps)), "calls": calls}
    for n_sets in (1, 7):
        if n_sets == 1:
            q = sets[0]
            fn = lambda: sv.logsv_mc_chain_pricer_fixed_randoms(  # noqa: E731
                W0s=res, W1s=None, dts=None, v0=q.sigma0, theta=q.theta, kappa1=q.kappa1, kappa2=q.kappa2, beta=q.beta,
                volvol=q.volvol, vol_backbone_etas=np.ones(4), return_ivols=True, **chain)
        else:
            fn = lambda: sv.logsv_mc_chain_pricer_fixed_randoms_batch(  # noqa: E731
                params_list=sets[:n_sets], W0s=res, return_ivols=True, **chain)
        fn(); fn()
        inner.clear()
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        row = {"python_ms": 1e3 * float(np.median(ts)), "c_call_ms": 1e3 * float(np.median(inner))}
        # the recorded call's result pointers died with its result block: the loop writes into buffers of its own
        dp = C.POINTER(C.c_double)
        mine = [np.empty(n_sets * 52) for _ in range(3)]
        a = tuple(last_args["a"][:-3]) + tuple(b.ctypes.data_as(dp) for b in mine)
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            rc = real(*a)
            ts.append(time.perf_counter() - t0)
        assert rc == 0
        row["c_loop_ms"] = 1e3 * float(np.median(ts))
        row["c_loop_min_ms"] = 1e3 * float(min(ts))
        out[f"{n_sets}_sets"] = row
    L.svmc_logsv_chain_price_frozen_sets = real
    res.free()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
