#!/usr/bin/env python
"""
The frozen-randoms chain for 1..8 parameter sets: median wall time of the public batched pricer and a digest of the prices,
per library (SVMC_LIB: the A/B builds of tools/ubench/build_variants.sh), one process per library.

    python tools/r05/frozen_sets_timing.py --child nb_path calls          one JSON line for the library SVMC_LIB names
    python tools/r05/frozen_sets_timing.py [nb_path] [calls] [lib.so ...] the in-tree library and the given ones, side by side
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(nb_path, calls):
    import numpy as np
    import stochvolmodels_amd as sv
    ttms = np.array([1 / 12, 0.25, 0.5, 1.0])
    k = np.linspace(0.7, 1.3, 13)
    ty = np.where(k >= 1.0, "C", "P")
    chain = dict(ttms=ttms, forwards=np.ones(4), discfactors=np.ones(4), strikes_ttms=(k,) * 4, optiontypes_ttms=(ty,) * 4)
    p = sv.LOGSV_BTC_PARAMS
    sets = [sv.LogSvParams(sigma0=p.sigma0 + 1e-3 * j, theta=p.theta, kappa1=p.kappa1 + 1e-2 * j, kappa2=p.kappa2, beta=p.beta,
                           volvol=p.volvol - 1e-2 * j) for j in range(8)]
    res = sv.draw_fixed_randoms_on_device(ttms, nb_path=nb_path, nb_steps_per_year=360, seed=10)
    out = {}
    for n_sets in range(1, 9):
        fn = lambda: sv.logsv_mc_chain_pricer_fixed_randoms_batch(params_list=sets[:n_sets], W0s=res, return_ivols=True, **chain)  # noqa: E731
        got = fn()
        fn()
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        digest = float(sum(float(np.sum(a)) for r in got for part in r[:2] for a in part)).hex()
        out[str(n_sets)] = [round(1e3 * float(np.median(ts)), 4), digest]
    res.free()
    print(json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]), int(sys.argv[3]))
    nb_path = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    table = {}
    for lib in [None] + sys.argv[3:]:
        env = dict(os.environ)
        env.pop("SVMC_LIB", None)
        if lib:
            env["SVMC_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(nb_path), str(calls)], env=env,
                           capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        table[os.path.basename(lib) if lib else "in-tree"] = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
    base = table["in-tree"]
    rows = {}
    for tag, row in table.items():
        rows[tag] = row if "error" in row else {n: {"ms": v[0], "same_bits": v[1] == base[n][1]} for n, v in row.items()}
    print(json.dumps({"nb_path": nb_path, "calls": calls, "table": rows}))


if __name__ == "__main__":
    main()
