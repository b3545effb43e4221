#!/usr/bin/env python
"""
The on-device-RNG pricers at default- and calibration-sized path counts: the few-waves kernels (logsv_rng_few_kernel,
logsv_chain_rng_few_kernel, heston_*_few_kernel) against the full-launch kernels (SVMC_FEW_WAVES_MAX_PATHS=0), wall time of the
public pricer for a LogSV 4 x 13 chain (364 steps), a one-expiry chain (360 steps) and a Heston 4 x 13 chain (Euler, QE), and
whether the prices are the same bits.  One JSON line.

    python tools/r05/few_waves_sweep.py [calls]
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SIZES = (16384, 65536, 100000, 131072, 200000, 262144, 400000, 524288)


def child(calls):
    import numpy as np
    import stochvolmodels_amd as sv
    k = np.linspace(0.7, 1.3, 13)
    ty = np.where(k >= 1.0, "C", "P")
    p = sv.LOGSV_BTC_PARAMS
    out = {}
    for tag, ttms in (("chain4", np.array([1 / 12, 0.25, 0.5, 1.0])), ("one", np.array([1.0]))):
        m = len(ttms)
        chain = dict(ttms=ttms, forwards=np.ones(m), discfactors=np.ones(m), strikes_ttms=(k,) * m, optiontypes_ttms=(ty,) * m)
        row = {}
        for n in SIZES:
            fn = lambda: sv.logsv_mc_chain_pricer(v0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta,  # noqa: E731
                                                  volvol=p.volvol, vol_backbone_etas=np.ones(m), nb_path=n, nb_steps_per_year=360,
                                                  seed=10, **chain)
            got = fn()
            fn()
            ts = []
            for _ in range(calls):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            row[str(n)] = [round(1e3 * float(np.median(ts)), 4), float(sum(float(np.sum(a)) for a in got[0] + got[1])).hex()]
        out[tag] = row
    h0 = dict(v0=0.04, theta=0.04, kappa=4.0, rho=-0.5, volvol=0.4)
    ttms = np.array([0.25, 0.5, 0.75, 1.0])
    chain = dict(ttms=ttms, forwards=np.ones(4), discfactors=np.ones(4), strikes_ttms=(k,) * 4, optiontypes_ttms=(ty,) * 4)
    for scheme in ("euler", "qe"):
        row = {}
        for n in SIZES:
            fn = lambda: sv.heston_mc_chain_pricer(nb_path=n, scheme=scheme, nb_steps_per_year=360, seed=10, **chain, **h0)  # noqa: E731
            got = fn()
            fn()
            ts = []
            for _ in range(calls):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            row[str(n)] = [round(1e3 * float(np.median(ts)), 4), float(sum(float(np.sum(a)) for a in got[0] + got[1])).hex()]
        out["heston_" + scheme] = row
    print(json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]))
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    res = {}
    for tag, val in (("few_waves", None), ("full_launch_kernels", "0")):
        env = dict(os.environ)
        env.pop("SVMC_FEW_WAVES_MAX_PATHS", None)
        if val is not None:
            env["SVMC_FEW_WAVES_MAX_PATHS"] = val
        if tag == "few_waves":
            env["SVMC_FEW_WAVES_MAX_PATHS"] = str(1 << 30)          # everywhere, to see where it stops paying
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(calls)], env=env, capture_output=True, text=True,
                           timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        res[tag] = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
    out = {"calls": calls}
    for case in ("chain4", "one", "heston_euler", "heston_qe"):
        a, b = res["few_waves"].get(case, {}), res["full_launch_kernels"].get(case, {})
        out[case] = {n: {"few_ms": a[n][0], "full_ms": b[n][0], "same_bits": a[n][1] == b[n][1]} for n in a if n in b}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
