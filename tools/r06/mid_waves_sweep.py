#!/usr/bin/env python
"""
Round 6: the on-device-RNG pricers between two and eight waves per SIMD (2^17 < paths <= 2^20), every compiled form of the few- /
mid-waves generators (SVMC_GEN_VARIANT = index into LOGSV_LAT_VARIANTS / HESTON_LAT_VARIANTS of svmc_kernels.hip; -1 = the
full-launch kernels) against each other: wall time of the public pricer for a LogSV 4 x 13 chain (364 steps), a one-expiry chain
(360 steps), a Heston 4 x 13 chain (Euler, QE), C2's launch (2^20 x 1024, one expiry) -- and whether the prices are the same bits.

    python tools/r06/mid_waves_sweep.py [calls] [--variants -1,0,1,...] [--product]

--product: one more child with no SVMC_GEN_VARIANT (what the library picks by itself) -- the line the DESIGN table quotes.
One JSON line.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SIZES = (16384, 65536, 100000, 131072, 200000, 262144, 400000, 524288, 786432, 1048576)
if os.environ.get("SVMC_SWEEP_SIZES"):          # e.g. the path counts either side of the few-waves / full-launch switch
    SIZES = tuple(int(v) for v in os.environ["SVMC_SWEEP_SIZES"].split(","))


def timed(fn, calls):
    import numpy as np
    got = fn()
    fn()
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return [round(1e3 * float(np.median(ts)), 4), float(sum(float(np.sum(a)) for a in got[0] + got[1])).hex()]


def child(calls, cases):
    import numpy as np
    import stochvolmodels_amd as sv
    k = np.linspace(0.7, 1.3, 13)
    ty = np.where(k >= 1.0, "C", "P")
    p = sv.LOGSV_BTC_PARAMS
    out = {}
    for tag, ttms in (("chain4", np.array([1 / 12, 0.25, 0.5, 1.0])), ("one", np.array([1.0]))):
        if tag not in cases:
            continue
        m = len(ttms)
        chain = dict(ttms=ttms, forwards=np.ones(m), discfactors=np.ones(m), strikes_ttms=(k,) * m, optiontypes_ttms=(ty,) * m)
        row = {}
        for n in SIZES:
            row[str(n)] = timed(lambda: sv.logsv_mc_chain_pricer(v0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2,
                                                                beta=p.beta, volvol=p.volvol, vol_backbone_etas=np.ones(m), nb_path=n,
                                                                nb_steps_per_year=360, seed=10, **chain), calls)
        out[tag] = row
    h0 = dict(v0=0.04, theta=0.04, kappa=4.0, rho=-0.5, volvol=0.4)
    ttms = np.array([0.25, 0.5, 0.75, 1.0])
    chain = dict(ttms=ttms, forwards=np.ones(4), discfactors=np.ones(4), strikes_ttms=(k,) * 4, optiontypes_ttms=(ty,) * 4)
    for scheme in ("euler", "qe"):
        if "heston_" + scheme not in cases:
            continue
        row = {}
        for n in SIZES:
            row[str(n)] = timed(lambda: sv.heston_mc_chain_pricer(nb_path=n, scheme=scheme, nb_steps_per_year=360, seed=10, **chain, **h0),
                                calls)
        out["heston_" + scheme] = row
    if "c2" in cases:
        k21 = np.linspace(0.5, 1.5, 21)
        ty21 = np.where(k21 >= 1.0, "C", "P")
        c2 = dict(ttms=np.array([1.0]), forwards=np.ones(1), discfactors=np.ones(1), strikes_ttms=(k21,), optiontypes_ttms=(ty21,))
        row = {}
        for n in (1 << 20, 1 << 21):
            row[str(n)] = timed(lambda: sv.logsv_mc_chain_pricer(v0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta,
                                                                volvol=p.volvol, vol_backbone_etas=np.ones(1), nb_path=n,
                                                                nb_steps_per_year=1023, seed=10, **c2), max(20, calls // 4))
        out["c2"] = row
    print(json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]), sys.argv[3].split(","))
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    calls = int(args[0]) if args else 100
    variants = [-1, 0, 1, 2, 3]
    cases = "chain4,one,heston_euler,heston_qe,c2"
    for i, a in enumerate(sys.argv):
        if a == "--variants":
            variants = [int(v) for v in sys.argv[i + 1].split(",")]
        if a == "--cases":
            cases = sys.argv[i + 1]
    runs = [("v%d" % v, str(v)) for v in variants]
    if "--product" in sys.argv:
        runs.append(("product", None))
    res = {}
    for tag, val in runs:
        env = dict(os.environ)
        for key in ("SVMC_GEN_VARIANT", "SVMC_GEN_FORM", "SVMC_FEW_WAVES_MAX_PATHS", "SVMC_MID_WAVES_MAX_PATHS"):
            env.pop(key, None)
        if val is not None:
            env["SVMC_GEN_VARIANT"] = val
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(calls), cases], env=env, capture_output=True,
                           text=True, timeout=1500)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        res[tag] = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
    out = {"calls": calls, "ms": {}, "same_bits": {}}
    for case in cases.split(","):
        tags = [t for t in res if case in res[t]]
        if not tags:
            continue
        sizes = list(res[tags[0]][case])
        out["ms"][case] = {n: {t: res[t][case][n][0] for t in tags} for n in sizes}
        out["same_bits"][case] = {n: len({res[t][case][n][1] for t in tags}) == 1 for n in sizes}
    errs = {t: res[t]["error"] for t in res if "error" in res[t]}
    if errs:
        out["errors"] = errs
    print(json.dumps(out))


if __name__ == "__main__":
    main()
