#!/usr/bin/env python
"""
Randomised end-to-end parity sweep on one MI355X (not part of the pytest suites): random chains -- ragged strike
counts, all payoff codes, both measures, both payoff variables, vol backbones, odd path counts -- priced by
  (a) logsv_mc_chain_pricer / heston_mc_chain_pricer (whole-chain kernels),
  (b) the same slice by slice,
  (c) the CPU oracle on the same counter-based randoms,
requiring (a) == (b) bit for bit and (a) ~ (c) at 1e-9.

    python tests/fuzz_parity.py [n_cases] [seed]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402
from oracle import oracle  # noqa: E402
from stochvolmodels_amd.pricers import heston_pricer as hp, logsv_pricer as lp  # noqa: E402
from stochvolmodels_amd.utils.funcs import set_time_grid  # noqa: E402


def oracle_chain(model, kw, seed):
    n = kw["nb_path"]
    x, v, q = np.zeros(n), np.full(n, kw["v0"]), np.zeros(n)
    t0, step, prices = 0.0, 0, []
    vt = kw["variable_type"].value
    for i, ttm in enumerate(kw["ttms"]):
        nb, dt, _ = set_time_grid(ttm - t0, kw["nb_steps_per_year"])
        if model == "logsv":
            x, v, q = oracle.logsv_terminal_rng(x, v, q, nb, dt, kw["theta"], kw["kappa1"], kw["kappa2"], kw["beta"],
                                                kw["volvol"], seed, eta=float(kw["vol_backbone_etas"][i]),
                                                is_spot_measure=kw["is_spot_measure"], step_offset=step)
        else:
            x, v, q = oracle.heston_terminal_rng(x, v, q, nb, dt, kw["theta"], kw["kappa"], kw["rho"], kw["volvol"], seed,
                                                 scheme=oracle.HESTON_QE if kw["scheme"] == "qe" else oracle.HESTON_EULER_FLOOR,
                                                 step_offset=step)
        p, _ = oracle.payoff(x, q, float(ttm), float(kw["forwards"][i]), kw["strikes_ttms"][i], kw["optiontypes_ttms"][i],
                             float(kw["discfactors"][i]), vt)
        prices.append(p)
        step += nb
        t0 = ttm
    return prices


def main(n_cases=None, seed=None):
    if n_cases is None:
        n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    if seed is None:
        seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    worst, ranking = 0.0, []
    for case in range(n_cases):
        model = ("logsv", "heston")[case % 2]
        m = int(rng.integers(1, 7))
        ttms = np.cumsum(rng.uniform(0.01, 0.12, m))
        fw = np.exp(0.05 * ttms) * rng.uniform(0.5, 2.0)
        vt = sv.VariableType(int(rng.integers(1, 3)))
        counts = rng.integers(1, 12, m)
        if vt == sv.VariableType.Q_VAR:
            strikes = [np.sort(rng.uniform(0.005, 0.6, c)) for c in counts]
            types = [rng.choice(["C", "P"], c) for c in counts]
        else:
            strikes = [f * np.sort(rng.uniform(0.6, 1.5, c)) for c, f in zip(counts, fw)]
            types = [rng.choice(["C", "P", "IC", "IP"], c) for c in counts]
        seed = int(rng.integers(1, 1 << 40))
        kw = dict(ttms=ttms, forwards=fw, discfactors=np.exp(-0.03 * ttms), strikes_ttms=strikes, optiontypes_ttms=types,
                  nb_path=int(rng.integers(300, 5000)), nb_steps_per_year=int(rng.integers(50, 400)), variable_type=vt,
                  seed=seed)
        if model == "logsv":
            kw.update(v0=rng.uniform(0.2, 1.0), theta=rng.uniform(0.2, 1.0), kappa1=rng.uniform(0.5, 4), kappa2=rng.uniform(0.5, 4),
                      beta=rng.uniform(-1, 0.5), volvol=rng.uniform(0.3, 1.8), vol_backbone_etas=rng.uniform(0.8, 1.2, m),
                      is_spot_measure=bool(rng.integers(0, 2)))
            fn, mod = sv.logsv_mc_chain_pricer, lp
        else:
            kw.update(v0=rng.uniform(0.02, 0.3), theta=rng.uniform(0.02, 0.3), kappa=rng.uniform(0.5, 5), rho=rng.uniform(-0.9, 0.5),
                      volvol=rng.uniform(0.2, 1.5), scheme=("euler", "qe")[int(rng.integers(0, 2))])
            fn, mod = sv.heston_mc_chain_pricer, hp
        a, ea = fn(**kw)
        mod.WHOLE_CHAIN_STEPPING = False
        try:
            b, eb = fn(**kw)
        finally:
            mod.WHOLE_CHAIN_STEPPING = True
        for x, y in zip(a + ea, b + eb):
            assert np.array_equal(x, y, equal_nan=True), (case, "whole-chain vs slice-by-slice")
        c = oracle_chain(model, kw, seed)
        for x, y in zip(a, c):
            ok = np.isfinite(y)
            err = np.max(np.abs(x[ok] - y[ok]) / (np.abs(y[ok]) + 1e-8), initial=0.0)
            worst = max(worst, err)
            if err > 0.0:
                k = int(np.argmax(np.where(ok, np.abs(x - y) / (np.abs(y) + 1e-8), 0.0)))
                ranking.append((err, case, model, vt.name, kw.get("is_spot_measure", True), kw.get("scheme", ""),
                                str(np.asarray(kw["optiontypes_ttms"][[id(p) for p in a].index(id(x))])[k]), float(y[k]),
                                float(x[k] - y[k])))
            assert err < 1e-8 and np.array_equal(np.isfinite(x), ok), (case, model, err, x, y)
        print(f"case {case:3d} {model:6s} m={m} n={kw['nb_path']:5d} {vt.name:10s} ok")
    for err, case, model, vname, spot, scheme, ty, price, diff in sorted(ranking, reverse=True)[:8]:
        print(f"  worst: {err:.2e} case {case} {model} {scheme} {vname} spot_measure={spot} type {ty} oracle price {price:.6e} gpu - oracle {diff:+.2e}")
    print(f"{n_cases} cases, worst relative deviation from the CPU oracle {worst:.2e}")
    return worst


if __name__ == "__main__":
    main()
