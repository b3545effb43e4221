#!/usr/bin/env python
"""
Config C5 (BASELINE.json: LogSV analytic-vs-MC sweep over 5 parameter sets), measured and REPORTED, not hidden in a
tolerance:  python tools/c5_bias.py [log2_paths] > profiles/r02_c5_bias.json      (GPU box)

For each of the five parameter sets of SURVEY.md 8d on the chain ttms = [0.25, 0.5, 0.75, 1], 21 strikes per expiry:
  * the GPU's analytic prices (Fourier inversion of the affine expansion, first and second order) next to the reference's
    own analytic prices for the same chain (tests/golden/analytic.npz, second order, SciPy RK45 as shipped),
  * the GPU Monte Carlo prices at 2^23 paths (spy = 508: 128 steps per expiry) with their standard errors,
  * (MC - analytic) in units of the MC standard error and relative to the price: at this path count the standard error
    is far below the truncation error of the expansion, so the difference IS the expansion's bias (plus the time
    discretisation of the log-Euler scheme) -- an error of the reference's analytic approximation, which the reference's own
    acceptance test (tests/test_logsv_characterization.py:407, 40 000 paths) cannot see and tests/test_gpu_parity.py
    test_c5_reference_criterion_at_reference_scale repeats verbatim.
Also the same comparison for quadratic-variance calls (40 000-point psi grid) on the two sets of analytic_qvar.npz.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import stochvolmodels_amd as sv  # noqa: E402
from stochvolmodels_amd.pricers.logsv.affine_expansion import ExpansionOrder  # noqa: E402


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
    n = 1 << log2n
    g = np.load(os.path.join(ROOT, "tests", "golden", "analytic.npz"))
    kk, types, ttms = g["strikes"], g["types"], g["ttms"]
    one = np.ones(len(ttms))
    chain = sv.OptionChain(ttms=ttms, forwards=one, strikes_ttms=(kk,) * len(ttms), optiontypes_ttms=(types,) * len(ttms),
                           ids=None)
    pricer = sv.LogSVPricer()
    # the analytic side of all five sets in one batch of launches (timed; the per-set prices below are the same bits)
    all_sets = []
    for tag in ("btc", "readme", "quick", "test", "fig3"):
        v = [float(a) for a in g[f"logsv_{tag}_params"]]
        all_sets.append(sv.LogSvParams(sigma0=v[0], theta=v[1], kappa1=v[2], kappa2=v[3], beta=v[4], volvol=v[5]))
    pricer.price_chain_batch(chain, all_sets)                     # warm-up (library load, first launches)
    t0 = time.perf_counter()
    batch = pricer.price_chain_batch(chain, all_sets)
    t_batch = time.perf_counter() - t0
    t0 = time.perf_counter()
    for p_ in all_sets:
        pricer.price_chain(chain, p_)
    t_loop = time.perf_counter() - t0
    out = {"analytic_five_sets_batched_ms": 1e3 * t_batch, "analytic_five_sets_one_by_one_ms": 1e3 * t_loop, "paths": n, "steps_per_year": 508, "ttms": ttms.tolist(), "strikes": kk.tolist(), "types": types.tolist(),
           "note": "z = (MC - analytic second order) / MC stderr; rel = (MC - analytic) / analytic; per expiry: the "
                   "largest |z| and |rel| over the 21 strikes (rel over strikes whose price exceeds 1e-4)", "sets": {}}
    for i, tag in enumerate(("btc", "readme", "quick", "test", "fig3")):
        v = [float(a) for a in g[f"logsv_{tag}_params"]]
        params = sv.LogSvParams(sigma0=v[0], theta=v[1], kappa1=v[2], kappa2=v[3], beta=v[4], volvol=v[5])
        t0 = time.perf_counter()
        a2 = np.stack(pricer.price_chain(chain, params))
        a1 = np.stack(pricer.price_chain(chain, params, expansion_order=ExpansionOrder.FIRST))
        t_an = time.perf_counter() - t0
        t0 = time.perf_counter()
        mc, sd = pricer.model_mc_price_chain(chain, params, nb_path=n, nb_steps=508, seed=100 + i)
        t_mc = time.perf_counter() - t0
        mc, sd = np.stack(mc), np.stack(sd)
        ref = g[f"logsv_{tag}_prices"]
        big = a2 > 1e-4
        row = {"params": dict(zip(("sigma0", "theta", "kappa1", "kappa2", "beta", "volvol"), v)),
               "analytic_vs_reference_max_abs": float(np.max(np.abs(a2 - ref))),
               "max_abs_z_second_order": [float(x) for x in np.max(np.abs(mc - a2) / sd, axis=1)],
               "max_abs_z_first_order": [float(x) for x in np.max(np.abs(mc - a1) / sd, axis=1)],
               "max_rel_bias_second_order": [float(np.max(np.abs((mc - a2) / a2)[j][big[j]])) for j in range(len(ttms))],
               "max_rel_bias_first_order": [float(np.max(np.abs((mc - a1) / a1)[j][big[j]])) for j in range(len(ttms))],
               "max_rel_stderr": [float(np.max((sd / np.abs(mc))[j][big[j]])) for j in range(len(ttms))],
               "seconds_analytic_both_orders": t_an, "seconds_mc": t_mc,
               "mc": mc.tolist(), "stderr": sd.tolist(), "analytic_second_order": a2.tolist(),
               "analytic_first_order": a1.tolist(), "reference_analytic_second_order": ref.tolist()}
        out["sets"][tag] = row
        print(tag, "z2", np.round(row["max_abs_z_second_order"], 1), "rel2", np.round(row["max_rel_bias_second_order"], 4),
              "rel1", np.round(row["max_rel_bias_first_order"], 4), file=sys.stderr)
    # quadratic-variance calls
    gq = np.load(os.path.join(ROOT, "tests", "golden", "analytic_qvar.npz"))
    out["qvar"] = {}
    for tag in ("test", "btc"):
        v = [float(a) for a in gq[f"{tag}_params"]]
        params = sv.LogSvParams(sigma0=v[0], theta=v[1], kappa1=v[2], kappa2=v[3], beta=v[4], volvol=v[5])
        kq = gq[f"{tag}_strikes"]
        chq = sv.OptionChain(ttms=gq["ttms"], forwards=gq["forwards"], strikes_ttms=(kq, kq),
                             optiontypes_ttms=(np.array(["C"] * len(kq)),) * 2, ids=None, discfactors=gq["discfactors"])
        an = np.stack(pricer.price_chain(chq, params, variable_type=sv.VariableType.Q_VAR))
        mc, sd = pricer.model_mc_price_chain(chq, params, variable_type=sv.VariableType.Q_VAR, nb_path=n, nb_steps=720, seed=8)
        mc, sd = np.stack(mc), np.stack(sd)
        out["qvar"][tag] = {"strikes": kq.tolist(), "ttms": gq["ttms"].tolist(), "analytic": an.tolist(), "mc": mc.tolist(),
                            "stderr": sd.tolist(), "reference_analytic": gq[f"{tag}_prices"].tolist(),
                            "max_abs_z": [float(x) for x in np.max(np.abs(mc - an) / sd, axis=1)],
                            "rel_bias": ((mc - an) / an).tolist()}
        print("qvar", tag, "z", np.round(out["qvar"][tag]["max_abs_z"], 1), "rel", np.round((mc - an) / an, 3).tolist(),
              file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
