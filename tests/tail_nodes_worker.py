"""worker of tests/test_gpu_parity.py::test_one_device_tail_equals_five_node_tail: prices a few chains through the fused C drivers
and prints every result's bytes; the test runs it twice, with and without SVMC_CHAIN_TAIL_NODES=5 (the round-5 tail: reduce, payoff,
reduce, [implied vols,] copy) and compares"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402


def main():
    p = sv.LOGSV_BTC_PARAMS
    out = {}
    k = np.array([0.7, 0.85, 1.0, 1.1, 1.3])
    for n in (4099, 100_000, 300_001):
        for tag, ttms, ty, vt in (("chain", np.array([0.1, 0.3, 0.55]), np.array(["P", "IP", "C", "IC", "C"]), sv.VariableType.LOG_RETURN),
                                  ("one", np.array([0.4]), np.array(["P", "P", "C", "C", "C"]), sv.VariableType.LOG_RETURN),
                                  ("qvar", np.array([0.2, 0.5]), np.array(["P", "P", "C", "C", "C"]), sv.VariableType.Q_VAR)):
            m = len(ttms)
            fw, df = 1.0 + 0.02 * np.arange(m), np.exp(-0.03 * ttms)
            kk = k * 0.8 if tag == "qvar" else k
            pr, sd = sv.logsv_mc_chain_pricer(ttms=ttms, forwards=fw, discfactors=df, strikes_ttms=(kk,) * m, optiontypes_ttms=(ty,) * m,
                                              v0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta, volvol=p.volvol,
                                              vol_backbone_etas=np.ones(m), nb_path=n, nb_steps_per_year=120, seed=5, variable_type=vt)
            out[f"logsv_{tag}_{n}"] = np.concatenate(pr + sd).tobytes().hex()
            for scheme in ("euler", "qe"):
                pr, sd = sv.heston_mc_chain_pricer(ttms=ttms, forwards=fw, discfactors=df, strikes_ttms=(kk,) * m, optiontypes_ttms=(ty,) * m,
                                                   v0=0.04, theta=0.05, kappa=3.0, rho=-0.6, volvol=0.5, nb_path=n, scheme=scheme,
                                                   nb_steps_per_year=120, seed=6, variable_type=vt)
                out[f"heston_{scheme}_{tag}_{n}"] = np.concatenate(pr + sd).tobytes().hex()
    # the calibration objective on frozen randoms: one set and five sets, prices + standard errors + implied vols
    ttms = np.array([1 / 12, 0.25, 0.5])
    k13 = np.linspace(0.75, 1.3, 9)
    ty = np.where(k13 >= 1.0, "C", "P")
    chain = dict(ttms=ttms, forwards=np.ones(3), discfactors=np.ones(3), strikes_ttms=(k13,) * 3, optiontypes_ttms=(ty,) * 3)
    res = sv.draw_fixed_randoms_on_device(ttms, nb_path=50_000, nb_steps_per_year=120, seed=10)
    sets = [sv.LogSvParams(sigma0=p.sigma0 + 1e-3 * j, theta=p.theta, kappa1=p.kappa1 + 1e-2 * j, kappa2=p.kappa2, beta=p.beta,
                           volvol=p.volvol - 1e-2 * j) for j in range(5)]
    one = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, v0=p.sigma0, theta=p.theta, kappa1=p.kappa1,
                                                 kappa2=p.kappa2, beta=p.beta, volvol=p.volvol, vol_backbone_etas=np.ones(3),
                                                 return_ivols=True, **chain)
    out["frozen_one"] = np.concatenate(one[0] + one[1] + one[2]).tobytes().hex()
    five = sv.logsv_mc_chain_pricer_fixed_randoms_batch(params_list=sets, W0s=res, return_ivols=True, **chain)
    out["frozen_five"] = np.concatenate([np.concatenate(r[0] + r[1] + r[2]) for r in five]).tobytes().hex()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
