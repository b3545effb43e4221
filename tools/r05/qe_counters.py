#!/usr/bin/env python
"""C3's Heston chains (2^22 paths x 4 x 128 steps, Euler and QE, both parameter sets) priced once each: run under
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace to read the DYNAMIC instruction count per wave-step of the stepping kernels
(tools/rocpd_summary.py prints the counters per kernel; instructions / (65536 waves x 512 steps))."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402

kk = np.linspace(0.6, 1.6, 21)
ty = np.where(kk >= 1.0, "C", "P")
ttms = np.array([0.25, 0.5, 0.75, 1.0])
chain = dict(ttms=ttms, forwards=np.ones(4), discfactors=np.ones(4), strikes_ttms=(kk,) * 4, optiontypes_ttms=(ty,) * 4)
hb = sv.BTC_HESTON_PARAMS
for par in (dict(v0=0.04, theta=0.04, kappa=4.0, rho=-0.5, volvol=0.4),
            dict(v0=hb.v0, theta=hb.theta, kappa=hb.kappa, rho=hb.rho, volvol=hb.volvol)):
    for scheme in ("euler", "qe"):
        pr, _ = sv.heston_mc_chain_pricer(nb_path=1 << 22, scheme=scheme, nb_steps_per_year=508, seed=20240603, **chain, **par)
        print(scheme, par["volvol"], float(pr[3][10]))
