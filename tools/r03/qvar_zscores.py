"""diagnostic: z-scores (MC - analytic) / stderr of the BTC-set quadratic-variance calls of tests/test_gpu_parity.py
test_analytic_qvar for a range of seeds, and at 2^22 paths -- is a 4.7-sigma reading at one seed the draw or a bias?"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import stochvolmodels_amd as sv

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "analytic_qvar.npz"))
v = [float(a) for a in g["btc_params"]]
params = sv.LogSvParams(sigma0=v[0], theta=v[1], kappa1=v[2], kappa2=v[3], beta=v[4], volvol=v[5])
kk = g["btc_strikes"]
chain = sv.OptionChain(ttms=g["ttms"], forwards=g["forwards"], strikes_ttms=(kk, kk), optiontypes_ttms=(np.array(["C"] * 8),) * 2,
                       ids=None, discfactors=g["discfactors"])
pricer = sv.LogSVPricer()
an = np.stack(pricer.price_chain(chain, params, variable_type=sv.VariableType.Q_VAR))
out = {"lib": os.environ.get("SVMC_LIB", "in-tree"), "analytic": an.tolist(), "z_40000": [], "seeds": list(range(1, 25))}
for seed in out["seeds"]:
    mc, sd = pricer.model_mc_price_chain(chain, params, variable_type=sv.VariableType.Q_VAR, nb_path=40_000, nb_steps=720, seed=seed)
    out["z_40000"].append(np.round((np.stack(mc) - an) / np.stack(sd), 2)[1].tolist())
z = np.array(out["z_40000"])
out["z_40000_mean_over_seeds"] = np.round(z.mean(axis=0), 2).tolist()
out["z_40000_max_abs"] = float(np.abs(z).max())
mc, sd = pricer.model_mc_price_chain(chain, params, variable_type=sv.VariableType.Q_VAR, nb_path=1 << 22, nb_steps=720, seed=1)
out["big_2p22"] = {"mc": np.stack(mc).tolist(), "stderr": np.stack(sd).tolist(), "z": np.round((np.stack(mc) - an) / np.stack(sd), 2).tolist(),
                   "rel_bias": np.round((np.stack(mc) - an) / an, 4).tolist()}
print(json.dumps(out))
