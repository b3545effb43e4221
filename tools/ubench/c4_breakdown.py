import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import stochvolmodels_amd as sv
from stochvolmodels_amd.engine import get_engine
P = sv.LOGSV_BTC_PARAMS
ttms8 = np.arange(1, 9) / 8.0
fw = 67000.0 * np.exp(0.05 * ttms8)
strikes8 = tuple(f * np.linspace(0.6, 1.6, 21) for f in fw)
types8 = tuple(np.where(k >= f, "C", "P") for k, f in zip(strikes8, fw))
chain8 = sv.OptionChain(ttms=ttms8, forwards=fw, strikes_ttms=strikes8, optiontypes_ttms=types8, ids=None)
n = 1 << 21
pr = sv.LogSVPricer()
f = lambda: pr.model_mc_price_chain(chain8, P, nb_path=n, nb_steps=1016, seed=4)
f(); f()
eng = get_engine(n)
eng.start_kernel_timing()
t0 = time.perf_counter()
for _ in range(3): f()
dt = (time.perf_counter() - t0) / 3
k = eng.stop_kernel_timing()["logsv_chain_rng_kernel"]
# one long kernel of the same total work
eng.fill_state(0.0, P.sigma0, 0.0)
eng.logsv_rng(1024, 1/1024, P.theta, P.kappa1, P.kappa2, P.beta, P.volvol, 1.0, True, 4, 0, 0)
eng.start_kernel_timing()
for _ in range(3):
    eng.fill_state(0.0, P.sigma0, 0.0)
    eng.logsv_rng(1024, 1/1024, P.theta, P.kappa1, P.kappa2, P.beta, P.volvol, 1.0, True, 4, 0, 0)
k1 = eng.stop_kernel_timing()["logsv_rng_kernel"]
print(json.dumps(dict(call_ms=1e3*dt, stepping_ms=float(np.sum(k))/3, chain_kernel_ms=[round(float(v),3) for v in k], one_long_kernel_ms=float(np.mean(k1)))))
