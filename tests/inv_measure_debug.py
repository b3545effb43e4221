"""Checker script (run by hand on a GPU box): the C4 rank share in the inverse measure, GPU vs the CPU oracle, per expiry -- the
round-3 investigation of the additive recentring under the BTC set.  Lives under tests/ because it drives
the oracle (test infrastructure)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stochvolmodels_amd as sv
from stochvolmodels_amd.engine import get_engine
from oracle import oracle as cpu
cpu.set_threads(cpu.effective_cores())
p = sv.LOGSV_BTC_PARAMS
n, seed, spy = 1 << 21, 20240607, 1016
ttms = np.arange(1, 9) / 8.0
fw = 67000.0 * np.exp(0.05 * ttms); dfs = np.exp(-0.05 * ttms)
strikes = tuple(f * np.linspace(0.6, 1.6, 21) for f in fw)
types = tuple(np.where(k >= f, "IC", "IP") for k, f in zip(strikes, fw))
pr, sd = sv.logsv_mc_chain_pricer(ttms=ttms, forwards=fw, discfactors=dfs, strikes_ttms=strikes, optiontypes_ttms=types, v0=p.sigma0, theta=p.theta, kappa1=p.kappa1,
                                  kappa2=p.kappa2, beta=p.beta, volvol=p.volvol, vol_backbone_etas=np.ones(8), is_spot_measure=False, nb_path=n, nb_steps_per_year=spy, seed=seed)
eng = get_engine(n)
gx, gs, gq = eng.get_state()
x, s, q = np.zeros(n), p.sigma0 * np.ones(n), np.zeros(n)
t0, step0 = 0.0, 0
for i, ttm in enumerate(ttms):
    nb, dt, _ = sv.set_time_grid(ttm - t0, spy)
    x, s, q = cpu.logsv_terminal_rng(x, s, q, nb, dt, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, seed, is_spot_measure=False, step_offset=step0)
    a, b = cpu.payoff(x, q, float(ttm), float(fw[i]), strikes[i], types[i], float(dfs[i]))
    rel = np.abs(pr[i] - a) / np.abs(a)
    print(i, "max rel price dev", rel.max(), "at strike", rel.argmax(), "price", a[rel.argmax()], "stderr", b[rel.argmax()], "x range", x.min(), x.max(),
          "n nonfinite", int((~np.isfinite(x)).sum()))
    t0, step0 = ttm, step0 + nb
# last expiry: payoff of the oracle on the GPU's terminal x
a2, b2 = cpu.payoff(gx, gq, 1.0, float(fw[7]), strikes[7], types[7], float(dfs[7]))
print("oracle payoff on GPU states vs GPU prices:", np.max(np.abs(a2 - pr[7]) / np.abs(a2)), " vs oracle on CPU states:", np.max(np.abs(a2 - a) / np.abs(a)))
d = np.abs(gx - x)
k = np.argsort(d)[-5:]
print("largest |dx|:", d[k], "x there", x[k], "S there", fw[7] * np.exp(x[k]))
m = np.argsort(x)[:5]
print("smallest x:", x[m], gx[m], "dx", gx[m] - x[m])
# payoff contributions of the smallest-S paths, put side
S = fw[7] * np.exp(x) * fw[7] / np.nanmean(fw[7] * np.exp(x))
K = strikes[7][0]
pay = np.maximum(K - S, 0) / S
print("IP payoff top5:", np.sort(pay)[-5:], "sum", pay.sum(), "mean", pay.mean())
