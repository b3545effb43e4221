"""Overflow / collapse regimes (checker script, run by hand on a GPU box: python tests/stress_extremes.py): LogSV with volvol 6-8
over 6-8 years, terminal-state classes (finite / +inf / -inf / NaN) of the GPU generator against the CPU oracle on the same
stream.  Lives under tests/ because it drives the oracle (test infrastructure)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from stochvolmodels_amd.engine import get_engine
n = 20000
eng = get_engine(n)
for (volvol, beta, nb, dt, tag) in ((8.0, 2.0, 400, 0.02, "explosive"), (6.0, -3.0, 300, 0.02, "collapse"), (1.8, 0.15, 64, 1/360, "normal")):
    for spot in (True, False):
        eng.fill_state(0.0, 0.8, 0.0)
        eng.logsv_rng(nb, dt, 1.0, 3.0, 3.0, beta, volvol, 1.0, spot, 5, 0, 0)
        x, s, q = eng.get_state()
        ox, os_, oq = oracle.logsv_terminal_rng(np.zeros(n), np.full(n, 0.8), np.zeros(n), nb, dt, 1.0, 3.0, 3.0, beta, volvol, 5, is_spot_measure=spot)
        def cls(a): return np.where(np.isnan(a), 2, np.where(np.isinf(a), np.sign(a), 0))
        same = [int(np.sum(cls(a) != cls(b))) for a, b in ((x, ox), (s, os_), (q, oq))]
        fin = np.isfinite(ox) & np.isfinite(x) & np.isfinite(oq) & np.isfinite(q)
        rel = np.max(np.abs(x[fin] - ox[fin]) / (1 + np.abs(ox[fin]))) if fin.any() else 0
        relq = np.max(np.abs(q[fin] - oq[fin]) / (1e-300 + np.abs(oq[fin]))) if fin.any() else 0
        print(tag, "spot" if spot else "inv", "class mismatches (x, sigma, qvar):", same, "nonfinite x (gpu, cpu):", int((~np.isfinite(x)).sum()), int((~np.isfinite(ox)).sum()), "max rel dev x %.2e q %.2e" % (rel, relq))
