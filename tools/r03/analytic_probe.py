import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import stochvolmodels_amd as sv
ttms = np.array([0.25, 0.5, 0.75, 1.0]); one = np.ones(4)
kk = np.linspace(0.6, 1.4, 21)
types = np.array(["P"] * 10 + ["C"] * 11)
chain = sv.OptionChain(ttms=ttms, forwards=one, strikes_ttms=(kk,) * 4, optiontypes_ttms=(types,) * 4, ids=None)
pricer = sv.LogSVPricer()
P = sv.LOGSV_BTC_PARAMS
for _ in range(5): pricer.price_chain(chain, P)
t = time.perf_counter()
for _ in range(20): pricer.price_chain(chain, P)
print("ms per chain", (time.perf_counter() - t) / 20 * 1e3)
