import cProfile, pstats, io, os, sys
sys.path.insert(0, os.getcwd())
import bench, stochvolmodels_amd as sv
P = sv.LOGSV_BTC_PARAMS
wl = bench.make_workload(sys.argv[1] if len(sys.argv) > 1 else "c2", sv)
n = 1 << (20 if wl["name"] == "c2" else 21)
for i in range(20): bench.price(sv, wl, P, n, i)
pr = cProfile.Profile(); pr.enable()
for i in range(300): bench.price(sv, wl, P, n, 100 + i)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:6000])
