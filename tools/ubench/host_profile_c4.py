import cProfile, pstats, io, sys, time
sys.path.insert(0, '/root/repo')
import bench
import stochvolmodels_amd as sv
wl = bench.make_workload("c4", sv)
P = sv.LOGSV_BTC_PARAMS
n = 1 << 21
for i in range(30): bench.price(sv, wl, P, n, 100 + i)
t0 = time.perf_counter()
for i in range(50): bench.price(sv, wl, P, n, 200 + i)
print("ms/call", (time.perf_counter() - t0) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable()
for i in range(100): bench.price(sv, wl, P, n, 300 + i)
pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
