import sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np
import stochvolmodels_amd as sv
p = sv.LOGSV_BTC_PARAMS
pr = sv.LogSVPricer()
pr.simulate_vol_paths(p, ttm=1.0, nb_path=4096, nb_steps=1023, seed=1)
for rep in range(2):
    t = [time.perf_counter()]
    dev, _ = pr.simulate_vol_paths(p, ttm=1.0, nb_path=1 << 20, nb_steps=1023, seed=5, return_device=True); dev.synchronize(); t.append(time.perf_counter())
    m = dev.row_moments(center=p.theta, n_moments=4); t.append(time.perf_counter())
    q = dev.expanding_mean_of_squares(); q.synchronize(); t.append(time.perf_counter())
    m2 = q.row_moments(center=0.0, n_moments=1); t.append(time.perf_counter())
    q.free(); t.append(time.perf_counter())
    dev.free(); t.append(time.perf_counter())
    print(json.dumps(dict(zip(["simulate", "row_moments4", "expanding", "row_moments1", "free_q", "free_dev"], [round(1e3*(b-a), 3) for a, b in zip(t, t[1:])]))))
