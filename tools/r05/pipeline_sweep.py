#!/usr/bin/env python
"""simulate_vol_paths(2^20 x 1024) -> a fresh NumPy array (8.6 GB), wall time by pipeline shape: host threads that drain the pinned
ring x ring slots x chunk size (SVMC_PIPELINE_*), one process per shape.  One JSON line per shape."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = """
import sys, time, json; sys.path.insert(0, %r)
import stochvolmodels_amd as sv
from stochvolmodels_amd import engine
pr = sv.LogSVPricer(); p = sv.LOGSV_BTC_PARAMS
pr.simulate_vol_paths(p, ttm=1.0, nb_path=4096, nb_steps=1023, seed=1)
ts = []
for rep in range(3):
    t0 = time.perf_counter(); a, _ = pr.simulate_vol_paths(p, ttm=1.0, nb_path=1 << 20, nb_steps=1023, seed=5); ts.append(time.perf_counter() - t0); del a
print(json.dumps({"threads": engine.PIPELINE_THREADS, "slots": engine.PIPELINE_SLOTS, "chunk_MiB": engine.PIPELINE_CHUNK_BYTES >> 20,
                  "seconds": [round(t, 4) for t in ts], "GBps_best": round(8.5983232 / min(ts), 1)}))
""" % ROOT
for threads, slots, chunk in ((2, 4, 32), (4, 6, 32), (6, 8, 32), (8, 10, 32), (12, 14, 32), (16, 18, 32), (8, 10, 16), (8, 10, 64)):
    env = dict(os.environ, SVMC_PIPELINE_THREADS=str(threads), SVMC_PIPELINE_SLOTS=str(slots), SVMC_PIPELINE_CHUNK_MIB=str(chunk))
    run = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(([ln for ln in run.stdout.splitlines() if ln.startswith("{")] or [json.dumps({"threads": threads, "error": run.stderr[-200:]})])[-1], flush=True)
