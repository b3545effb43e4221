"""Is the one-time ~60 ms step of a chain-pricing process (profiles/r02_runtime_stall.txt) CPython's generational
garbage collector?  Time 400 C2 chain calls one by one with a gc callback that records every collection (generation,
duration, objects tracked); then the same with gc.freeze() after the warm-up.  Run on the GPU box."""
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import bench
import stochvolmodels_amd as sv

P = sv.LOGSV_BTC_PARAMS
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
wl = bench.make_workload(cfg, sv)
n = 1 << (20 if cfg == "c2" else 21)
events, t_start = [], [0.0]


def cb(phase, info):
    if phase == "start":
        t_start[0] = time.perf_counter()
    else:
        events.append((info["generation"], 1e3 * (time.perf_counter() - t_start[0]), info["collected"], len(gc.get_objects())))


gc.callbacks.append(cb)
out = {"config": cfg, "gc_threshold": gc.get_threshold(), "objects_tracked_at_start": len(gc.get_objects())}
for label in ("plain", "after gc.collect + gc.freeze"):
    if label != "plain":
        gc.collect()
        gc.freeze()
    events.clear()
    steps, marks = [], []
    for i in range(400):
        t0 = time.perf_counter()
        n_ev = len(events)
        bench.price(sv, wl, P, n, 1000 + i)
        steps.append(1e3 * (time.perf_counter() - t0))
        if len(events) > n_ev:
            marks.append((i, [(g, round(ms, 3)) for g, ms, _, _ in events[n_ev:]]))
    steps = np.array(steps)
    slow = [int(i) for i in np.argsort(steps)[::-1][:4]]
    out[label] = {"median_ms": round(float(np.median(steps)), 3), "slowest": [[i, round(float(steps[i]), 3)] for i in slow],
                  "gen2_collections": [(i, ev) for i, ev in marks if any(g == 2 for g, _ in ev)],
                  "gen1_collections": sum(1 for _, ev in marks for g, _ in ev if g == 1),
                  "gen0_collections": sum(1 for _, ev in marks for g, _ in ev if g == 0),
                  "max_gen0_ms": max([ms for _, ev in marks for g, ms in ev if g == 0] or [0.0]),
                  "max_gen1_ms": max([ms for _, ev in marks for g, ms in ev if g == 1] or [0.0])}
print(json.dumps(out))
