#!/usr/bin/env python
"""
Rough-LogSV kernel timing on one MI355X: 2^20 paths x 360 steps, N = 1, 2, 3 factors, with the
normals drawn in the kernel and streamed from HBM.  One JSON object per line.

    python tools/bench_rough.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from stochvolmodels_amd.engine import get_engine  # noqa: E402

RULES = {1: ([1e-3], [1.0]), 2: ([0.00181, 1.524], [0.9, 1.7]), 3: ([0.0772, 5.19, 108.46], [0.777, 1.554, 8.516])}


def main():
    n, nb = 1 << 20, 360
    eng = get_engine(n)
    h = 1.0 / nb
    for nf, (nodes, weights) in RULES.items():
        nodes, weights = np.array(nodes), np.array(weights)
        v0 = np.full(nf, 0.377 / weights.sum())
        args = (nb, h, nodes, weights, v0, 0.347, 1.29, 1.93, 2.45 / np.hypot(2.45, 1.81), float(np.hypot(2.45, 1.81)))
        z0, z1 = eng.fill_normals(nb, 11)
        for mode, kw in (("device_rng", dict(seed=5)), ("streamed", dict(z0_ptr=z0, z1_ptr=z1))):
            for _ in range(2):
                eng.rough_logsv(*args, **kw)
            eng.start_kernel_timing()
            for _ in range(5):
                eng.rough_logsv(*args, **kw)
            ms = np.mean(eng.stop_kernel_timing()["rough_logsv_kernel"])
            x, _, y = eng.get_state()
            print(json.dumps(dict(kernel="rough_logsv_kernel", n_factors=nf, mode=mode, paths=n, steps=nb, ms=float(ms),
                                  path_steps_per_s=n * nb / (ms * 1e-3), mean_spot=float(np.exp(x).mean()),
                                  mean_qvar=float(y.mean()))))


if __name__ == "__main__":
    main()
