#!/usr/bin/env python
"""
Round 6, random stream version 4 (the lattice point of a word is the signed integer itself): the first eight raw moments of the
DEVICE's normals (svmc_fill_normals: stream 0, both normals of 1024 steps of 2^20 paths, several seeds / call ids -- 2^31 draws
per (seed, call)) against N(0,1)'s 0, 1, 0, 3, 0, 15, 0, 105, each with its z-score (the sampling error of E[z^k] is
sqrt((E[z^2k] - E[z^k]^2) / N)), summed on the device (svmc_row_power_sums) so that 8 numbers per row cross PCIe.

    python tools/r06/stream4_moments.py [n_seeds]          one text block
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402
from stochvolmodels_amd import _lib  # noqa: E402
from stochvolmodels_amd.engine import DeviceBuffer, get_engine  # noqa: E402

DOUBLE_FACT = {2: 1.0, 4: 3.0, 6: 15.0, 8: 105.0, 10: 945.0, 12: 10395.0, 14: 135135.0, 16: 2027025.0}


def gauss_moment(k):
    return 0.0 if k % 2 else DOUBLE_FACT[k]


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n, nb = 1 << 20, 1024
    lib = _lib.load()
    eng = get_engine(n)
    sums, ws = DeviceBuffer(nb * 8), DeviceBuffer(nb * 4 * 8)
    tot = np.zeros(8, dtype=np.longdouble)
    count = 0
    for s in range(n_seeds):
        seed, call_id = 20241001 + 7919 * s, s
        w0, w1 = eng.fill_normals(nb, seed, call_id)
        for ptr in (w0, w1):
            _lib.check(lib.svmc_row_power_sums(ptr, n, nb, n, 0.0, 4, sums.ptr, ws.ptr, ws.nbytes, eng.stream))
            host = np.empty(nb * 8)
            _lib.check(lib.svmc_memcpy_d2h(host.ctypes.data, sums.ptr, host.nbytes, eng.stream))
            eng.synchronize()
            tot += host.reshape(8, nb).astype(np.longdouble).sum(axis=1)
            count += n * nb
    print(f"# random stream version {sv.RNG_STREAM_VERSION}: raw moments of {count} device normals ({n_seeds} seeds x 2 normals x {nb} steps x "
          f"{n} paths), summed on the device; z = (sample - N(0,1)'s) / sampling error")
    worst = 0.0
    for k in range(1, 9):
        m = float(tot[k - 1] / count)
        want = gauss_moment(k)
        se = np.sqrt((gauss_moment(2 * k) - want * want) / count)
        z = (m - want) / se
        worst = max(worst, abs(z))
        print(f"E[z^{k}] = {m:+.9f}   N(0,1): {want:g}   z = {z:+.2f}")
    print(f"worst |z| over the eight moments: {worst:.2f}")
    sums.free()
    ws.free()
    return 0 if worst < 4.5 else 1


if __name__ == "__main__":
    sys.exit(main())
