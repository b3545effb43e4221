#!/usr/bin/env python
"""
Round 6, random stream version 4: distributional checks of the DEVICE's normals beyond their moments (tools/r06/stream4_moments.py),
on the arrays svmc_fill_normals leaves in HBM, reduced there with torch (zero-copy view through __cuda_array_interface__):
  * chi-square of the normals in 1024 EQUIPROBABLE bins of N(0,1) (bin = floor(1024 Phi(z)): df 1023) -- the inverse-CDF table as
    the device evaluates it, over the whole range;
  * counts beyond 3, 4, 5 and 6 sigma against their expectations (the tails, where the segments are geometric);
  * lag-1 autocorrelation along the step axis, the correlation between a step's two normals (w0, w1), and between neighbouring
    paths -- each ~ N(0, 1/N) for independent draws.
    python tools/r06/stream4_normals_battery.py [n_seeds]
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402
from stochvolmodels_amd.engine import DeviceArray, DeviceBuffer, get_engine  # noqa: E402


class View:
    """a borrowed [rows][cols] float64 view of device memory for torch.as_tensor"""
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f8", "data": (int(ptr), False), "version": 3, "strides": None}


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n, nb, bins = 1 << 20, 512, 1024
    eng = get_engine(n)
    hist = torch.zeros(bins, dtype=torch.float64, device="cuda")
    tails = {3: 0.0, 4: 0.0, 5: 0.0, 6: 0.0}
    s_lag = s_cross = s_path = 0.0
    n_lag = n_cross = n_path = 0
    count = 0
    for s in range(n_seeds):
        w0p, w1p = eng.fill_normals(nb, 20241001 + 104729 * s, s)
        eng.synchronize()
        w0 = torch.as_tensor(View(w0p, (nb, n)), device="cuda")
        w1 = torch.as_tensor(View(w1p, (nb, n)), device="cuda")
        for w in (w0, w1):
            for r0 in range(0, nb, 64):                                   # in slabs: the temporaries stay small
                z = w[r0:r0 + 64]
                u = 0.5 * torch.erfc(-z / math.sqrt(2.0))
                b = torch.clamp((u * bins).to(torch.int64), 0, bins - 1)
                hist += torch.bincount(b.reshape(-1), minlength=bins).to(torch.float64)
                a = z.abs()
                for k in tails:
                    tails[k] += float((a > k).sum())
            count += w.numel()
            s_lag += float((w[:-1] * w[1:]).sum())
            n_lag += (nb - 1) * n
            s_path += float((w[:, :-1] * w[:, 1:]).sum())
            n_path += nb * (n - 1)
        s_cross += float((w0 * w1).sum())
        n_cross += nb * n
        torch.cuda.synchronize()
    h = hist.cpu().numpy()
    exp = count / bins
    chi2 = float(np.sum((h - exp) ** 2 / exp))
    df = bins - 1
    z_chi = (chi2 - df) / math.sqrt(2.0 * df)
    print(f"# random stream version {sv.RNG_STREAM_VERSION}: {count} device normals ({n_seeds} seeds x 2 normals x {nb} steps x {n} paths)")
    print(f"chi-square over {bins} equiprobable bins: {chi2:.1f} (df {df}; z = {z_chi:+.2f})")
    worst = abs(z_chi)
    for k, got in tails.items():
        p = math.erfc(k / math.sqrt(2.0))
        want = count * p
        z = (got - want) / math.sqrt(want)
        worst = max(worst, abs(z))
        print(f"|z| > {k}: {int(got)} draws, expected {want:.1f}   z = {z:+.2f}")
    for tag, sm, nn in (("lag-1 autocorrelation along the steps", s_lag, n_lag), ("correlation of a step's two normals", s_cross, n_cross),
                        ("correlation of neighbouring paths", s_path, n_path)):
        r = sm / nn
        z = r * math.sqrt(nn)
        worst = max(worst, abs(z))
        print(f"{tag}: {r:+.3e}   z = {z:+.2f}")
    print(f"worst |z|: {worst:.2f}")
    return 0 if worst < 4.5 else 1


if __name__ == "__main__":
    sys.exit(main())
