"""Socket power and clocks (rocm-smi, sampled every ~0.2 s by a background thread) while one kernel repeats for a few seconds:
the device-RNG vol-paths kernel, its supplied-brownians instantiation, the C2 stepping kernel and an idle baseline -- evidence for
(or against) DESIGN.md's reading that the shader clock under these kernels is a power equilibrium.  Prints one JSON per case;
fields are null where the box does not expose the sensor."""
import ctypes as C
import json
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from stochvolmodels_amd import _lib
from stochvolmodels_amd.engine import DeviceBuffer, get_engine

L = _lib.load()
n, nb = 1 << 20, 1024
eng = get_engine(n)
out = DeviceBuffer((nb + 1) * n)
import torch  # noqa: E402  (device plumbing: the scaled increments of the supplied-brownians case)
_wb = torch.randn(nb, n, dtype=torch.float64, device="cuda") * (1.0 / 360) ** 0.5
torch.cuda.synchronize()
wb = C.c_void_p(_wb.data_ptr())
L.svmc_clock_probe_arm(1)              # round 5: the in-kernel clock probe is per thread and off by default


def smi():
    try:
        txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(txt)
        card = next(iter(d.values()))
        power = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
        sclk = next((str(v) for k, v in card.items() if "sclk" in k.lower()), None)
        return power, sclk
    except Exception:
        return None, None


def case(name, launch, seconds=4.0):
    samples, stop = [], threading.Event()

    def poll():
        while not stop.wait(0.2):
            samples.append(smi())
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    t_end = time.perf_counter() + seconds
    k = 0
    while time.perf_counter() < t_end:
        for _ in range(20):
            launch()
        eng.synchronize()
        k += 20
    stop.set()
    th.join()
    st = (C.c_uint64 * 8)()
    L.svmc_clock_probe_read(st, None)
    mhz = [round(100.0 * (st[i + 2] - st[i]) / (st[i + 3] - st[i + 1]), 1) for i in (0, 4) if st[i + 3] > st[i + 1]]
    pw = [p for p, _ in samples[len(samples) // 2:] if p is not None]
    print(json.dumps({"case": name, "launches": k, "ms_per_launch": round(1e3 * seconds / max(k, 1), 3), "clock_mhz_in_kernel": mhz,
                      "socket_power_w_median": float(np.median(pw)) if pw else None, "sclk_samples": [s for _, s in samples[-3:]]}))


case("idle", lambda: time.sleep(0.01), 2.0)
case("vol paths, device RNG", lambda: L.svmc_logsv_vol_paths(out.ptr, n, n, nb, 1.0 / 360, 0.8, 1.0, 3.0, 3.0, 0.15, 1.8, 1, None, n, 5, 0, 0, None))
# (round 4 ran this case at volvol 0.01 on UNSCALED normals -- other dynamics than the timed 3.46 ms figure, which itself ran on
# NaN data; round 5: the same parameters as the device-RNG case on properly scaled increments sqrt(dt) N(0,1), `wb` below)
case("vol paths, supplied brownians", lambda: L.svmc_logsv_vol_paths(out.ptr, n, n, nb, 1.0 / 360, 0.8, 1.0, 3.0, 3.0, 0.15, 1.8, 1, wb, n, 5, 0, 0, None))
case("C2 stepping kernel", lambda: L.svmc_logsv_terminal_rng(eng.x.ptr, eng.vol.ptr, eng.qvar.ptr, n, nb, 1.0 / 1024, 1.0413, 3.1844, 3.058, 0.1514,
                                                                1.8458, 1.0, 1, 7, 0, 0, 0, None))
