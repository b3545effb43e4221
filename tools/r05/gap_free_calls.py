#!/usr/bin/env python
"""Would an enqueue / finish split of the chain driver pay?  C2's device work (stepping launch with its spot-sums reduce, payoff
launch, column reduce) queued K times back to back through the engine's asynchronous entry points with ONE synchronisation at the
end, against the same work with a synchronisation after every chain (what a synchronous pricing call does), HIP events around
both; and the in-kernel clock of the last stepping launch in each mode.  One JSON line."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402
from stochvolmodels_amd import _lib  # noqa: E402
from stochvolmodels_amd.engine import get_engine, option_type_codes, payoff_shifts  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    L = _lib.load()
    P = sv.LOGSV_BTC_PARAMS
    n, nb = 1 << 20, 1024
    kk = np.linspace(0.6, 1.6, 21)
    codes = option_type_codes(np.where(kk >= 1.0, "C", "P"))
    shifts = payoff_shifts(kk, codes, 1.0, 1)
    eng = get_engine(n)
    eng.reserve_snapshots(1)
    spot = eng.sums_buffer("spot", 2) if hasattr(eng, "sums_buffer") else None
    from stochvolmodels_amd.dist import get_default_comm
    comm = get_default_comm()
    spot_ptr, _ = comm.alloc(eng, 2, "spot")
    sums_ptr, _ = comm.alloc(eng, 3 * 21 + 1, "payoff")

    def chain(i):
        eng.logsv_slice_rng(nb, 1.0 / nb, P.theta, P.kappa1, P.kappa2, P.beta, P.volvol, 1.0, True, 20240602, i, 0, 1.0, 0, None, spot_ptr,
                            start=(0.0, P.sigma0, 0.0))
        eng.payoff_sums_chain([0], None, [1.0], [1.0], spot_ptr, [kk], [codes], [shifts], 1, sums_ptr)

    def sync():
        _lib.check(L.svmc_stream_synchronize(None))

    def timed(per_chain_sync):
        for i in range(10):
            chain(i)
        sync()
        e0, e1 = C.c_void_p(), C.c_void_p()
        L.svmc_event_create(C.byref(e0)); L.svmc_event_create(C.byref(e1))
        L.svmc_event_record(e0, None)
        for i in range(K):
            chain(100 + i)
            if per_chain_sync:
                sync()
        L.svmc_event_record(e1, None)
        sync()
        ms = C.c_float()
        L.svmc_event_elapsed_ms(e0, e1, C.byref(ms))
        # the clock inside one more stepping launch of the same pattern
        _lib.check(L.svmc_clock_probe_arm(1))
        chain(999)
        sync()
        st = (C.c_uint64 * 8)()
        _lib.check(L.svmc_clock_probe_read(st, None))
        _lib.check(L.svmc_clock_probe_arm(0))
        mhz = 100.0 * (st[2] - st[0]) / (st[3] - st[1]) if st[3] > st[1] else None
        return ms.value / K, mhz

    a, ca = timed(True)
    b, cb = timed(False)
    a2, _ = timed(True)
    print(json.dumps({"chains": K, "sync_after_every_chain_ms": a, "again_ms": a2, "queued_back_to_back_ms": b,
                      "clock_mhz_after_synced": ca, "clock_mhz_after_queued": cb,
                      "psps_synced": (1 << 30) / (a * 1e-3), "psps_queued": (1 << 30) / (b * 1e-3)}))


if __name__ == "__main__":
    main()
