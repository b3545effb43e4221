"""Stream-ordered collectives under load (run under torch.distributed.run with SVMC_DIST_BACKEND=gloo on one GPU, or
with RCCL on several): a chain whose stepping kernels take milliseconds is priced with the host synchronisations
around the all-reduces (SVMC_DIST_STRICT_SYNC=1) and without them; both must give the same bits on every rank, and
rank 0 prints them for comparison with a single-process run."""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402
from stochvolmodels_amd import dist as svdist  # noqa: E402

comm = svdist.init_from_env()
ttms = np.array([0.25, 0.5, 1.0])
kk = np.linspace(0.7, 1.3, 9)
ty = np.where(kk >= 1, "C", "P")
p = sv.LOGSV_BTC_PARAMS
kw = dict(ttms=ttms, forwards=np.ones(3), discfactors=np.ones(3), strikes_ttms=(kk,) * 3, optiontypes_ttms=(ty,) * 3,
          v0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta, volvol=p.volvol,
          vol_backbone_etas=np.ones(3), nb_path=1 << 21, nb_steps_per_year=512, seed=42)
out = {}
for mode in ("1", "0", "1", "0"):
    os.environ["SVMC_DIST_STRICT_SYNC"] = mode
    for _ in range(3):
        pr, sd = sv.logsv_mc_chain_pricer(**kw)
    out.setdefault(mode, []).append(np.concatenate(pr + sd))
same = all(np.array_equal(a, out["1"][0]) for v in out.values() for a in v)
if comm.rank == 0:
    print(json.dumps(dict(world=comm.world, identical_strict_vs_stream_ordered=bool(same), prices_head=[float(v) for v in out["0"][0][:4]])))
assert same
