"""worker of tests/test_dist_gloo.py: one rank of a world_size-N gloo group pricing a LogSV and a Heston chain
through the PRODUCT chain drivers with the engine replaced by the CPU test double."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(out_path):
    import torch.distributed as dist

    from fake_engine import FakeEngine
    from stochvolmodels_amd import dist as svdist
    from stochvolmodels_amd.pricers import heston_pricer, logsv_pricer
    from stochvolmodels_amd.utils.config import VariableType

    phases = []
    ladder = None
    if os.environ.get("SVMC_TEST_LADDER") == "1":
        # the fallback ladder on a box without a GPU: the nccl and rccl rungs fail (in their probe children when the probe is
        # forced, else at the device count) and the job runs on the gloo control plane
        comm, ladder = svdist.init_with_fallback(on_phase=phases.append,
                                                 probe_timeout=float(os.environ.get("SVMC_TEST_PROBE_TIMEOUT", "60")))
    else:
        comm = svdist.init_from_env(backend="gloo", on_phase=phases.append)
    engines = {}

    def fake_get_engine(n_path, path_offset=0, device=None):
        return engines.setdefault((n_path, path_offset), FakeEngine(n_path, path_offset))

    logsv_pricer.get_engine = fake_get_engine
    heston_pricer.get_engine = fake_get_engine

    from cases import HESTON_CASE, LOGSV_CASE
    res = {}
    pr, sd = logsv_pricer.logsv_mc_chain_pricer(**LOGSV_CASE)
    res["logsv_prices"], res["logsv_stderrs"] = np.stack(pr), np.stack(sd)
    pr, sd = logsv_pricer.logsv_mc_chain_pricer(**{**LOGSV_CASE, "variable_type": VariableType.Q_VAR,
                                                   "strikes_ttms": tuple(0.5 * k for k in LOGSV_CASE["strikes_ttms"]),
                                                   "optiontypes_ttms": tuple(np.array(["C", "P", "C"])
                                                                             for _ in LOGSV_CASE["ttms"])})
    res["logsv_qv_prices"], res["logsv_qv_stderrs"] = np.stack(pr), np.stack(sd)
    pr, sd = heston_pricer.heston_mc_chain_pricer(**HESTON_CASE)
    res["heston_prices"], res["heston_stderrs"] = np.stack(pr), np.stack(sd)
    # fixed randoms: every rank holds the full host arrays and uses only its own columns
    W0s, W1s, dts = logsv_pricer.get_randoms_for_chain_valuation(LOGSV_CASE["ttms"], nb_path=LOGSV_CASE["nb_path"],
                                                                 nb_steps_per_year=LOGSV_CASE["nb_steps_per_year"],
                                                                 seed=3)
    kw = {k: v for k, v in LOGSV_CASE.items() if k not in ("nb_path", "nb_steps_per_year", "seed")}
    pr, sd = logsv_pricer.logsv_mc_chain_pricer_fixed_randoms(W0s=W0s, W1s=W1s, dts=dts, **kw)
    res["fixed_prices"], res["fixed_stderrs"] = np.stack(pr), np.stack(sd)
    # rough LogSV: fixed randoms (full host arrays on every rank) and device-side draws
    from cases import ROUGH_CASE
    Z0, Z1, grids = logsv_pricer.get_randoms_for_rough_vol_chain_valuation(ROUGH_CASE["ttms"], nb_path=1001,
                                                                           nb_steps_per_year=120, seed=4)
    pr, sd = logsv_pricer.rough_logsv_mc_chain_pricer_fixed_randoms(Z0=Z0, Z1=Z1, timegrids=grids, **ROUGH_CASE)
    res["rough_prices"], res["rough_stderrs"] = np.stack(pr), np.stack(sd)
    pr, sd = logsv_pricer.rough_logsv_mc_chain_pricer(nb_path=1001, nb_steps_per_year=120, seed=5, **ROUGH_CASE)
    res["rough_rng_prices"], res["rough_rng_stderrs"] = np.stack(pr), np.stack(sd)
    # a SHORTER chain after the longer ones: the collectives must reduce only the live prefix of the persistent buffers
    short = {k: (v[:1] if k in ("ttms", "forwards", "discfactors", "strikes_ttms", "optiontypes_ttms", "vol_backbone_etas")
                 else v) for k, v in LOGSV_CASE.items()}
    pr, sd = logsv_pricer.logsv_mc_chain_pricer(**short)
    res["short_prices"], res["short_stderrs"] = np.stack(pr), np.stack(sd)
    # un-seeded calls: rank 0's entropy seed and call counter are shared when the group is built
    from stochvolmodels_amd.utils import funcs
    res["rng_state"] = np.array(funcs.get_rng_state(), dtype=np.uint64)
    pr, sd = logsv_pricer.logsv_mc_chain_pricer(**{**short, "seed": None})
    res["unseeded_prices"] = np.stack(pr)
    res["phases"] = np.array(phases)
    if ladder is not None:
        import json
        res["ladder"] = np.array(json.dumps(ladder))
    res["rank_paths"] = np.array([e.n_path for e in engines.values()])
    res["rank_offsets"] = np.array([e.path_offset for e in engines.values()])
    np.savez(out_path + f".rank{comm.rank}.npz", **res)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
