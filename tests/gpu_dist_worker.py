"""worker of test_gpu_parity.py::test_two_ranks_share_one_gpu: a rank of a gloo group on the real HIP engine."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(out_path):
    import torch.distributed as dist

    from cases import HESTON_CASE, LOGSV_CASE
    from stochvolmodels_amd import dist as svdist
    from stochvolmodels_amd.pricers import heston_pricer, logsv_pricer

    comm = svdist.init_from_env()
    if os.environ.get("SVMC_EXPECT_BACKEND"):
        assert dist.is_initialized() and dist.get_backend() == os.environ["SVMC_EXPECT_BACKEND"], "group not built"
        assert type(comm).__name__ == os.environ.get("SVMC_EXPECT_COMM", "TorchComm")
    res = {}
    pr, sd = logsv_pricer.logsv_mc_chain_pricer(**LOGSV_CASE)
    res["logsv_prices"], res["logsv_stderrs"] = np.stack(pr), np.stack(sd)
    pr, sd = heston_pricer.heston_mc_chain_pricer(**HESTON_CASE)
    res["heston_prices"], res["heston_stderrs"] = np.stack(pr), np.stack(sd)
    W0s, W1s, dts = logsv_pricer.get_randoms_for_chain_valuation(LOGSV_CASE["ttms"], nb_path=LOGSV_CASE["nb_path"],
                                                                 nb_steps_per_year=LOGSV_CASE["nb_steps_per_year"], seed=3)
    kw = {k: v for k, v in LOGSV_CASE.items() if k not in ("nb_path", "nb_steps_per_year", "seed")}
    pr, sd = logsv_pricer.logsv_mc_chain_pricer_fixed_randoms(W0s=W0s, W1s=W1s, dts=dts, **kw)
    res["fixed_prices"], res["fixed_stderrs"] = np.stack(pr), np.stack(sd)
    np.savez(out_path + f".rank{comm.rank}.npz", **res)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
