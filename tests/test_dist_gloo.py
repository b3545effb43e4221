"""
The N>1 path on CPU: world_size-1/2/3/8 gloo groups run the PRODUCT chain drivers (path sharding by global path
id, per-slice step offsets, the two packed all-reduces, host finalisation) with the engine swapped for the
oracle-backed test double.  Every world size must reproduce the single-process oracle result for the full
path set: sharding must not change which randoms a path sees, and the partial sums must add up.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _expected(oracle):
    from cases import HESTON_CASE, LOGSV_CASE
    from stochvolmodels_amd.utils.funcs import set_time_grid
    c = LOGSV_CASE
    n = c["nb_path"]
    out = {}
    for tag, vt, strikes, types in (
            ("logsv", 1, c["strikes_ttms"], c["optiontypes_ttms"]),
            ("logsv_qv", 2, tuple(0.5 * k for k in c["strikes_ttms"]), tuple(np.array(["C", "P", "C"]) for _ in c["ttms"]))):
        x, s, q = np.zeros(n), c["v0"] * np.ones(n), np.zeros(n)
        t0, step0, P, E = 0.0, 0, [], []
        for i, ttm in enumerate(c["ttms"]):
            nb, dt, _ = set_time_grid(ttm - t0, c["nb_steps_per_year"])
            x, s, q = oracle.logsv_terminal_rng(x, s, q, nb, dt, c["theta"], c["kappa1"], c["kappa2"], c["beta"],
                                                c["volvol"], c["seed"], eta=float(c["vol_backbone_etas"][i]),
                                                step_offset=step0)
            step0, t0 = step0 + nb, ttm
            p, e = oracle.payoff(x, q, float(ttm), float(c["forwards"][i]), strikes[i], types[i],
                                 float(c["discfactors"][i]), vt)
            P.append(p), E.append(e)
        out[f"{tag}_prices"], out[f"{tag}_stderrs"] = np.stack(P), np.stack(E)
    h = HESTON_CASE
    x, v, q = np.zeros(n), h["v0"] * np.ones(n), np.zeros(n)
    t0, step0, P, E = 0.0, 0, [], []
    for i, ttm in enumerate(h["ttms"]):
        nb, dt, _ = set_time_grid(ttm - t0, h["nb_steps_per_year"])
        x, v, q = oracle.heston_terminal_rng(x, v, q, nb, dt, h["theta"], h["kappa"], h["rho"], h["volvol"], h["seed"],
                                             scheme=oracle.HESTON_QE, step_offset=step0)
        step0, t0 = step0 + nb, ttm
        p, e = oracle.payoff(x, q, float(ttm), float(h["forwards"][i]), h["strikes_ttms"][i], h["optiontypes_ttms"][i],
                             float(h["discfactors"][i]))
        P.append(p), E.append(e)
    out["heston_prices"], out["heston_stderrs"] = np.stack(P), np.stack(E)
    from stochvolmodels_amd.pricers.logsv_pricer import get_randoms_for_chain_valuation
    W0s, W1s, dts = get_randoms_for_chain_valuation(c["ttms"], nb_path=n, nb_steps_per_year=c["nb_steps_per_year"], seed=3)
    P, E = oracle.logsv_chain_fixed_randoms(c["ttms"], c["forwards"], c["discfactors"], c["strikes_ttms"],
                                            c["optiontypes_ttms"], W0s, W1s, dts, c["v0"], c["theta"], c["kappa1"],
                                            c["kappa2"], c["beta"], c["volvol"], c["vol_backbone_etas"])
    out["fixed_prices"], out["fixed_stderrs"] = np.stack(P), np.stack(E)
    from cases import ROUGH_CASE as r
    args = (r["ttms"], r["forwards"], r["discfactors"], r["strikes_ttms"], r["optiontypes_ttms"])
    pars = (r["sigma0"], r["theta"], r["kappa1"], r["kappa2"], r["beta"], r["orthog_vol"], r["weights"], r["nodes"])
    Z0, Z1, grids = oracle.rough_randoms(r["ttms"], 1001, 120, seed=4)
    P, E = oracle.rough_logsv_chain_fixed_randoms(*args, Z0, Z1, *pars, grids)
    out["rough_prices"], out["rough_stderrs"] = np.stack(P), np.stack(E)
    Z0, Z1 = oracle.fill_normals(5, 1001, grids[-1].size - 1, call_id=0, stream=3)
    P, E = oracle.rough_logsv_chain_fixed_randoms(*args, Z0, Z1, *pars, grids)
    out["rough_rng_prices"], out["rough_rng_stderrs"] = np.stack(P), np.stack(E)
    return out


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_sharded_chain_matches_single_process(oracle, tmp_path, world):
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29611 + world), WORLD_SIZE=str(world),
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), out],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    exp = _expected(oracle)
    exp["short_prices"], exp["short_stderrs"] = exp["logsv_prices"][:1], exp["logsv_stderrs"][:1]
    offsets = []
    rank0 = np.load(out + ".rank0.npz")
    for r in range(world):
        got = np.load(out + f".rank{r}.npz")
        # every rank keys Philox with rank 0's (seed, call counter): un-seeded prices are one global result
        assert np.array_equal(got["rng_state"], rank0["rng_state"]), (r, got["rng_state"], rank0["rng_state"])
        assert np.array_equal(got["unseeded_prices"], rank0["unseeded_prices"])
        for key, ref in exp.items():                       # every rank returns the global result
            np.testing.assert_allclose(got[key], ref, rtol=1e-11, atol=1e-14, err_msg=f"{key} rank {r}/{world}")
        # the phases a launcher's watchdog arms its deadlines from (bench.py): none for a lone process without a group
        assert list(got["phases"]) == ([] if world == 1 else ["rendezvous", "collective_init", "ready"])
        assert set(got["rank_paths"]) == {(1001 * (r + 1)) // world - (1001 * r) // world}
        offsets.append(int(got["rank_offsets"][0]))
    assert offsets == [(1001 * r) // world for r in range(world)]


def test_shard_range_partitions_exactly():
    from stochvolmodels_amd.dist import shard_range
    for n in (1, 7, 1000, 1 << 20, (1 << 24) + 5):
        for world in (1, 2, 3, 4, 8):
            lo = 0
            for r in range(world):
                off, cnt = shard_range(n, r, world)
                assert off == lo and cnt >= 0
                lo += cnt
            assert lo == n


@pytest.mark.parametrize("mode", ["no_gpu", "probe_fails", "probe_hangs"])
def test_fallback_ladder_lands_on_gloo_with_a_result(oracle, tmp_path, mode):
    """dist.init_with_fallback on a box where RCCL cannot come up: every rung above gloo is tried -- directly refused without a
    device ("no_gpu"), failing inside its probe child ("probe_fails": the children really start, rendezvous on their own
    port and report why they failed), or HANGING inside its probe child ("probe_hangs": fault injection; the parent kills
    the child at the deadline) -- and the job still prices on the gloo control plane: the same numbers, the reasons reported"""
    import json
    world = 2
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29651 + ["no_gpu", "probe_fails", "probe_hangs"].index(mode)),
               WORLD_SIZE=str(world), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), SVMC_TEST_LADDER="1")
    if mode != "no_gpu":
        env["SVMC_DIST_FORCE_PROBE"] = "1"
    if mode == "probe_hangs":
        env.update(SVMC_BENCH_FAULT="nccl_hang", SVMC_TEST_PROBE_TIMEOUT="6", SVMC_DIST_RUNGS="nccl,gloo")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), out],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    exp = _expected(oracle)
    for r in range(world):
        got = np.load(out + f".rank{r}.npz")
        for key in ("logsv_prices", "logsv_stderrs", "heston_prices", "fixed_prices"):
            np.testing.assert_allclose(got[key], exp[key], rtol=1e-11, atol=1e-14, err_msg=f"{key} rank {r}")
        rep = json.loads(str(got["ladder"]))
        assert rep["rung"] == "gloo" and rep["comm"] == "TorchComm" and rep["backend"] == "gloo" and rep["control_plane"] == "gloo"
        tried = [p["rung"] for p in rep["probes"]]
        assert tried == (["nccl", "gloo"] if mode == "probe_hangs" else ["nccl", "rccl", "gloo"])
        assert all(not p["ok"] for p in rep["probes"][:-1]) and rep["probes"][-1]["ok"]
        reason = rep["comm_fallback_reason"]
        if mode == "no_gpu":
            assert "no HIP device visible" in reason
        elif mode == "probe_fails":
            assert "nccl: rank 0:" in reason and "rccl: rank 0:" in reason       # the children's own last words
        else:
            assert "did not finish within 6 s (killed)" in reason and 5.0 <= rep["probes"][0]["seconds"] < 60.0
        phases = list(got["phases"])
        assert phases[0] == "rendezvous" and phases[-1] == "ready"
