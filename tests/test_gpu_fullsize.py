"""
Full-size, same-stream parity on the BASELINE configurations themselves (run with `-m gpu` on an MI355X).

C2 (LogSV, 2^20 paths x 1024 steps, 21 strikes), C3 (Heston, 2^22 paths x 4 expiries x 128 steps, 21 strikes per
expiry, both parameter sets, Euler and QE), one rank's share of C4 (LogSV, 2^21 paths x 8 expiries x 128 steps,
8 x 21 strikes) -- plain C / P, and once each with inverse options [P, IP, C, IC], as calls / puts on the quadratic
variance, and in the inverse measure -- and one rank's share of C5's Monte Carlo leg (five parameter sets, 2^20 paths x
4 expiries x 128 steps) are priced on the GPU through the product entry points AND on the CPU oracle fed the SAME
counter-based stream (oracle/svmc_oracle.c svo_*_terminal_rng, OpenMP over paths, a few seconds per configuration on
the box's host cores).  Per configuration the test PRINTS the largest deviation it observed (run with -s to read
them; profiles/r03_fullsize_parity.txt holds the round's) and asserts, per quantity, the tightest power of ten that
holds with the committed build:
  * terminal states path by path, |gpu - cpu| / (1 + |cpu|),
  * prices and standard errors, |gpu - cpu| / (|cpu| + 1e-3 x forward)   [Q_VAR: forward -> 1],
  * and BASELINE.json north_star's criterion verbatim: |price_gpu - price_cpu| <= 2 x MC-stderr, per option.
The two sides share the draw bit for bit (stream version 4: the same table and FMA sequence); they differ in the rounding
of exp / log / 1/x and in the accumulator form of the device step, amplified by sigma = exp(sum of increments).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BASE_HESTON = dict(v0=0.04, theta=0.04, kappa=4.0, rho=-0.5, volvol=0.4)        # C1 / HestonParams defaults

# asserted bounds (states, prices, stderrs).  Observed with the round-3 build (profiles/r03_fullsize_parity.txt): states
# 3e-16 .. 4.5e-14, prices 1e-15 .. 7e-14, standard errors 5e-17 .. 5e-14 -- SURVEY.md appendix B.7's 1e-12 (stepping) holds
# for all three everywhere but the inverse payoffs of the BTC set, whose division by the recentred spot shows 1.6e-12 in the
# prices and 2.4e-11 in the standard errors (B.7's figure for prices is 1e-11)
DEFAULT_BOUND = (1e-12, 1e-12, 1e-12)
BOUNDS = {"C4 share IC/IP": (1e-12, 1e-11, 1e-10)}


@pytest.fixture(scope="module")
def sv():
    import stochvolmodels_amd as sv
    from stochvolmodels_amd import _lib
    _lib.load()
    return sv


@pytest.fixture(scope="module")
def cpu(oracle):
    oracle.set_threads(oracle.effective_cores())
    return oracle


def _dev(g, o, floor):
    g, o = np.asarray(g, dtype=np.float64), np.asarray(o, dtype=np.float64)
    same = (g == o) | (np.isnan(g) & np.isnan(o))                       # equal infinities / NaN patterns count as equal
    d = np.where(same, 0.0, np.abs(g - o) / (np.abs(o) + floor))
    return float(np.max(d)) if d.size else 0.0


def _check(tag, gpu_state, cpu_state, pr, sd, opr, osd, price_floor=1e-3):
    b_state, b_price, b_stderr = BOUNDS.get(tag, DEFAULT_BOUND)
    devs = {name: _dev(g, o, 1.0) for name, g, o in zip(("x", "vol", "qvar"), gpu_state, cpu_state)}
    devs["price"] = max(_dev(pr[i], opr[i], price_floor) for i in range(len(pr)))
    devs["stderr"] = max(_dev(sd[i], osd[i], price_floor) for i in range(len(pr)))
    devs["price_in_stderr"] = max(float(np.max(np.abs(pr[i] - opr[i])[osd[i] > 0] / osd[i][osd[i] > 0], initial=0.0))
                                  for i in range(len(pr)))          # (an option no path reaches has stderr 0 on both sides)
    print(f"FULLSIZE PARITY {tag}: " + "  ".join(f"{k} {v:.2e}" for k, v in devs.items())
          + f"   [asserted: states {b_state:g}, prices {b_price:g}, stderrs {b_stderr:g}]")
    for name in ("x", "vol", "qvar"):
        assert devs[name] <= b_state, (tag, name, devs[name])
    assert devs["price"] <= b_price and devs["stderr"] <= b_stderr, (tag, devs)
    for i in range(len(pr)):
        # north_star: "option prices within 2x MC-stderr of the CPU reference" -- verbatim, on identical randoms
        assert np.all(np.abs(pr[i] - opr[i]) <= 2.0 * osd[i]), (tag, i)


def test_c2_logsv_full_size_same_stream(sv, cpu):
    from stochvolmodels_amd.engine import get_engine
    p = sv.LOGSV_BTC_PARAMS
    n, spy, seed = 1 << 20, 1023, 20240602
    kk = np.linspace(0.5, 1.5, 21)
    types = np.where(kk >= 1.0, "C", "P")
    pr, sd = sv.logsv_mc_chain_pricer(ttms=np.array([1.0]), forwards=np.ones(1), discfactors=np.ones(1),
                                      strikes_ttms=(kk,), optiontypes_ttms=(types,), v0=p.sigma0, theta=p.theta,
                                      kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta, volvol=p.volvol,
                                      vol_backbone_etas=np.ones(1), nb_path=n, nb_steps_per_year=spy, seed=seed)
    nb, dt, _ = sv.set_time_grid(1.0, spy)
    assert nb == 1024 and dt == 2.0 ** -10
    ox, os_, oq = cpu.logsv_terminal_rng(np.zeros(n), p.sigma0 * np.ones(n), np.zeros(n), nb, dt, p.theta, p.kappa1,
                                         p.kappa2, p.beta, p.volvol, seed)
    opr, osd = cpu.payoff(ox, oq, 1.0, 1.0, kk, types)
    _check("C2", get_engine(n).get_state(), (ox, os_, oq), pr, sd, [opr], [osd])


@pytest.mark.parametrize("scheme", ["euler", "qe"])
@pytest.mark.parametrize("tag", ["base", "btc"])
def test_c3_heston_full_size_same_stream(sv, cpu, tag, scheme):
    from stochvolmodels_amd.engine import get_engine
    if tag == "btc":
        h = sv.BTC_HESTON_PARAMS
        par = dict(v0=h.v0, theta=h.theta, kappa=h.kappa, rho=h.rho, volvol=h.volvol)
    else:
        par = BASE_HESTON
    n, spy, seed = 1 << 22, 508, 20240603
    ttms = np.array([0.25, 0.5, 0.75, 1.0])
    kk = np.linspace(0.5, 1.5, 21)
    types = np.where(kk >= 1.0, "C", "P")
    pr, sd = sv.heston_mc_chain_pricer(ttms=ttms, forwards=np.ones(4), discfactors=np.ones(4), strikes_ttms=(kk,) * 4,
                                       optiontypes_ttms=(types,) * 4, nb_path=n, scheme=scheme, nb_steps_per_year=spy,
                                       seed=seed, **par)
    x, v, q = np.zeros(n), par["v0"] * np.ones(n), np.zeros(n)
    opr, osd, t0, step0 = [], [], 0.0, 0
    for ttm in ttms:
        nb, dt, _ = sv.set_time_grid(ttm - t0, spy)
        assert nb == 128
        x, v, q = cpu.heston_terminal_rng(x, v, q, nb, dt, par["theta"], par["kappa"], par["rho"], par["volvol"], seed,
                                          scheme=cpu.HESTON_QE if scheme == "qe" else cpu.HESTON_EULER_FLOOR,
                                          step_offset=step0)
        a, b = cpu.payoff(x, q, float(ttm), 1.0, kk, types)
        opr.append(a), osd.append(b)
        t0, step0 = ttm, step0 + nb
    _check(f"C3 {tag} {scheme}", get_engine(n).get_state(), (x, v, q), pr, sd, opr, osd)


def _c4_chain(m=8):
    ttms = np.arange(1, m + 1) / 8.0
    fw = 67000.0 * np.exp(0.05 * ttms)
    dfs = np.exp(-0.05 * ttms)
    return ttms, fw, dfs


def _logsv_chain_gpu_vs_cpu(sv, cpu, tag, p, n, ttms, fw, dfs, strikes, types, seed, spy=1016, is_spot_measure=True,
                            variable_type=None, etas=None, price_floor=None, rank=0, world=1):
    """one LogSV chain on the GPU (product entry point) and slice by slice on the oracle, same stream.  world > 1: the n
    paths are RANK `rank`'s share of a job of n x world paths -- global path ids rank n .. (rank + 1) n - 1 -- priced through
    the product's sharded route with the cross-rank sums left out (a communicator that reduces nothing), so the prices are
    that share's own: what rank `rank` contributes, checked path by path and sum by sum against the oracle at that offset"""
    from stochvolmodels_amd import dist as svdist
    from stochvolmodels_amd.engine import get_engine
    m = len(ttms)
    etas = np.ones(m) if etas is None else etas
    vt = sv.VariableType.LOG_RETURN if variable_type is None else variable_type
    comm = None
    if world > 1:
        comm = svdist.SingleComm()
        comm.rank, comm.world = rank, world
        assert svdist.shard_range(n * world, rank, world) == (rank * n, n)
    pr, sd = sv.logsv_mc_chain_pricer(ttms=ttms, forwards=fw, discfactors=dfs, strikes_ttms=strikes,
                                      optiontypes_ttms=types, v0=p.sigma0, theta=p.theta, kappa1=p.kappa1,
                                      kappa2=p.kappa2, beta=p.beta, volvol=p.volvol, vol_backbone_etas=etas,
                                      is_spot_measure=is_spot_measure, nb_path=n * world, nb_steps_per_year=spy, seed=seed,
                                      variable_type=vt, comm=comm)
    sd = [e * np.sqrt(world) for e in sd]          # the job's standard error divides by sqrt(n world); the share's own by sqrt(n)
    x, s, q = np.zeros(n), p.sigma0 * np.ones(n), np.zeros(n)
    opr, osd, t0, step0 = [], [], 0.0, 0
    for i, ttm in enumerate(ttms):
        nb, dt, _ = sv.set_time_grid(ttm - t0, spy)
        assert nb == 128
        x, s, q = cpu.logsv_terminal_rng(x, s, q, nb, dt, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, seed,
                                         eta=float(etas[i]), is_spot_measure=is_spot_measure, step_offset=step0,
                                         path_offset=rank * n)
        a, b = cpu.payoff(x, q, float(ttm), float(fw[i]), strikes[i], types[i], float(dfs[i]), variable_type=int(vt.value))
        opr.append(a), osd.append(b)
        t0, step0 = ttm, step0 + nb
    floor = price_floor if price_floor is not None else 1e-3 * float(fw[0])
    _check(tag, get_engine(n, path_offset=rank * n).get_state(), (x, s, q), pr, sd, opr, osd, price_floor=floor)
    return pr, sd


def test_c4_rank_share_full_size_same_stream(sv, cpu):
    ttms, fw, dfs = _c4_chain()
    strikes = tuple(f * np.linspace(0.6, 1.6, 21) for f in fw)
    types = tuple(np.where(k >= f, "C", "P") for k, f in zip(strikes, fw))
    _logsv_chain_gpu_vs_cpu(sv, cpu, "C4 share", sv.LOGSV_BTC_PARAMS, 1 << 21, ttms, fw, dfs, strikes, types, 20240604)


@pytest.mark.parametrize("rank", [7, 3])
def test_c4_last_ranks_share_full_size_same_stream(sv, cpu, rank):
    """C4 as rank 7 (and rank 3) of the 8-GPU job sees it: 2^21 paths at path offset rank x 2^21 of the 2^24 -- the shares
    the one-GPU suite never ran before round 5 (every other full-size case is rank 0's share, offset 0): the counter's
    path words beyond 2^21, the engine's path_offset plumbing and the whole-chain kernel at a non-zero offset, against the
    oracle at the same offset"""
    ttms, fw, dfs = _c4_chain()
    strikes = tuple(f * np.linspace(0.6, 1.6, 21) for f in fw)
    types = tuple(np.where(k >= f, "C", "P") for k, f in zip(strikes, fw))
    _logsv_chain_gpu_vs_cpu(sv, cpu, f"C4 share of rank {rank}", sv.LOGSV_BTC_PARAMS, 1 << 21, ttms, fw, dfs, strikes, types,
                            20240604, rank=rank, world=8)


def test_c4_rank_share_inverse_options(sv, cpu):
    """the C4 rank share with the option types cycling [P, IP, C, IC] ("also one run with IC/IP"): the
    payoff_group_kernel<..., HAS_INV = 1> instantiation at full size -- inverse payoffs divide by the recentred spot and
    drop out of nanmean / nanstd where it is not finite (utils/mc_payoffs.py:66-83)"""
    ttms, fw, dfs = _c4_chain()
    strikes = tuple(f * np.linspace(0.6, 1.6, 21) for f in fw)
    cyc = np.array(["P", "IP", "C", "IC"])
    types = tuple(cyc[np.arange(21) % 4] for _ in fw)
    # an inverse payoff is a fraction of the spot: O(0.1), not O(forward)
    _logsv_chain_gpu_vs_cpu(sv, cpu, "C4 share IC/IP", sv.LOGSV_BTC_PARAMS, 1 << 21, ttms, fw, dfs, strikes, types, 20240605,
                            price_floor=1e-3)


def test_c4_rank_share_quadratic_variance(sv, cpu):
    """the C4 rank share as calls / puts on the annualised quadratic variance (strikes in variance units): the
    <..., NEED_Q = 1> instantiation and the qvar snapshot rows at full size, with a vol backbone"""
    ttms, fw, dfs = _c4_chain()
    kv = np.linspace(0.2, 1.6, 21)                       # BTC set: sigma0^2 = 0.70, theta^2 = 1.08
    strikes = tuple(kv for _ in fw)
    types = tuple(np.where(kv >= 0.8, "C", "P") for _ in fw)
    etas = np.linspace(0.9, 1.1, 8)
    _logsv_chain_gpu_vs_cpu(sv, cpu, "C4 share Q_VAR", sv.LOGSV_BTC_PARAMS, 1 << 21, ttms, fw, dfs, strikes, types, 20240606,
                            variable_type=sv.VariableType.Q_VAR, etas=etas, price_floor=1e-3)


def test_c4_rank_share_inverse_measure(sv, cpu):
    """the C4 rank share simulated in the inverse measure (pricers/logsv_pricer.py:1032-1035: alpha = +1, adj = beta eta).
    Plain calls / puts: under the BTC set the ADDITIVE recentring of utils/mc_payoffs.py:61-63 leaves spots near and below
    zero at these maturities (E[S] under the inverse measure is several forwards), which the plain payoffs tolerate; the
    inverse payoffs divide by those spots -- that combination is checked on a parameter set where it is well posed
    (test_inverse_measure_inverse_options)"""
    ttms, fw, dfs = _c4_chain()
    strikes = tuple(f * np.linspace(0.6, 1.6, 21) for f in fw)
    types = tuple(np.where(k >= f, "C", "P") for k, f in zip(strikes, fw))
    _logsv_chain_gpu_vs_cpu(sv, cpu, "C4 share inverse measure", sv.LOGSV_BTC_PARAMS, 1 << 21, ttms, fw, dfs, strikes, types,
                            20240607, is_spot_measure=False)


def test_inverse_measure_inverse_options(sv, cpu):
    """inverse calls / puts priced in the inverse measure at the C4 rank share's size on the 20 %-vol parameter set, where
    the recentred spot stays positive and the inverse payoffs are well conditioned"""
    ttms, fw, dfs = _c4_chain()
    strikes = tuple(f * np.linspace(0.7, 1.4, 21) for f in fw)
    types = tuple(np.where(k >= f, "IC", "IP") for k, f in zip(strikes, fw))
    p = sv.LogSvParams(**C5_SETS["test"])
    _logsv_chain_gpu_vs_cpu(sv, cpu, "inverse measure IC/IP (20 % vol set)", p, 1 << 21, ttms, fw, dfs, strikes, types, 20240608,
                            is_spot_measure=False, price_floor=1e-3)


C5_SETS = {   # LOGSV_BTC_PARAMS, README calibrated, quickstart, the tests' stiff set, the article's Fig. 3
    "btc": None,
    "readme": dict(sigma0=0.8327, theta=1.0139, kappa1=4.8609, kappa2=4.7940, beta=0.1988, volvol=2.3694),
    "quick": dict(sigma0=1.0, theta=1.0, kappa1=5.0, kappa2=5.0, beta=0.2, volvol=2.0),
    "test": dict(sigma0=0.2, theta=0.22, kappa1=3.0, kappa2=12.0, beta=-0.3, volvol=0.4),
    "fig3": dict(sigma0=1.5, theta=1.0, kappa1=4.0, kappa2=4.0, beta=0.0, volvol=1.5),
}


@pytest.mark.parametrize("tag", list(C5_SETS))
def test_c5_monte_carlo_leg_rank_share(sv, cpu, golden, tag):
    """C5's Monte Carlo leg on ITS workload: each of the five parameter sets on C4's first four expiries, one rank's share
    of the 2^23 paths (2^20 paths x 4 x 128 steps, 4 x 21 strikes), GPU vs the oracle on the same stream -- the stiff
    kappa2 = 12 set, sigma0 = 1.5 and volvol 2.37 meet the oracle at scale here.

    And C5's own criterion on C5's own chain, as VERDICT PARITY: per option, does |analytic - MC| <= 4 stderr hold?  The
    reference's answer is committed (tests/golden/c5_verdict.npz, made by make_golden.py g_c5_verdict: the UNMODIFIED
    reference's analytic chain against the oracle's Monte Carlo on this very stream); the GPU's answer -- its analytic chain
    against its Monte Carlo -- must be the same map, option by option, outside a dead band around the threshold, and the
    z-scores themselves must agree (NB the oracle's analytic integrator is a 5(4) pair, the device's an 8(5,3) pair: the two
    meet at their tolerances, 1e-10, not bit for bit -- on the analytic side the "CPU twin" is a tolerance twin).  (Four sets pass on all 84 options; the kappa2 = 12
    set fails 22 of them at the first expiries, where the standard error of 2^20 paths is far below the truncation error of
    the reference's second-order expansion -- a property of the reference's approximation, tabulated in
    profiles/r02_c5_bias.json, that a drop-in must reproduce, not hide.)"""
    p = sv.LOGSV_BTC_PARAMS if C5_SETS[tag] is None else sv.LogSvParams(**C5_SETS[tag])
    ttms, fw, dfs = _c4_chain(4)
    strikes = tuple(f * np.linspace(0.6, 1.6, 21) for f in fw)
    types = tuple(np.where(k >= f, "C", "P") for k, f in zip(strikes, fw))
    g = golden("c5_verdict")
    assert [int(v) for v in g["mc"]] == [1 << 20, 1016, 20240610]              # the golden's Monte Carlo leg is this run
    np.testing.assert_array_equal(g["strikes"], np.stack(strikes))
    np.testing.assert_array_equal(g["ttms"], ttms)
    pr, sd = _logsv_chain_gpu_vs_cpu(sv, cpu, f"C5 {tag}", p, 1 << 20, ttms, fw, dfs, strikes, types, 20240610)
    chain = sv.OptionChain(ttms=ttms, forwards=fw, strikes_ttms=strikes, optiontypes_ttms=types, ids=None, discfactors=dfs)
    an = np.stack(sv.LogSVPricer().price_chain(chain, p))
    pr, sd = np.stack(pr), np.stack(sd)
    z = (pr - an) / np.where(sd > 0, sd, np.nan)
    verdict = np.where(np.isnan(z), -1, (np.abs(z) <= 4.0).astype(int))
    print(f"C5 {tag}: (MC - analytic) / stderr per expiry, min .. max over the strikes some path reaches: "
          + ", ".join(f"[{np.nanmin(row):+.1f} .. {np.nanmax(row):+.1f}]" for row in z)
          + f"; verdict: pass {int(np.sum(verdict == 1))}, fail {int(np.sum(verdict == 0))}, unreached {int(np.sum(verdict == -1))}"
          + f"; closest |z| to the threshold 4 +- {np.nanmin(np.abs(np.abs(z) - 4.0)):.3f}"
          + f"; max |z_gpu - z_reference| {np.nanmax(np.abs(z - g[f'{tag}_z'])):.2e}")
    # the GPU's Monte Carlo leg IS the golden's (same stream; _logsv_chain_gpu_vs_cpu held it to the oracle at 1e-12) and its
    # analytic chain meets the reference's to the reference's own solver tolerance (rtol 1e-3 RK45: 2e-6 of the forward)
    np.testing.assert_allclose(pr, g[f"{tag}_mc"], rtol=1e-11, atol=1e-11 * float(fw[0]))
    np.testing.assert_allclose(an, g[f"{tag}_analytic"], rtol=0, atol=2e-6 * float(fw[0]))
    # verdict parity with a dead band (tests/test_gpu_parity.py VERDICT_DELTA): the z-scores agree to DELTA everywhere some
    # path reaches the option, and the maps are equal wherever the reference's |z| is more than 10 DELTA from the threshold
    # (the "test" set's closest option sits 0.11 of a z-unit from it: its side is not a statement about parity)
    delta = 0.05
    z_ref = g[f"{tag}_z"]
    reached = ~np.isnan(z_ref)
    np.testing.assert_array_equal(np.isnan(z), ~reached)
    assert np.nanmax(np.abs(z - z_ref)) <= delta, (tag, np.nanmax(np.abs(z - z_ref)))
    clear = ~reached | (np.abs(np.abs(np.where(reached, z_ref, 0.0)) - 4.0) > 10.0 * delta)
    np.testing.assert_array_equal(verdict[clear], g[f"{tag}_pass"][clear],
                                  err_msg=f"C5 {tag}: the GPU's accept / reject map differs from the reference's away from the threshold")


@pytest.mark.parametrize("tag", ["btc", "test"])
def test_c5_monte_carlo_leg_at_the_stated_2e23_paths_on_one_gpu(sv, cpu, tag):
    """C5's Monte Carlo leg at the path count SURVEY.md 8d states -- 2^23 paths on ONE device, not a rank's 2^20 share (it
    fits: 2^23 x 11 state / snapshot rows = 740 MB) -- for the BTC set and the stiff kappa2 = 12 set, GPU vs the oracle on the
    same stream, path by path and option by option"""
    p = sv.LOGSV_BTC_PARAMS if C5_SETS[tag] is None else sv.LogSvParams(**C5_SETS[tag])
    ttms, fw, dfs = _c4_chain(4)
    strikes = tuple(f * np.linspace(0.6, 1.6, 21) for f in fw)
    types = tuple(np.where(k >= f, "C", "P") for k, f in zip(strikes, fw))
    _logsv_chain_gpu_vs_cpu(sv, cpu, f"C5 {tag} 2^23", p, 1 << 23, ttms, fw, dfs, strikes, types, 20240611)
