"""
Full-size, same-stream parity on the BASELINE configurations themselves (run with `-m gpu` on an MI355X).

C2 (LogSV, 2^20 paths x 1024 steps, 21 strikes), C3 (Heston, 2^22 paths x 4 expiries x 128 steps, 21 strikes per
expiry, both parameter sets, Euler and QE) and one rank's share of C4 (LogSV, 2^21 paths x 8 expiries x 128 steps,
8 x 21 strikes) are priced on the GPU through the product entry points AND on the CPU oracle fed the SAME
counter-based stream (oracle/svmc_oracle.c svo_*_terminal_rng, OpenMP over paths, a few seconds per configuration on
the box's host cores).  Asserted, per configuration:
  * terminal states path by path: 1e-9 (the two sides differ only in the rounding of their elementary functions
    and in FMA contraction; sigma / variance dynamics are contracting, so the differences do not grow),
  * prices and standard errors: 1e-9,
  * and BASELINE.json north_star's criterion verbatim: |price_gpu - price_cpu| <= 2 x MC-stderr, per option.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BASE_HESTON = dict(v0=0.04, theta=0.04, kappa=4.0, rho=-0.5, volvol=0.4)        # C1 / HestonParams defaults


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


def _check(tag, gpu_state, cpu_state, pr, sd, opr, osd, scale=1.0):
    for name, g, o in zip(("x", "vol", "qvar"), gpu_state, cpu_state):
        np.testing.assert_allclose(g, o, rtol=1e-9, atol=1e-9, err_msg=f"{tag}: terminal {name}")
    for i in range(len(pr)):
        np.testing.assert_allclose(pr[i], opr[i], rtol=1e-9, atol=1e-9 * scale, err_msg=f"{tag}: prices, expiry {i}")
        np.testing.assert_allclose(sd[i], osd[i], rtol=1e-9, atol=1e-9 * scale, err_msg=f"{tag}: stderrs, expiry {i}")
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


def test_c4_rank_share_full_size_same_stream(sv, cpu):
    from stochvolmodels_amd.engine import get_engine
    p = sv.LOGSV_BTC_PARAMS
    n, spy, seed = 1 << 21, 1016, 20240604
    ttms = np.arange(1, 9) / 8.0
    fw = 67000.0 * np.exp(0.05 * ttms)
    dfs = np.exp(-0.05 * ttms)
    strikes = tuple(f * np.linspace(0.6, 1.6, 21) for f in fw)
    types = tuple(np.where(k >= f, "C", "P") for k, f in zip(strikes, fw))
    pr, sd = sv.logsv_mc_chain_pricer(ttms=ttms, forwards=fw, discfactors=dfs, strikes_ttms=strikes,
                                      optiontypes_ttms=types, v0=p.sigma0, theta=p.theta, kappa1=p.kappa1,
                                      kappa2=p.kappa2, beta=p.beta, volvol=p.volvol, vol_backbone_etas=np.ones(8),
                                      nb_path=n, nb_steps_per_year=spy, seed=seed)
    x, s, q = np.zeros(n), p.sigma0 * np.ones(n), np.zeros(n)
    opr, osd, t0, step0 = [], [], 0.0, 0
    for i, ttm in enumerate(ttms):
        nb, dt, _ = sv.set_time_grid(ttm - t0, spy)
        assert nb == 128
        x, s, q = cpu.logsv_terminal_rng(x, s, q, nb, dt, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, seed,
                                         step_offset=step0)
        a, b = cpu.payoff(x, q, float(ttm), float(fw[i]), strikes[i], types[i], float(dfs[i]))
        opr.append(a), osd.append(b)
        t0, step0 = ttm, step0 + nb
    _check("C4 share", get_engine(n).get_state(), (x, s, q), pr, sd, opr, osd, scale=67000.0)
