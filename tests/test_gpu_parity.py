"""
GPU parity tests (run with `-m gpu` on an MI355X).  Everything goes through libsvmc's C ABI via the host
mirror of the reference API and is compared with
  - the golden vectors produced from the reference (tests/golden/*.npz), and
  - the CPU oracle on the same inputs / the same Philox stream.

Tolerances (fp64 throughout): the HIP kernels keep the reference's evaluation order but contract a*b+c to
FMA and use hand-written exp / log / sqrt (<= 1-4 ULP), so on identical randoms the state agrees to rounding level
amplified by the path's own dynamics (sigma is an exponential of a sum of ~100-1000 increments).  Every tolerance
below was set from a recording run (SVMC_RECORD_TOLERANCES, tests/conftest.py; the observed deviations are committed as
profiles/r03_observed_tolerances.txt): the tightest power of ten that is >= 10 x the largest deviation observed, never
below SURVEY App. B.7's 1e-12 (stepping) -- so 1e-12 on states / sums, ST = 1e-11 on prices and standard errors against
the reference's golden vectors (numpy's libm on the other side).  Statistical checks use the reference's own criterion
|MC - analytic| <= 4 stderr (reference tests/test_logsv_characterization.py:407).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ST = dict(rtol=1e-11, atol=1e-13)          # prices 1e-11 (stepping states are asserted at 1e-12)
# verdict parity (|analytic - MC| <= 4 stderr, option by option, GPU vs reference): the z-scores must agree to VERDICT_DELTA
# (observed <= 1.4e-2: the two sides' analytic solvers meet at their tolerances, profiles/r04_fullsize_parity.txt) and the maps
# must be equal wherever the reference's |z| is more than 10 VERDICT_DELTA away from the threshold
VERDICT_DELTA = 0.05


def P(v):
    return dict(zip(("v0", "theta", "kappa1", "kappa2", "beta", "volvol"), (float(a) for a in v)))


@pytest.fixture(scope="module")
def sv():
    import stochvolmodels_amd as sv
    from stochvolmodels_amd import _lib
    _lib.load()
    return sv


def _engine(n, offset=0):
    from stochvolmodels_amd.engine import HipEngine
    return HipEngine(n, path_offset=offset)


# ---------------------------------------------------------------------------------------------------
def test_library_loaded_and_device(sv):
    import ctypes as C
    from stochvolmodels_amd import _lib
    L = _lib.load()
    n = C.c_int()
    assert L.svmc_device_count(C.byref(n)) == 0 and n.value >= 1
    name = C.create_string_buffer(128)
    cus, khz, mem = C.c_int(), C.c_int(), C.c_size_t()
    assert L.svmc_device_info(0, name, 128, C.byref(cus), C.byref(khz), C.byref(mem)) == 0
    assert b"gfx950" in name.value, name.value
    assert cus.value == 256


def test_normals_match_oracle_stream(sv, oracle):
    n, nb, seed = 5000, 37, 20240601
    eng = _engine(n, offset=123456789012)
    w0p, w1p = eng.fill_normals(nb, seed, call_id=3, step_offset=11)
    W0 = eng.download(w0p, nb * n).reshape(nb, n)
    W1 = eng.download(w1p, nb * n).reshape(nb, n)
    O0, O1 = oracle.fill_normals(seed, n, nb, call_id=3, path_offset=123456789012, step_offset=11)
    np.testing.assert_array_equal(W0, O0)                   # the stream is DEFINED by the twin's expression: bit for bit
    np.testing.assert_array_equal(W1, O1)
    eng.close()


def test_normal_stream_at_scale_on_device(sv):
    """the counter-based stream at the scale it is used at: 2^28 normals of each component (2^22 paths x 64 steps:
    global path ids and step indices both exercised, Philox-7, two steps per call) through the product's own
    reduction -- sum exp(z) / n against the exact E[exp(Z)] = e^(1/2), four standard errors, no slack; the statistic
    weighs the upper tail, where a generator defect or a wrong radius / angle map would show first"""
    n, nb = 1 << 22, 64
    eng = _engine(n, offset=3 * (1 << 40))
    out_ptr, _ = eng.alloc_sums(2, "mgf")
    exact, sd1 = np.exp(0.5), np.sqrt(np.exp(2.0) - np.exp(1.0))
    for seed in (1, 20240927):
        w0p, w1p = eng.fill_normals(nb, seed, call_id=2, step_offset=1001)       # an odd first step: half a call
        for ptr in (w0p, w1p):
            tot = 0.0
            for t in range(nb):                          # each row is one time step of all paths
                eng.spot_sums(ptr + 8 * t * n, 1.0, out_ptr)
                sums = eng.download(out_ptr, 2)
                assert sums[1] == n
                tot += sums[0]
            assert abs(tot / (n * nb) - exact) <= 4.0 * sd1 / np.sqrt(n * nb), (seed, tot / (n * nb) - exact)
    eng.close()


def test_uniforms_match_oracle_stream(sv, oracle):
    import ctypes as C
    from stochvolmodels_amd import _lib
    from stochvolmodels_amd.engine import DeviceBuffer
    n, nb = 3000, 5
    eng = _engine(n)
    buf = DeviceBuffer(n * nb)
    _lib.check(eng.lib.svmc_fill_uniforms(buf.ptr, n, n, nb, 77, 1, 5, 9, None))
    U = eng.download(buf.ptr, n * nb).reshape(nb, n)
    np.testing.assert_array_equal(U, oracle.fill_uniforms(77, n, nb, call_id=1, path_offset=5, step_offset=9))
    eng.close()


def test_time_grid(sv, golden):
    for ttm, spy, n, dt in golden("time_grid")["cases"]:
        nb, d, grid = sv.set_time_grid(ttm, int(spy))
        assert (nb, d) == (int(n), dt) and grid.shape == (nb + 1,)


def test_logsv_zero_noise(sv, golden):
    g = golden("logsv_zero_noise")
    p = P(g["params"])
    z = np.zeros((int(g["nb_steps"]), 1))
    for row, spot in zip(g["terminal"], (True, False)):
        x, s, q = sv.simulate_logsv_x_vol_terminal(ttm=0.25, x0=np.zeros(1), sigma0=np.array([p["v0"]]),
                                                   qvar0=np.zeros(1), theta=p["theta"], kappa1=p["kappa1"],
                                                   kappa2=p["kappa2"], beta=p["beta"], volvol=p["volvol"], nb_path=1,
                                                   W0=z, W1=z, dt=float(g["dt"]), is_spot_measure=spot)
        np.testing.assert_allclose([x[0], s[0], q[0]], row, rtol=1e-12)


def test_logsv_tiny_chain_fixed_randoms(sv, golden):
    g = golden("logsv_tiny_chain")
    p = P(g["params"])
    pr, sd = sv.logsv_mc_chain_pricer_fixed_randoms(
        ttms=g["ttms"], forwards=g["forwards"], discfactors=g["discfactors"], strikes_ttms=tuple(g["strikes"]),
        optiontypes_ttms=tuple(g["types"]), W0s=[g["W0_0"], g["W0_1"]], W1s=[g["W1_0"], g["W1_1"]], dts=g["dts"],
        vol_backbone_etas=np.ones(2), **p)
    np.testing.assert_allclose(np.stack(pr), g["prices"], **ST)
    np.testing.assert_allclose(np.stack(sd), g["stderrs"], **ST)
    # replay of the reference's RandomState contract (tests/test_logsv_characterization.py:583-602)
    W0s, W1s, dts = sv.get_randoms_for_chain_valuation(g["ttms"], nb_path=8, nb_steps_per_year=int(g["spy"]), seed=7)
    np.testing.assert_array_equal(W0s[0], g["W0_0"])
    np.testing.assert_array_equal(W1s[1], g["W1_1"])
    np.testing.assert_array_equal(dts, g["dts"])


def _philox_randoms(oracle, g):
    W0s, W1s, step0 = [], [], 0
    for nb in g["nb_steps"]:
        W0, W1 = oracle.fill_normals(int(g["seed"]), int(g["n_path"]), int(nb), step_offset=step0)
        W0s.append(W0), W1s.append(W1)
        step0 += int(nb)
    return W0s, W1s


@pytest.mark.parametrize("tag,spot,vt", [("spot", True, 1), ("inv", False, 1), ("qvar", True, 2)])
def test_logsv_chain_philox_vs_reference(sv, oracle, golden, tag, spot, vt):
    """three expiries, vol backbone, IC/IP payoffs: (a) streamed kernel fed the stream the reference was fed,
    (b) on-device RNG kernel with the same seed -- both against the reference's outputs."""
    g = golden("logsv_chain_philox")
    p = P(g["params"])
    vtype = sv.VariableType(vt)
    strikes, types = (g["qv_strikes"], g["qv_types"]) if tag == "qvar" else (g["strikes"], g["types"])
    W0s, W1s = _philox_randoms(oracle, g)
    common = dict(ttms=g["ttms"], forwards=g["forwards"], discfactors=g["discfactors"], strikes_ttms=tuple(strikes),
                  optiontypes_ttms=tuple(types), vol_backbone_etas=g["etas"], is_spot_measure=spot,
                  variable_type=vtype, **p)
    pr, sd = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=W0s, W1s=W1s, dts=g["dts"], **common)
    np.testing.assert_allclose(np.stack(pr), g[f"prices_{tag}"], **ST)
    np.testing.assert_allclose(np.stack(sd), g[f"stderrs_{tag}"], **ST)
    pr, sd = sv.logsv_mc_chain_pricer(nb_path=int(g["n_path"]), nb_steps_per_year=int(g["spy"]), seed=int(g["seed"]),
                                      **common)
    np.testing.assert_allclose(np.stack(pr), g[f"prices_{tag}"], **ST)
    np.testing.assert_allclose(np.stack(sd), g[f"stderrs_{tag}"], **ST)
    if tag != "qvar":
        from stochvolmodels_amd.engine import get_engine
        x, s, q = get_engine(int(g["n_path"])).get_state()      # terminal state of the last expiry, all paths
        np.testing.assert_allclose(np.stack([x, s, q]), g[f"states_{tag}"][-1], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("tag,spot,vt", [("spot", True, 1), ("inv", False, 1), ("qvar", True, 2)])
def test_logsv_philox_c4_shape_vs_reference(sv, golden, tag, spot, vt):
    """the on-device-RNG chain pricer at 2^16 paths on bench config C4's chain (8 x 128 steps, 8 x 21 strikes, BTC-scale forwards)
    against the REFERENCE's logsv_mc_chain_pricer_fixed_randoms fed the same Philox stream (philox_c4_shape.npz: prices, standard
    errors, the first 256 paths' terminal states) -- the whole-chain stepping kernel, its slice loop and epilogues, the payoff
    pass, 32 x the paths of logsv_chain_philox.npz"""
    g = golden("philox_c4_shape")
    p = P(g["logsv_params"])
    strikes, types = (g["qv_strikes"], g["qv_types"]) if tag == "qvar" else (g["strikes"], g["types"])
    n = int(g["n_path"])
    pr, sd = sv.logsv_mc_chain_pricer(ttms=g["ttms"], forwards=g["forwards"], discfactors=g["discfactors"], strikes_ttms=tuple(strikes),
                                      optiontypes_ttms=tuple(types), vol_backbone_etas=np.ones(8), is_spot_measure=spot,
                                      variable_type=sv.VariableType(vt), nb_path=n, nb_steps_per_year=int(g["spy"]),
                                      seed=int(g["seed"]), **p)
    scale = g["forwards"][:, None] if tag != "qvar" else 1.0
    np.testing.assert_allclose(np.stack(pr), g[f"logsv_prices_{tag}"], rtol=1e-11, atol=0)
    np.testing.assert_allclose(np.stack(sd) / scale, g[f"logsv_stderrs_{tag}"] / scale, rtol=1e-11, atol=1e-14)
    if tag != "qvar":
        from stochvolmodels_amd.engine import get_engine
        x, s, q = get_engine(n).get_state()
        np.testing.assert_allclose(np.stack([x[:256], s[:256], q[:256]]), g[f"logsv_states_{tag}"][-1], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("tag", ["btc", "base"])
def test_heston_philox_c4_shape_vs_reference(sv, golden, tag):
    """the same for the reference's heston_mc_chain_pricer (Euler with the floor, 8 x 46 steps) at 2^16 paths"""
    g = golden("philox_c4_shape")
    v0, theta, kappa, rho, volvol = (float(a) for a in g[f"heston_params_{tag}"])
    n = int(g["n_path"])
    pr, sd = sv.heston_mc_chain_pricer(ttms=g["ttms"], forwards=g["forwards"], discfactors=g["discfactors"],
                                       strikes_ttms=tuple(g["strikes"]), optiontypes_ttms=tuple(g["types"]), v0=v0, theta=theta,
                                       kappa=kappa, rho=rho, volvol=volvol, nb_path=n, seed=int(g["seed"]))
    scale = g["forwards"][:, None]
    np.testing.assert_allclose(np.stack(pr) / scale, g[f"heston_prices_{tag}"] / scale, rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(np.stack(sd) / scale, g[f"heston_stderrs_{tag}"] / scale, rtol=1e-11, atol=1e-14)
    from stochvolmodels_amd.engine import get_engine
    x, v, q = get_engine(n).get_state()
    np.testing.assert_allclose(np.stack([x[:256], v[:256], q[:256]]), g[f"heston_states_{tag}"][-1], rtol=1e-10, atol=1e-12)


def test_logsv_reference_test_case(sv, golden):
    """the reference's own fixed-random test (tests/test_logsv_characterization.py:346-458) on the GPU"""
    g = golden("logsv_reference_test_case")
    p = P(g["params"])
    n, nb = int(g["nb_path"]), int(g["nb_steps"])
    rng = np.random.default_rng(123)
    W0 = rng.standard_normal((nb, n))
    W1 = rng.standard_normal((nb, n))
    pr, sd = sv.logsv_mc_chain_pricer_fixed_randoms(
        ttms=np.array([float(g["ttm"])]), forwards=np.array([1.0]), discfactors=np.array([float(g["discfactor"])]),
        strikes_ttms=(g["strikes"],), optiontypes_ttms=(g["types"],), W0s=[W0], W1s=[W1], dts=[float(g["dt"])],
        vol_backbone_etas=np.ones(1), **p)
    np.testing.assert_allclose(pr[0], g["prices"], **ST)
    np.testing.assert_allclose(sd[0], g["stderrs"], **ST)
    assert np.all(np.abs(g["analytic"] - pr[0]) <= 4.0 * sd[0])
    x, s, q = sv.simulate_logsv_x_vol_terminal(ttm=float(g["ttm"]), x0=np.zeros(n), sigma0=np.full(n, p["v0"]),
                                               qvar0=np.zeros(n), theta=p["theta"], kappa1=p["kappa1"],
                                               kappa2=p["kappa2"], beta=p["beta"], volvol=p["volvol"], nb_path=n,
                                               W0=W0, W1=W1, dt=float(g["dt"]))
    np.testing.assert_allclose(x[:256], g["x_head"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(s[:256], g["sigma_head"], rtol=1e-12)
    np.testing.assert_allclose(q[:256], g["qvar_head"], rtol=1e-12)
    assert np.all(np.isfinite(x)) and np.all(s > 0) and np.all(q >= 0)
    ttm = float(g["ttm"])
    assert abs(np.mean(np.exp(x)) - 1.0) <= 4.0 * np.std(np.exp(x), ddof=1) / np.sqrt(n)
    assert abs(np.mean(s) - float(g["expected_sigma"])) <= 4.0 * np.std(s, ddof=1) / np.sqrt(n)
    assert abs(np.mean(q / ttm) - float(g["expected_qvar"])) <= 4.0 * np.std(q / ttm, ddof=1) / np.sqrt(n)


def test_heston_euler_vs_reference(sv, oracle, golden):
    g = golden("heston")
    v0, theta, kappa, rho, volvol = (float(a) for a in g["seed42_params"])
    x, v, q = sv.simulate_heston_x_vol_terminal(ttm=0.05, x0=np.zeros(4), var0=v0 * np.ones(4), qvar0=np.zeros(4),
                                                theta=theta, kappa=kappa, rho=rho, volvol=volvol, nb_path=4,
                                                W0=g["seed42_W0"], W1=g["seed42_W1"], dt=float(g["seed42_dt"]))
    np.testing.assert_allclose(np.stack([x, v, q]), g["seed42_terminal"], rtol=1e-12, atol=1e-15)
    for tag in ("base", "btc"):
        v0, theta, kappa, rho, volvol = (float(a) for a in g[f"params_{tag}"])
        pr, sd = sv.heston_mc_chain_pricer(ttms=g["ttms"], forwards=g["forwards"], discfactors=g["discfactors"],
                                           strikes_ttms=tuple(g["strikes"]), optiontypes_ttms=tuple(g["types"]),
                                           v0=v0, theta=theta, kappa=kappa, rho=rho, volvol=volvol,
                                           nb_path=int(g["n_path"]), seed=int(g["seed"]))
        np.testing.assert_allclose(np.stack(pr), g[f"prices_{tag}"], **ST)
        np.testing.assert_allclose(np.stack(sd), g[f"stderrs_{tag}"], **ST)
        from stochvolmodels_amd.engine import get_engine
        x, v, q = get_engine(int(g["n_path"])).get_state()
        np.testing.assert_allclose(np.stack([x, v, q]), g[f"states_{tag}"][-1], rtol=1e-10, atol=1e-12)
        assert v.min() >= 1e-4 and q.min() >= 0            # floor semantics, reference test :248-266


def test_heston_invariants_small(sv):
    """reference tests/test_heston_characterization.py:248-266"""
    x, v, q = sv.HestonPricer().simulate_terminal_values(sv.HestonParams(v0=0.04, theta=0.05, kappa=2.0, rho=-0.5,
                                                                         volvol=0.4), ttm=0.02, nb_path=256, seed=1)
    assert x.shape == v.shape == q.shape == (256,)
    assert np.all(np.isfinite(x)) and np.all(v >= 1e-4) and np.all(q >= 0)


def test_heston_euler_nan_variance_propagates(sv, oracle):
    """np.maximum(v, 1e-4) keeps a NaN variance (pricers/heston_pricer.py:379); the kernel floors with one v_max_f64, which
    does not, and restores the NaN at the fold: a NaN incoming variance (and NaN constants) must come out NaN, as from
    the CPU twin, and must not touch the other paths."""
    n, nb = 512, 8
    eng = _engine(n)
    x0, q0 = np.zeros(n), np.zeros(n)
    v0 = np.full(n, 0.04)
    v0[[3, 200]] = np.nan
    v0[77] = np.inf
    eng.set_state(x0, v0, q0)
    eng.heston_rng(nb, 0.01, 0.04, 4.0, -0.5, 0.4, 0, 5, 0, 0)
    x, v, q = eng.get_state()
    ox, ov, oq = oracle.heston_terminal_rng(x0, v0, q0, nb, 0.01, 0.04, 4.0, -0.5, 0.4, 5, scheme=oracle.HESTON_EULER_FLOOR)
    bad = np.zeros(n, dtype=bool)
    bad[[3, 77, 200]] = True
    assert np.all(np.isnan(v[bad])) and np.all(np.isnan(ov[bad])) and np.all(np.isnan(x[bad]))
    np.testing.assert_allclose(v[~bad], ov[~bad], rtol=1e-12)
    np.testing.assert_allclose(x[~bad], ox[~bad], rtol=1e-11, atol=1e-12)
    eng.fill_state(0.0, 0.04, 0.0)
    eng.heston_rng(nb, 0.01, float("nan"), 4.0, -0.5, 0.4, 0, 5, 0, 0)         # theta = NaN
    x, v, q = eng.get_state()
    assert np.all(np.isnan(v))
    eng.close()


def test_payoff_vs_reference(sv, golden):
    g = golden("payoff")
    for name in g["names"]:
        ttm, fwd, df, vt = g[f"{name}_scalars"]
        pr, sd = sv.compute_mc_vars_payoff(x0=g[f"{name}_x"], sigma0=np.ones_like(g[f"{name}_x"]),
                                           qvar0=g[f"{name}_qvar"], ttm=ttm, forward=fwd,
                                           strikes_ttm=g[f"{name}_strikes"], optiontypes_ttm=g[f"{name}_types"],
                                           discfactor=df, variable_type=sv.VariableType(int(vt)))
        np.testing.assert_allclose(pr, g[f"{name}_prices"], rtol=1e-12, atol=1e-14, err_msg=str(name))
        np.testing.assert_allclose(sd, g[f"{name}_stderrs"], rtol=1e-12, atol=1e-14, err_msg=str(name))


def test_payoff_reference_known_answers(sv):
    """reference tests/test_numerical_utilities.py:73-141"""
    spots = np.array([0.8, 1.0, 1.2])
    pr, sd = sv.compute_mc_vars_payoff(x0=np.log(spots), sigma0=np.ones(3), qvar0=np.zeros(3), ttm=1.0, forward=1.0,
                                       strikes_ttm=np.ones(4), optiontypes_ttm=np.array(["C", "P", "IC", "IP"]),
                                       discfactor=0.95)
    pay = np.vstack([np.maximum(spots - 1, 0), np.maximum(1 - spots, 0), np.maximum(spots - 1, 0) / spots,
                     np.maximum(1 - spots, 0) / spots])
    np.testing.assert_allclose(pr, 0.95 * pay.mean(axis=1), atol=1e-13)
    np.testing.assert_allclose(sd, 0.95 * pay.std(axis=1) / np.sqrt(3), atol=1e-13)
    x0 = np.log(np.array([0.75, 0.95, 1.05, 1.25]))
    kw = dict(ttm=1.0, forward=1.0, strikes_ttm=np.array([1.0]), optiontypes_ttm=np.array(["C"]))
    p1, s1 = sv.compute_mc_vars_payoff(x0=x0, sigma0=np.ones(4), qvar0=np.zeros(4), **kw)
    x4 = np.tile(x0, 4)
    p4, s4 = sv.compute_mc_vars_payoff(x0=x4, sigma0=np.ones(16), qvar0=np.zeros(16), **kw)
    np.testing.assert_allclose(p4, p1, atol=1e-13)
    np.testing.assert_allclose(s4, s1 / 2.0, atol=1e-13)


def test_payoff_errors(sv):
    z = np.zeros(4)
    with pytest.raises(ValueError, match="payoff"):
        sv.compute_mc_vars_payoff(x0=z, sigma0=z + 1, qvar0=z, ttm=1.0, forward=1.0, strikes_ttm=np.array([1.0]),
                                  optiontypes_ttm=np.array(["BAD"]))
    with pytest.raises(NotImplementedError):
        sv.compute_mc_vars_payoff(x0=z, sigma0=z + 1, qvar0=z, ttm=1.0, forward=1.0, strikes_ttm=np.array([1.0]),
                                  optiontypes_ttm=np.array(["C"]), variable_type=sv.VariableType.SIGMA)
    # the C ABI itself reports the same conditions as status codes
    import ctypes as C
    from stochvolmodels_amd import _lib
    eng = _engine(4)
    k, ty, sh = np.array([1.0]), np.array([7], dtype=np.int8), np.zeros(1)
    ptr, _ = eng.alloc_sums(8, "t")
    pd = C.POINTER(C.c_double)
    rc = eng.lib.svmc_payoff_sums(eng.x.ptr, None, 4, 1.0, 1.0, ptr, k.ctypes.data_as(pd),
                                  ty.ctypes.data_as(C.POINTER(C.c_int8)), sh.ctypes.data_as(pd), 1, 1, ptr + 16,
                                  eng.ws.ptr, eng.ws_bytes, None)
    assert rc == _lib.ERR_UNKNOWN_PAYOFF
    ty[0] = 0
    rc = eng.lib.svmc_payoff_sums(eng.x.ptr, None, 4, 1.0, 1.0, ptr, k.ctypes.data_as(pd),
                                  ty.ctypes.data_as(C.POINTER(C.c_int8)), sh.ctypes.data_as(pd), 1, 3, ptr + 16,
                                  eng.ws.ptr, eng.ws_bytes, None)
    assert rc == _lib.ERR_UNSUPPORTED_VARIABLE
    rc = eng.lib.svmc_payoff_sums(eng.x.ptr, None, 4, 1.0, 1.0, ptr, k.ctypes.data_as(pd),
                                  ty.ctypes.data_as(C.POINTER(C.c_int8)), sh.ctypes.data_as(pd), 1, 1, ptr + 16,
                                  eng.ws.ptr, 8, None)
    assert rc == _lib.ERR_WORKSPACE
    eng.close()


def test_state_length_assertion(sv):
    """reference :1010-1020 asserts on state-vector length"""
    with pytest.raises(AssertionError):
        sv.simulate_logsv_x_vol_terminal(ttm=0.1, x0=np.zeros(3), sigma0=np.ones(5), qvar0=np.zeros(5), theta=1.0,
                                         kappa1=1.0, kappa2=1.0, beta=0.0, volvol=1.0, nb_path=5)


def test_ragged_sizes_and_many_strikes(sv, oracle):
    """n_path not a multiple of the wave/block size, one path, and more strikes than one payoff launch holds"""
    p = sv.LOGSV_BTC_PARAMS
    for n in (1, 63, 257, 1000):
        kk = np.linspace(0.5, 1.5, 37)
        types = np.array((["P", "IP", "C", "IC"] * 10)[:37])
        pr, sd = sv.logsv_mc_chain_pricer(ttms=np.array([0.05]), forwards=np.array([1.0]),
                                          discfactors=np.array([1.0]), strikes_ttms=(kk,), optiontypes_ttms=(types,),
                                          v0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta,
                                          volvol=p.volvol, vol_backbone_etas=np.ones(1), nb_path=n,
                                          nb_steps_per_year=100, seed=5)
        nb, dt, _ = sv.set_time_grid(0.05, 100)
        x, s, q = oracle.logsv_terminal_rng(np.zeros(n), p.sigma0 * np.ones(n), np.zeros(n), nb, dt, p.theta,
                                            p.kappa1, p.kappa2, p.beta, p.volvol, 5)
        opr, osd = oracle.payoff(x, q, 0.05, 1.0, kk, types)
        np.testing.assert_allclose(pr[0], opr, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(sd[0], osd, rtol=1e-12, atol=1e-13)


def test_seed_semantics(sv):
    p = sv.LOGSV_BTC_PARAMS
    pricer = sv.LogSVPricer()
    a = pricer.simulate_terminal_values(p, ttm=0.05, nb_path=512, seed=9)
    b = pricer.simulate_terminal_values(p, ttm=0.05, nb_path=512, seed=9)
    np.testing.assert_array_equal(a[0], b[0])                 # explicit seed replays
    sv.set_seed(123)
    c = pricer.simulate_terminal_values(p, ttm=0.05, nb_path=512)
    d = pricer.simulate_terminal_values(p, ttm=0.05, nb_path=512)
    assert not np.array_equal(c[0], d[0])                     # successive un-seeded calls draw fresh randoms
    sv.set_seed(123)
    e = pricer.simulate_terminal_values(p, ttm=0.05, nb_path=512)
    np.testing.assert_array_equal(c[0], e[0])                 # set_seed rewinds the stream


def test_sharding_invariance(sv):
    """a path range generated with a path offset equals the same range of the full run, bitwise"""
    p = sv.LOGSV_BTC_PARAMS
    n = 4096
    full = _engine(n)
    full.fill_state(0.0, p.sigma0, 0.0)
    full.logsv_rng(40, 1e-3, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, 1.0, True, 31, 0, 7)
    fx, fs, fq = full.get_state()
    for off, m in ((0, 1000), (1000, 3096)):
        part = _engine(m, offset=off)
        part.fill_state(0.0, p.sigma0, 0.0)
        part.logsv_rng(40, 1e-3, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, 1.0, True, 31, 0, 7)
        px, ps, pq = part.get_state()
        np.testing.assert_array_equal(px, fx[off:off + m])
        np.testing.assert_array_equal(ps, fs[off:off + m])
        np.testing.assert_array_equal(pq, fq[off:off + m])
        part.close()
    full.close()


def test_heston_qe(sv, oracle, golden):
    """QE-M: (a) kernel == CPU twin on supplied (Z0, Z1, U) and on the device draw; (b) prices within 4 stderr of the
    reference's analytic Heston prices, both parameter sets of config C3, and the martingale property."""
    from stochvolmodels_amd.engine import DeviceBuffer
    g = golden("analytic")
    n, nb = 4096, 16
    Z0, Z1 = oracle.fill_normals(3, n, nb)
    U = oracle.fill_uniforms(3, n, nb)
    for tag in ("base", "btc"):
        v0, theta, kappa, rho, volvol = (float(a) for a in g[f"heston_{tag}_params"])
        eng = _engine(n)
        eng.fill_state(0.0, v0, 0.0)
        z0, z1, u = eng.upload_randoms((Z0, Z1, U))
        eng.heston_qe_w(nb, 0.25 / nb, theta, kappa, rho, volvol, z0, z1, u)
        x, v, q = eng.get_state()
        ox, ov, oq = oracle.heston_qe_terminal_w(np.zeros(n), v0 * np.ones(n), np.zeros(n), 0.25 / nb, theta, kappa,
                                                 rho, volvol, Z0, Z1, U)
        np.testing.assert_allclose(x, ox, rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(v, ov, rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(q, oq, rtol=1e-12, atol=1e-13)
        # rng route == oracle rng route
        eng.fill_state(0.0, v0, 0.0)
        eng.heston_rng(nb, 0.25 / nb, theta, kappa, rho, volvol, 1, 3, 0, 0)
        x, v, q = eng.get_state()
        ox, ov, oq = oracle.heston_terminal_rng(np.zeros(n), v0 * np.ones(n), np.zeros(n), nb, 0.25 / nb, theta,
                                                kappa, rho, volvol, 3, scheme=oracle.HESTON_QE)
        np.testing.assert_allclose(x, ox, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(v, ov, rtol=1e-11, atol=1e-13)
        eng.close()
        # (b) the scheme against the reference's analytic Heston prices, 4 standard errors, no additive terms: puts
        # without the forward recentring (cases.bounded_put_check -- an honest stderr also for BTC_HESTON_PARAMS, whose
        # E[S_T^2] is infinite around T ~ 1), terminal states taken expiry by expiry from the resident engine
        from cases import bounded_put_check
        nq, spy = 1 << 18, 127
        eng = _engine(nq)
        eng.fill_state(0.0, v0, 0.0)
        t0, step0 = 0.0, 0
        for i, ttm in enumerate(g["ttms"]):
            nb_i, dt_i, _ = sv.set_time_grid(float(ttm) - t0, spy)
            eng.heston_rng(nb_i, dt_i, theta, kappa, rho, volvol, 1, 11, 0, step0)
            x, v, q = eng.get_state()
            diff, sdp = bounded_put_check(x, g["strikes"], g["types"], g[f"heston_{tag}_prices"][i])
            assert np.all(diff <= 4.0 * sdp + 1e-6), (tag, i, diff / sdp)     # 1e-6: deep strikes no sampled path reaches
            if tag == "base":                  # a variance to speak of: the martingale test
                assert abs(np.mean(np.exp(x)) - 1.0) <= 4.0 * np.std(np.exp(x)) / np.sqrt(nq)
            t0, step0 = float(ttm), step0 + nb_i
        assert v.min() >= 0.0
        eng.close()


@pytest.mark.parametrize("tag,par", [
    ("exponential branch (volvol^2 = 17 x 3 kappa theta), rho < 0", dict(v0=0.02, theta=0.02, kappa=1.0, rho=-0.7, volvol=1.0)),
    ("mixed branches, rho > 0 (A > 0: the correction's existence is tested per path)", dict(v0=0.09, theta=0.09, kappa=2.0, rho=0.6, volvol=0.9)),
    ("quadratic only, rho > 0", dict(v0=0.09, theta=0.09, kappa=2.0, rho=0.6, volvol=0.5)),
    ("quadratic only, rho = 0", dict(v0=0.5, theta=0.6, kappa=3.0, rho=0.0, volvol=1.2)),
    # (smaller still and the scheme itself is ill-conditioned: K2 v1 ~ rho / volvol cancels against K0* to O(1), in the twin too)
    ("almost no vol of vol (psi ~ 1e-7: m - alpha is all cancellation)", dict(v0=0.04, theta=0.05, kappa=2.0, rho=-0.3, volvol=1e-5)),
])
def test_heston_qe_branches_the_parameters_decide(sv, oracle, tag, par):
    """the QE step skips what the parameters alone decide (svmc_models.h make_qe_consts: quad_only when volvol^2 <= 3 kappa theta,
    e_below_one when A <= 0): every combination -- never / sometimes / mostly exponential, rho of either sign and zero, vanishing
    vol of vol -- path by path against the CPU twin's textbook form on the same stream, one-slice kernel and chain kernel (few-waves
    and full-launch forms: 460 000 paths are past the seven-waves-per-SIMD switch), states and prices.  The two quadratic-only
    sets with rho <= 0 run the kernels COMPILED without the exponential branch (HESTON_QE_QUAD), the others the general ones"""
    seed = 17
    for n in (8192, 140_000, 460_000):
        nb = 48
        eng = _engine(n)
        eng.fill_state(0.0, par["v0"], 0.0)
        eng.heston_rng(nb, 0.5 / nb, par["theta"], par["kappa"], par["rho"], par["volvol"], 1, seed, 0, 0)
        x, v, q = eng.get_state()
        ox, ov, oq = oracle.heston_terminal_rng(np.zeros(n), par["v0"] * np.ones(n), np.zeros(n), nb, 0.5 / nb, par["theta"],
                                                par["kappa"], par["rho"], par["volvol"], seed, scheme=oracle.HESTON_QE)
        if "exponential" in tag or "mixed" in tag:
            assert np.mean(ov == 0.0) > 0 or np.min(ov) < 1e-3 * par["theta"]        # the exponential branch does run here
        np.testing.assert_allclose(x, ox, rtol=1e-9, atol=1e-10, err_msg=tag)
        np.testing.assert_allclose(v, ov, rtol=1e-9, atol=1e-12, err_msg=tag)
        np.testing.assert_allclose(q, oq, rtol=1e-10, atol=1e-13, err_msg=tag)
        assert np.array_equal(v == 0.0, ov == 0.0)                                    # the same paths sit at exactly zero
        eng.close()
    # the chain kernel through the pricer
    ttms = np.array([0.2, 0.5])
    k = np.array([0.85, 1.0, 1.15])
    ty = np.array(["P", "C", "C"])
    n = 20_000
    pr, sd = sv.heston_mc_chain_pricer(ttms=ttms, forwards=np.ones(2), discfactors=np.ones(2), strikes_ttms=(k, k), optiontypes_ttms=(ty, ty),
                                       nb_path=n, scheme="qe", nb_steps_per_year=100, seed=seed, **par)
    x, v, q = np.zeros(n), par["v0"] * np.ones(n), np.zeros(n)
    t0, step0 = 0.0, 0
    for i, ttm in enumerate(ttms):
        nb, dt, _ = sv.set_time_grid(ttm - t0, 100)
        x, v, q = oracle.heston_terminal_rng(x, v, q, nb, dt, par["theta"], par["kappa"], par["rho"], par["volvol"], seed,
                                             scheme=oracle.HESTON_QE, step_offset=step0)
        step0, t0 = step0 + nb, ttm
        opr, osd = oracle.payoff(x, q, float(ttm), 1.0, k, ty, 1.0)
        np.testing.assert_allclose(pr[i], opr, rtol=1e-9, atol=1e-12, err_msg=tag)
        np.testing.assert_allclose(sd[i], osd, rtol=1e-9, atol=1e-12, err_msg=tag)


def test_config_c1_heston_10k_100(sv, oracle):
    """BASELINE config 1: Heston Euler, 10k paths x 100 steps (ttm=1, spy=99), 5 quickstart strikes"""
    n, seed = 10_000, 20240601
    kk = np.array([0.8, 0.9, 1.0, 1.1, 1.2])
    types = np.array(["P", "P", "C", "C", "C"])
    pr, sd = sv.heston_mc_chain_pricer(ttms=np.array([1.0]), forwards=np.ones(1), discfactors=np.ones(1),
                                       strikes_ttms=(kk,), optiontypes_ttms=(types,), v0=0.04, theta=0.04, kappa=4.0,
                                       rho=-0.5, volvol=0.4, nb_path=n, nb_steps_per_year=99, seed=seed)
    nb, dt, _ = sv.set_time_grid(1.0, 99)
    assert nb == 100
    x, v, q = oracle.heston_terminal_rng(np.zeros(n), 0.04 * np.ones(n), np.zeros(n), nb, dt, 0.04, 4.0, -0.5, 0.4,
                                         seed)
    opr, osd = oracle.payoff(x, q, 1.0, 1.0, kk, types)
    np.testing.assert_allclose(pr[0], opr, **ST)
    np.testing.assert_allclose(sd[0], osd, rtol=1e-12)


def test_config_c2_full_size_properties(sv, golden):
    """BASELINE config 2 at full size (2^20 paths x 1024 steps, 21 strikes): size-independent properties --
    martingale, put-call parity of the recentred slice, determinism."""
    p = sv.LOGSV_BTC_PARAMS
    n = 1 << 20
    kk = np.linspace(0.5, 1.5, 21)
    chain = sv.OptionChain.slice_to_chain(ttm=1.0, forward=1.0, strikes=np.concatenate([kk, kk]),
                                          optiontypes=np.array(["C"] * 21 + ["P"] * 21))
    pricer = sv.LogSVPricer()
    pr, sd = pricer.model_mc_price_chain(chain, p, nb_path=n, nb_steps=1023, seed=20240602)
    call, put = pr[0][:21], pr[0][21:]
    np.testing.assert_allclose(call - put, 1.0 - kk, rtol=0, atol=1e-14)        # recentring => parity is exact
    pr2, sd2 = pricer.model_mc_price_chain(chain, p, nb_path=n, nb_steps=1023, seed=20240602)
    np.testing.assert_array_equal(pr[0], pr2[0])                        # deterministic reductions
    np.testing.assert_array_equal(sd[0], sd2[0])
    from stochvolmodels_amd.engine import get_engine
    x, s, q = get_engine(n).get_state()
    assert np.all(np.isfinite(x)) and np.all(s > 0) and np.all(q >= 0)
    assert abs(np.mean(np.exp(x)) - 1.0) <= 4.0 * np.std(np.exp(x)) / np.sqrt(n)
    assert np.all(sd[0] > 0.0)
    # (path-wise agreement with the CPU oracle on this configuration, and north_star's 2-stderr criterion:
    # tests/test_gpu_fullsize.py; analytic-vs-MC at this size measures the expansion's bias: tools/c5_bias.py)


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_share_one_gpu(oracle, tmp_path, world):
    """the multi-process path on real kernels: `world` ranks (gloo, all on this GPU) shard one path set through
    the product drivers; every rank must return the single-process oracle result (tests/test_dist_gloo.py does
    the same on CPU with an engine double)."""
    import os
    import subprocess
    import sys
    from test_dist_gloo import _expected
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29711 + world), WORLD_SIZE=str(world),
               SVMC_DIST_BACKEND="gloo", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "gpu_dist_worker.py"), out],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    exp = _expected(oracle)
    for r in range(world):
        got = np.load(out + f".rank{r}.npz")
        for key in got.files:
            np.testing.assert_allclose(got[key], exp[key], rtol=1e-12, atol=1e-12, err_msg=f"{key} rank {r}/{world}")


def test_rccl_group_on_one_rank(oracle, tmp_path):
    """the RCCL leg of the N>1 path on a 1-GPU box: a lone rank builds the "nccl" process group
    (SVMC_DIST_SINGLE_RANK_GROUP=1) and prices through TorchComm -- torch-owned reduction buffers written by
    libsvmc's kernels, all-reduces over RCCL ordered by the stream alone -- and must return the single-process
    oracle result; a second run with the host synchronisations forced must agree bit for bit."""
    import os
    import subprocess
    import sys
    from test_dist_gloo import _expected
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = _expected(oracle)
    runs = []
    for tag, strict in (("ordered", "0"), ("strict", "1")):
        out = str(tmp_path / tag)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
                   SVMC_DIST_SINGLE_RANK_GROUP="1", SVMC_EXPECT_BACKEND="nccl", SVMC_DIST_STRICT_SYNC=strict,
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        env.pop("SVMC_DIST_BACKEND", None)
        p = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_dist_worker.py"), out], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode()
        got = np.load(out + ".rank0.npz")
        for key in got.files:
            np.testing.assert_allclose(got[key], exp[key], rtol=1e-12, atol=1e-12, err_msg=f"{key} ({tag})")
        runs.append({k: got[k] for k in got.files})
    for key in runs[0]:
        np.testing.assert_array_equal(runs[0][key], runs[1][key], err_msg=key)


def _run_dist_workers(tmp_path, tag, world, port, extra_env, timeout=300):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / tag)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), **extra_env)
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "gpu_dist_worker.py"), out],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=timeout)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            logs.append(p.communicate()[0].decode() + "\n[timeout]")
    ok = all(p.returncode == 0 for p in procs)
    return ok, "\n".join(logs), [out + f".rank{r}.npz" for r in range(world)]


def test_rccl_through_the_c_abi_one_rank(oracle, tmp_path):
    """RCCL below Python: a lone rank prices through RcclComm -- libsvmc's own svmc_rccl_* entry points, plain device
    buffers, collectives issued by libsvmc on the engine's stream -- and must return the single-process oracle
    result, bit-identical to the torch.distributed route (test_rccl_group_on_one_rank)."""
    from test_dist_gloo import _expected
    exp = _expected(oracle)
    ok, log, files = _run_dist_workers(tmp_path, "cabi", 1, 29741, dict(
        SVMC_DIST_SINGLE_RANK_GROUP="1", SVMC_DIST_COMM="rccl", SVMC_EXPECT_BACKEND="gloo", SVMC_EXPECT_COMM="RcclComm"))
    assert ok, log
    got = np.load(files[0])
    for key in got.files:
        np.testing.assert_allclose(got[key], exp[key], rtol=1e-12, atol=1e-12, err_msg=key)
    ok, log, files2 = _run_dist_workers(tmp_path, "torch", 1, 29742, dict(
        SVMC_DIST_SINGLE_RANK_GROUP="1", SVMC_EXPECT_BACKEND="nccl"))
    assert ok, log
    ref = np.load(files2[0])
    for key in got.files:
        np.testing.assert_array_equal(got[key], ref[key], err_msg=key)


@pytest.mark.parametrize("route", ["rccl", "torch"])
def test_two_rccl_ranks_on_one_gpu(oracle, tmp_path, route):
    """a REAL two-rank RCCL communicator, both ranks on this box's single GPU, through either route (libsvmc's C ABI /
    torch.distributed "nccl").  RCCL may refuse two ranks on one device ("Duplicate GPU detected"): then there is
    nothing to test on a 1-GPU box and the case is skipped; where it is allowed, both ranks must return the
    single-process oracle result and the stream-ordered and host-synchronised runs must agree bit for bit."""
    from test_dist_gloo import _expected
    exp = _expected(oracle)
    runs = []
    for k, strict in enumerate(("0", "1")):
        extra = dict(SVMC_DIST_STRICT_SYNC=strict)
        extra.update(dict(SVMC_DIST_COMM="rccl", SVMC_EXPECT_COMM="RcclComm") if route == "rccl"
                     else dict(SVMC_EXPECT_BACKEND="nccl"))
        ok, log, files = _run_dist_workers(tmp_path, f"{route}{strict}", 2, 29751 + 2 * k + (10 if route == "torch" else 0),
                                           extra, timeout=180)
        if not ok:
            low = log.lower()
            if "duplicate gpu" in low or "invalid usage" in low or "[timeout]" in low or "nccl" in low or "rccl" in low:
                pytest.skip("RCCL does not run two ranks on one device here: " + log.strip().splitlines()[-1][:200])
            assert ok, log
        got = [np.load(f) for f in files]
        for r in range(2):
            for key in got[r].files:
                np.testing.assert_allclose(got[r][key], exp[key], rtol=1e-9, atol=1e-12, err_msg=f"{key} rank {r}")
        runs.append({k_: got[0][k_] for k_ in got[0].files})
        if route == "rccl":
            break                                # the C-ABI route is always stream-ordered: one run
    if len(runs) == 2:
        for key in runs[0]:
            np.testing.assert_array_equal(runs[0][key], runs[1][key], err_msg=key)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_c_abi_chain_drivers_sharded_on_one_gpu(sv, world):
    """The sharding logic of the fused C drivers with world > 1 (csrc/svmc_chain.hip: global path offsets into the
    counter-based randoms, the two sum all-reduces between the kernels, the finalisation by the job's path count) --
    RCCL refuses two ranks on one device, so the all-reduce is handed to the caller (svmc_session_set_reducer): `world`
    sessions of THIS process, one thread each, hold the shards of an odd-sized job on the one GPU there is and sum their
    reduction buffers through host memory.  Every rank must return the job's prices, equal to the unsharded session's to
    reduction-order rounding, for svmc_logsv_chain_price (LOG_RETURN with inverse options, and Q_VAR) and
    svmc_heston_chain_price (Euler and QE)."""
    import ctypes as C
    import threading
    from stochvolmodels_amd import _lib
    from stochvolmodels_amd.dist import shard_range
    from stochvolmodels_amd.engine import option_type_codes
    L = _lib.load()
    dp, pi8, psz = C.POINTER(C.c_double), C.POINTER(C.c_int8), C.POINTER(C.c_size_t)
    n_total = 6007
    ttms = np.array([0.1, 0.25, 0.4])
    fw = 100.0 * np.exp(0.03 * ttms)
    df = np.exp(-0.03 * ttms)
    etas = np.array([1.0, 1.05, 0.95])
    P = sv.LOGSV_BTC_PARAMS

    def chains(kind):
        if kind == "qvar":
            strikes = [np.array([0.02, 0.05, 0.1]), np.array([0.05, 0.15]), np.array([0.1, 0.2, 0.3, 0.4])]
            types = [np.array(["C", "P", "C"]), np.array(["P", "C"]), np.array(["C", "C", "P", "C"])]
        else:
            strikes = [f * np.array([0.8, 1.0, 1.2]) for f in fw]
            types = [np.array(["P", "IC", "C"]), np.array(["IP", "C", "IC"]), np.array(["P", "C", "C"])]
        offs = np.concatenate([[0], np.cumsum([len(k) for k in strikes])]).astype(np.uintp)
        return (np.concatenate(strikes), np.concatenate([option_type_codes(t) for t in types]).astype(np.int8), offs)

    def price(sess, model, kind, seed):
        k_all, c_all, offs = chains(kind)
        total = int(offs[-1])
        prices, stderrs = np.empty(total), np.empty(total)
        vt = 2 if kind == "qvar" else 1
        a = lambda v: np.ascontiguousarray(v, dtype=np.float64).ctypes.data_as(dp)      # noqa: E731
        if model == "logsv":
            rc = L.svmc_logsv_chain_price(sess, a(ttms), a(fw), a(df), a(etas), 3, a(k_all), c_all.ctypes.data_as(pi8),
                                          offs.ctypes.data_as(psz), P.sigma0, P.theta, P.kappa1, P.kappa2, P.beta, P.volvol,
                                          1, 200, vt, seed, 5, prices.ctypes.data_as(dp), stderrs.ctypes.data_as(dp))
        else:
            rc = L.svmc_heston_chain_price(sess, a(ttms), a(fw), a(df), 3, a(k_all), c_all.ctypes.data_as(pi8),
                                           offs.ctypes.data_as(psz), 0.04, 0.05, 2.0, -0.5, 0.4, 0 if model == "heston_euler" else 1,
                                           200, vt, seed, 5, prices.ctypes.data_as(dp), stderrs.ctypes.data_as(dp))
        _lib.check(rc)
        return prices, stderrs

    def new_session(n):
        sess = C.c_void_p()
        _lib.check(L.svmc_session_create(C.byref(sess), n, 3, 16))
        return sess

    one = new_session(n_total)
    shards = [shard_range(n_total, r, world) for r in range(world)]
    assert sum(n for _, n in shards) == n_total and shards[-1][0] > 0
    sessions = [new_session(n) for _, n in shards]
    barrier = threading.Barrier(world)
    staged = [None] * world
    calls = [0] * world

    def make_reducer(r):
        def reduce(user, buf, n, stream):
            try:
                host = np.empty(n)
                if L.svmc_stream_synchronize(stream) or L.svmc_memcpy_d2h(host.ctypes.data, buf, 8 * n, stream) or \
                        L.svmc_stream_synchronize(stream):
                    return 2
                staged[r] = host
                barrier.wait(timeout=60)
                total = staged[0].copy()
                for other in staged[1:]:                     # fixed rank order on every rank: identical sums everywhere
                    total += other
                barrier.wait(timeout=60)
                calls[r] += 1
                if L.svmc_memcpy_h2d(buf, total.ctypes.data, 8 * n, stream) or L.svmc_stream_synchronize(stream):
                    return 2
                return 0
            except Exception:                                # never raise through the C frame
                return 2
        return _lib.ALL_REDUCE_FN(reduce)

    reducers = [make_reducer(r) for r in range(world)]       # kept alive for the duration of the calls
    for r, (sess, (off, n)) in enumerate(zip(sessions, shards)):
        _lib.check(L.svmc_session_set_reducer(sess, reducers[r], None, r, world, n_total, off))
    try:
        for model, kind in (("logsv", "inverse"), ("logsv", "qvar"), ("heston_euler", "plain"), ("heston_qe", "qvar")):
            ref_p, ref_e = price(one, model, kind, 424242)
            out, errs = [None] * world, []

            def run(r):
                try:
                    out[r] = price(sessions[r], model, kind, 424242)
                except Exception as exc:                     # noqa: BLE001
                    errs.append(exc)
                    barrier.abort()
            before = list(calls)
            threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
            for t in threads:
                t.start()
            for t in threads:
                t.join(timeout=120)
            assert not errs, errs
            assert [c - b for c, b in zip(calls, before)] == [2] * world          # exactly the two all-reduces per chain
            worst = 0.0
            for r in range(world):
                np.testing.assert_array_equal(out[r][0], out[0][0])               # every rank holds the job's prices
                np.testing.assert_array_equal(out[r][1], out[0][1])
                np.testing.assert_allclose(out[r][0], ref_p, rtol=1e-12, atol=1e-14)
                np.testing.assert_allclose(out[r][1], ref_e, rtol=1e-12, atol=1e-14)
                nz = ref_p != 0.0                                                 # deep out-of-the-money strikes price to 0
                worst = max(worst, float(np.nanmax(np.abs(out[r][0][nz] / ref_p[nz] - 1.0))))
                assert np.all(out[r][0][~nz] == 0.0)
            print(f"C drivers, {world} shards on one GPU, {model}/{kind}: max relative deviation from the unsharded session {worst:.2e}")
        # detaching gives a single-GPU session again
        _lib.check(L.svmc_session_set_reducer(sessions[0], _lib.ALL_REDUCE_FN(0), None, 0, 1, 0, 0))
    finally:
        for sess in sessions + [one]:
            L.svmc_session_destroy(sess)


def test_c_host_rccl_example(sv, tmp_path):
    """examples/price_chain_rccl.c: a plain-C host drives the multi-GPU path -- its own RCCL communicator, the fused
    chain drivers issuing the two all-reduces (svmc_session_set_comm).  One rank with a real communicator must match
    the Python host bit for bit; two ranks on this GPU (where RCCL allows it) must agree with each other and with the
    one-rank result to reduction-order rounding."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "price_chain_rccl")
    libdir = os.path.join(root, "stochvolmodels_amd")
    subprocess.run(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "price_chain_rccl.c"),
                    "-o", exe, "-L" + libdir, "-lsvmc", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-lm"],
                   check=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    n, seed = 65536, 123
    def last_json(text):                       # RCCL / libdrm may print notices of their own ahead of the result line
        return json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])

    run = subprocess.run([exe, "1", "0", str(tmp_path / "id1"), str(n), str(seed)], capture_output=True, text=True, env=env,
                         timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    one = last_json(run.stdout)
    P_ = sv.LOGSV_BTC_PARAMS
    ttms, fw, df = np.array([0.1, 0.25]), np.array([1.0, 1.01]), np.array([0.99, 0.98])
    kk = np.array([0.8, 1.0, 1.2])
    strikes = (kk, 1.01 * kk)
    types = (np.array(["P", "C", "C"]), np.array(["IP", "IC", "C"]))
    pr, sd = sv.logsv_mc_chain_pricer(ttms=ttms, forwards=fw, discfactors=df, strikes_ttms=strikes, optiontypes_ttms=types,
                                      v0=P_.sigma0, theta=P_.theta, kappa1=P_.kappa1, kappa2=P_.kappa2, beta=P_.beta,
                                      volvol=P_.volvol, vol_backbone_etas=np.ones(2), nb_path=n, nb_steps_per_year=120,
                                      seed=seed)
    np.testing.assert_array_equal(np.concatenate(pr), one["logsv_prices"])
    np.testing.assert_array_equal(np.concatenate(sd), one["logsv_stderrs"])
    pr, sd = sv.heston_mc_chain_pricer(ttms=ttms, forwards=fw, discfactors=df, strikes_ttms=strikes, optiontypes_ttms=types,
                                       v0=0.04, theta=0.04, kappa=4.0, rho=-0.5, volvol=0.4, nb_path=n, scheme="qe",
                                       seed=seed)
    np.testing.assert_array_equal(np.concatenate(pr), one["heston_qe_prices"])
    np.testing.assert_array_equal(np.concatenate(sd), one["heston_qe_stderrs"])
    # two ranks, one GPU
    idf = str(tmp_path / "id2")
    procs = [subprocess.Popen([exe, "2", str(r), idf, str(n), str(seed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=120))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.skip("two RCCL ranks on one device did not complete here")
    if any(p.returncode != 0 for p in procs):
        pytest.skip("RCCL does not run two ranks on one device here: " + outs[0][1].strip()[-200:] + outs[1][1].strip()[-200:])
    two = [last_json(o[0]) for o in outs]
    for key in ("logsv_prices", "logsv_stderrs", "heston_qe_prices", "heston_qe_stderrs"):
        np.testing.assert_array_equal(two[0][key], two[1][key], err_msg=key)          # every rank: the job's result
        np.testing.assert_allclose(two[0][key], one[key], rtol=1e-12, atol=1e-15, err_msg=key)


def test_vol_paths(sv, oracle, golden):
    """simulate_vol_paths: reference outputs on supplied brownians; on-device draw vs the CPU twin; the reference's
    own shape / first-row / measure checks (tests/test_logsv_characterization.py:638-673)"""
    g = golden("vol_paths")
    p = P(g["params"])
    for tag, spot in (("spot", True), ("inv", False)):
        sig, grid = sv.simulate_vol_paths(ttm=float(g["ttm"]), nb_path=int(g["n_path"]), nb_steps_per_year=int(g["spy"]),
                                          brownians=g["brownians"], is_spot_measure=spot, **p)
        np.testing.assert_allclose(sig, g[f"sigma_{tag}"], rtol=1e-12)
        np.testing.assert_array_equal(grid, g["grid"])
    t = P(g["test_params"])
    pricer = sv.LogSVPricer()
    params = sv.LogSvParams(sigma0=t["v0"], theta=t["theta"], kappa1=t["kappa1"], kappa2=t["kappa2"], beta=t["beta"],
                            volvol=t["volvol"])
    spot_paths, grid = sv.simulate_vol_paths(ttm=0.02, nb_path=4, nb_steps_per_year=360, brownians=np.zeros((8, 4)), **t)
    inv_paths, _ = sv.simulate_vol_paths(ttm=0.02, nb_path=4, nb_steps_per_year=360, brownians=np.zeros((8, 4)),
                                         is_spot_measure=False, **t)
    assert spot_paths.shape == inv_paths.shape == (9, 4)
    np.testing.assert_allclose(spot_paths, g["test_sigma_zero"], rtol=1e-12)
    np.testing.assert_array_equal(spot_paths[0], t["v0"])
    assert np.all(spot_paths > 0) and not np.array_equal(spot_paths[-1], inv_paths[-1])
    sig, grid = pricer.simulate_vol_paths(params, ttm=0.05, nb_path=1000, seed=12)        # nb_steps -> ceil(360*0.05) = 18 /yr
    nb, dt, _ = sv.set_time_grid(0.05, 18)
    osig = oracle.logsv_vol_paths(nb, dt, t["v0"], t["theta"], t["kappa1"], t["kappa2"], t["beta"], t["volvol"], 1000,
                                  seed=12)
    assert sig.shape == (nb + 1, 1000)
    np.testing.assert_allclose(sig, osig, rtol=1e-12)


# ---- analytic side (row a11): libsvmc's Fourier kernels ------------------------------------------------------------
def test_vol_paths_resident_array_moments_and_pipelined_download(sv):
    """the bulk-output forms of simulate_vol_paths: return_device=True leaves the array in HBM (engine.DeviceArray) -- its
    .numpy() is the default call's array bit for bit, torch takes it zero-copy through __cuda_array_interface__ and DLPack,
    row_moments() / expanding_mean_of_squares() equal what the reference's scripts compute from the host array
    (papers/.../moments_vol_qvar.py:48, :98-104) -- and the NumPy return goes through the pinned pipeline (a 70 MB result,
    several chunks and a ragged tail; into a caller's array with out=)"""
    import pandas as pd
    from stochvolmodels_amd import engine
    p = sv.LogSvParams(sigma0=1.0, theta=1.0, kappa1=4.0, kappa2=4.0, beta=0.0, volvol=1.75)
    pricer = sv.LogSVPricer()
    n, ttm = 20_011, 1.2
    sig, grid = pricer.simulate_vol_paths(p, ttm=ttm, nb_path=n, seed=21)
    assert sig.shape == (520, n) and sig.nbytes > engine.PIPELINE_CHUNK_BYTES * 2          # 83 MB: the pipeline, 3 chunks (nb_steps is per year: int(1.2 x 432) + 1)
    dev, grid2 = pricer.simulate_vol_paths(p, ttm=ttm, nb_path=n, seed=21, return_device=True)
    assert isinstance(dev, engine.DeviceArray) and dev.shape == sig.shape
    np.testing.assert_array_equal(grid, grid2)
    np.testing.assert_array_equal(dev.numpy(), sig)
    out = np.full(sig.shape, np.nan)
    got, _ = pricer.simulate_vol_paths(p, ttm=ttm, nb_path=n, seed=21, out=out)
    assert got is out or np.shares_memory(got, out)
    np.testing.assert_array_equal(out, sig)
    with pytest.raises(ValueError):
        pricer.simulate_vol_paths(p, ttm=ttm, nb_path=n, seed=21, out=np.empty((5, 5)))
    import torch
    t = torch.as_tensor(dev, device="cuda")
    assert t.shape == sig.shape and t.dtype == torch.float64 and t.data_ptr() == dev.ptr
    np.testing.assert_array_equal(t.cpu().numpy(), sig)
    t2 = torch.from_dlpack(dev)
    assert t2.data_ptr() == dev.ptr and torch.equal(t, t2)
    del t, t2
    # the reductions of the reference's scripts, on the device
    mean, std = dev.row_moments(center=p.theta, n_moments=4)
    for k in range(4):
        m_k = np.power(sig - p.theta, k + 1)
        np.testing.assert_allclose(mean[:, k], np.mean(m_k, axis=1), rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(std[:, k], np.std(m_k, axis=1), rtol=1e-9, atol=1e-12)
    q = dev.expanding_mean_of_squares()
    want_q = pd.DataFrame(np.square(sig)).expanding().mean().to_numpy()
    np.testing.assert_allclose(q.numpy(), want_q, rtol=1e-13)
    q.free()
    mom = pricer.vol_path_moments(p, ttm=ttm, nb_path=n, seed=21, n_terms=4, with_qvar=True)
    np.testing.assert_array_equal(mom["mean"], mean)
    np.testing.assert_array_equal(mom["std"], std)
    np.testing.assert_allclose(mom["qvar_mean"], np.mean(want_q, axis=1), rtol=1e-12)
    np.testing.assert_allclose(mom["qvar_std"], np.std(want_q, axis=1), rtol=1e-9, atol=1e-13)
    np.testing.assert_array_equal(mom["grid_t"], grid)
    dev.free()
    with pytest.raises(Exception):
        dev.numpy()
    # a state download big enough for the pipeline (2^22 paths: 3 x 33 MB) equals the state a small engine sees for the same paths
    x, s, qv = pricer.simulate_terminal_values(p, ttm=0.1, nb_path=1 << 22, seed=8)
    assert x.shape == (1 << 22,) and np.all(np.isfinite(x)) and np.all(s > 0)
    from stochvolmodels_amd.engine import pipelined_download, get_engine
    eng = get_engine(1 << 22)
    np.testing.assert_array_equal(pipelined_download(eng.x.ptr, 1 << 22), x)
    np.testing.assert_array_equal(pipelined_download(eng.x.ptr + 8 * 12345, 999_983), x[12345:12345 + 999_983])


@pytest.mark.parametrize("n", [4098, 4100, 1026, 130])
def test_vol_path_reducers_on_even_and_misaligned_layouts(sv, n):
    """the two streaming reducers on layouts that take their 16-byte forms: an even path count (expanding mean: two columns per
    lane) whose quarter-row segments start on odd columns (row sums: the segment's first element peeled to reach a 16-byte
    boundary) and end with an odd element -- against NumPy / pandas on the host copy of the same array"""
    import pandas as pd
    p = sv.LogSvParams(sigma0=0.9, theta=1.0, kappa1=3.0, kappa2=3.0, beta=0.1, volvol=1.4)
    dev, _ = sv.LogSVPricer().simulate_vol_paths(p, ttm=0.3, nb_path=n, nb_steps=77, seed=3, return_device=True)
    sig = dev.numpy()
    mean, std = dev.row_moments(center=p.theta, n_moments=4)
    for k in range(4):
        m_k = np.power(sig - p.theta, k + 1)
        np.testing.assert_allclose(mean[:, k], np.mean(m_k, axis=1), rtol=1e-11, atol=1e-13)
        # (row 0 is the common start value: its exact standard deviation is 0, and E[d^2] - E[d]^2 leaves the rounding of two
        # equal numbers under the square root -- 1e-10 -- whichever kernel formed the sums)
        np.testing.assert_allclose(std[1:, k], np.std(m_k, axis=1)[1:], rtol=1e-9, atol=1e-12)
        assert np.all(std[0] < 1e-8)
    q = dev.expanding_mean_of_squares()
    want = pd.DataFrame(np.square(sig)).expanding().mean().to_numpy()
    np.testing.assert_allclose(q.numpy(), want, rtol=1e-13)
    q.free()
    dev.free()


def test_vol_paths_ragged_sizes(sv, oracle):
    """simulate_vol_paths on every tail of its loops: 1..9 and 18 steps (the drawing loop runs call by call, four steps each;
    the supplied-brownians loop prefetches groups of four) x path counts around the wave size, both measures, device draw
    and supplied increments, against the CPU twin"""
    t = dict(v0=0.3, theta=0.25, kappa1=2.0, kappa2=3.0, beta=-0.4, volvol=0.9)
    rng = np.random.default_rng(5)
    worst = 0.0
    for n in (1, 63, 64, 65, 1000):
        for nb in (1, 2, 3, 4, 5, 6, 7, 8, 9, 18):
            spy, ttm = nb - 1 if nb > 1 else 1, 1.0 if nb > 1 else 0.5           # int(ttm * spy) + 1 = nb
            nb_, dt, _ = sv.set_time_grid(ttm, spy)
            assert nb_ == nb
            for spot in (True, False):
                sig, grid = sv.simulate_vol_paths(ttm=ttm, nb_path=n, nb_steps_per_year=spy, is_spot_measure=spot, seed=3 + nb, **t)
                ref = oracle.logsv_vol_paths(nb, dt, t["v0"], t["theta"], t["kappa1"], t["kappa2"], t["beta"], t["volvol"], n,
                                             is_spot_measure=spot, seed=3 + nb)
                assert sig.shape == (nb + 1, n) and grid.shape == (nb + 1,)
                np.testing.assert_allclose(sig, ref, rtol=1e-12)
                worst = max(worst, float(np.max(np.abs(sig / ref - 1.0))))
                w = np.sqrt(dt) * rng.standard_normal((nb, n))
                sig, _ = sv.simulate_vol_paths(ttm=ttm, nb_path=n, nb_steps_per_year=spy, is_spot_measure=spot, brownians=w, **t)
                ref = oracle.logsv_vol_paths(nb, dt, t["v0"], t["theta"], t["kappa1"], t["kappa2"], t["beta"], t["volvol"], n,
                                             is_spot_measure=spot, brownians=w)
                np.testing.assert_allclose(sig, ref, rtol=1e-12)
                worst = max(worst, float(np.max(np.abs(sig / ref - 1.0))))
    print(f"vol paths, ragged sizes: largest relative deviation from the CPU twin {worst:.2e}")


def test_analytic_logsv_chain(sv, oracle, golden):
    """GPU analytic LogSV chain vs the reference with its ODE solver tightened (1e-9), vs the reference as shipped
    (2e-6 = its RK45 default error), the quickstart goldens, and the CPU twin"""
    g = golden("analytic_tight")
    strikes = tuple(g["strikes"])
    for tag in ("btc", "test"):
        v = [float(a) for a in g[f"{tag}_params"]]
        params = sv.LogSvParams(sigma0=v[0], theta=v[1], kappa1=v[2], kappa2=v[3], beta=v[4], volvol=v[5])
        for mtag, spot, ty in (("spot", True, g["types"]), ("inv", False, g["inv_types"])):
            pr = sv.logsv_chain_pricer(params=params, ttms=g["ttms"], forwards=g["forwards"],
                                       discfactors=g["discfactors"], strikes_ttms=strikes, optiontypes_ttms=tuple(ty),
                                       is_spot_measure=spot)
            np.testing.assert_allclose(np.stack(pr), g[f"{tag}_{mtag}_prices"], rtol=0, atol=1e-12)
        pr = sv.logsv_chain_pricer(params=params, ttms=g["ttms"], forwards=g["forwards"], discfactors=g["discfactors"],
                                   strikes_ttms=strikes, optiontypes_ttms=tuple(g["types"]),
                                   expansion_order=sv.ExpansionOrder.FIRST)
        np.testing.assert_allclose(np.stack(pr), g[f"{tag}_first_order_prices"], rtol=0, atol=1e-12)
    # raw coefficients with slice-to-slice carry and a vol backbone
    b = [float(a) for a in g["btc_params"]]
    z = np.zeros(13, dtype=np.complex128)
    kw = dict(phi_grid=g["mgf_phi"], psi_grid=z, theta_grid=z, sigma0=b[0], theta=b[1], kappa1=b[2], kappa2=b[3],
              beta=b[4], volvol=b[5])
    a1, lm1 = sv.compute_logsv_a_mgf_grid(ttm=0.3, vol_backbone_eta=0.9, **kw)
    a2, lm2 = sv.compute_logsv_a_mgf_grid(ttm=0.2, a_t0=a1, vol_backbone_eta=1.1, **kw)
    np.testing.assert_allclose(a1, g["mgf_a1"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(a2, g["mgf_a2"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(lm2, g["mgf_lm2"], rtol=1e-10, atol=1e-11)
    # quickstart goldens through the class API
    chain = sv.OptionChain.get_uniform_chain(ttms=np.array([0.25, 0.5]), ids=np.array(["3m", "6m"]),
                                             forwards=np.array([1.0, 1.0]), strikes=np.array([0.8, 0.9, 1.0, 1.1, 1.2]))
    q = sv.LogSvParams(sigma0=1.0, theta=1.0, kappa1=5.0, kappa2=5.0, beta=0.2, volvol=2.0)
    pr = sv.LogSVPricer().price_chain(chain, q)
    np.testing.assert_allclose(pr[0][2], 0.197331, rtol=5e-6, atol=1e-8)
    np.testing.assert_allclose(pr[1][2], 0.275202, rtol=5e-6, atol=1e-8)
    np.testing.assert_allclose(np.stack(pr), g["quick_chain_prices"], rtol=0, atol=2e-6)


def test_analytic_heston_and_c5_sweep(sv, golden):
    """Heston closed form vs the reference (1e-12); the analytic side of config C5: for the 5 LogSV parameter sets the GPU
    analytic chain (4 expiries x 21 strikes) reproduces the reference's analytic chain to the reference's solver error"""
    g = golden("analytic")
    kk, types, ttms = g["strikes"], g["types"], g["ttms"]
    one = np.ones(4)
    for tag in ("base", "btc"):
        v0, theta, kappa, rho, volvol = (float(v) for v in g[f"heston_{tag}_params"])
        pr = sv.heston_chain_pricer(v0=v0, theta=theta, kappa=kappa, volvol=volvol, rho=rho, ttms=ttms, forwards=one,
                                    strikes_ttms=(kk,) * 4, optiontypes_ttms=(types,) * 4, discfactors=one)
        np.testing.assert_allclose(np.stack(pr), g[f"heston_{tag}_prices"], rtol=0, atol=1e-13)
        hp = sv.HestonParams(v0=v0, theta=theta, kappa=kappa, rho=rho, volvol=volvol)
        chain = sv.OptionChain(ttms=ttms, forwards=one, strikes_ttms=(kk,) * 4, optiontypes_ttms=(types,) * 4, ids=None)
        np.testing.assert_allclose(np.stack(sv.HestonPricer().price_chain(chain, hp)), np.stack(pr), rtol=0, atol=0)
    chain = sv.OptionChain(ttms=ttms, forwards=one, strikes_ttms=(kk,) * 4, optiontypes_ttms=(types,) * 4, ids=None)
    pricer = sv.LogSVPricer()
    sets, single = [], []
    for tag in ("btc", "readme", "quick", "test", "fig3"):
        v = [float(a) for a in g[f"logsv_{tag}_params"]]
        params = sv.LogSvParams(sigma0=v[0], theta=v[1], kappa1=v[2], kappa2=v[3], beta=v[4], volvol=v[5])
        analytic = pricer.price_chain(chain, params)
        np.testing.assert_allclose(np.stack(analytic), g[f"logsv_{tag}_prices"], rtol=0, atol=2e-6)
        sets.append(params)
        single.append(np.stack(analytic))
    # the five sets in one batch of launches: the same bits as five chains one after the other, both orders and measures
    for a, b in zip(pricer.price_chain_batch(chain, sets), single):
        np.testing.assert_array_equal(np.stack(a), b)
    from stochvolmodels_amd.pricers.logsv.affine_expansion import ExpansionOrder
    batch = pricer.price_chain_batch(chain, sets, is_spot_measure=False, expansion_order=ExpansionOrder.FIRST)
    for a, p_ in zip(batch, sets):
        one_by_one = pricer.price_chain(chain, p_, is_spot_measure=False, expansion_order=ExpansionOrder.FIRST)
        np.testing.assert_array_equal(np.stack(a), np.stack(one_by_one))


def test_analytic_chain_ode_tolerance_knob(sv, golden):
    """ode_rtol= / ode_atol= of the analytic chain pricers: looser tolerances of the coefficient ODEs stay where
    stochvolmodels_amd/analytic.py says they do (1e-8 / 1e-10 within 1e-9 of the default's prices, 1e-6 / 1e-8 within
    5e-8; measured with DOP853: 1.1e-11 and 4.5e-9), single chain and batch alike, and the batch still equals the chains one by one to the bit"""
    g = golden("analytic")
    kk, types, ttms = g["strikes"], g["types"], g["ttms"]
    chain = sv.OptionChain(ttms=ttms, forwards=np.ones(4), strikes_ttms=(kk,) * 4, optiontypes_ttms=(types,) * 4, ids=None)
    pricer = sv.LogSVPricer()
    sets = []
    for tag in ("btc", "test"):
        v = [float(a) for a in g[f"logsv_{tag}_params"]]
        sets.append(sv.LogSvParams(sigma0=v[0], theta=v[1], kappa1=v[2], kappa2=v[3], beta=v[4], volvol=v[5]))
    base = [np.stack(pricer.price_chain(chain, p)) for p in sets]
    for (rt, at), bound in (((1e-8, 1e-10), 1e-9), ((1e-6, 1e-8), 5e-8)):
        loose = [np.stack(pricer.price_chain(chain, p, ode_rtol=rt, ode_atol=at)) for p in sets]
        for a, b in zip(loose, base):
            assert 0.0 < float(np.max(np.abs(a - b))) <= bound, (rt, float(np.max(np.abs(a - b))))
        batch = pricer.price_chain_batch(chain, sets, ode_rtol=rt, ode_atol=at)
        for a, b in zip(batch, loose):
            np.testing.assert_array_equal(np.stack(a), b)


def test_analytic_gives_up_on_unreasonably_stiff_coefficients(sv):
    """A parameter vector whose coefficient ODEs are stiff beyond reason (vol-of-vol of 5000 %: ~10^6 steps of the explicit
    pair) or blow up before the expiry is GIVEN UP within a fraction of a second -- step floor / try cap of
    csrc/svmc_analytic.hip: the grid point's log-MGF is NaN and the inversion drops it, as the reference's np.nansum does
    (utils/mgf_pricer.py:205) -- so every price stays inside [0, max(forward, strike)]; before, the launch ran 4.5-12 s into a
    10^6-try cap and the half-integrated state was inverted into prices of -6e4.  A sane set priced in the same launch, or
    right after, is untouched by the episode."""
    import time
    kk = np.linspace(0.6, 1.4, 21)
    ty = np.where(kk >= 1.0, "C", "P")
    chain = sv.OptionChain(ttms=np.array([1.25, 5.0]), forwards=np.ones(2), strikes_ttms=(kk,) * 2, optiontypes_ttms=(ty,) * 2,
                           ids=None)
    pricer = sv.LogSVPricer()
    sane = np.stack(pricer.price_chain(chain, sv.LOGSV_BTC_PARAMS))
    assert np.all(np.isfinite(sane)) and np.all(sane > 0.0)
    wild = sv.LogSvParams(sigma0=1.0, theta=1.0, kappa1=0.1, kappa2=0.1, beta=50.0, volvol=50.0)
    from stochvolmodels_amd.pricers import logsv_pricer as lp
    assert lp.LAST_ANALYTIC_GIVEN_UP == 0                              # the sane set: no grid point was given up
    t0 = time.perf_counter()
    with pytest.warns(RuntimeWarning, match="given up"):               # ... and the caller is TOLD when some were
        out = np.stack(pricer.price_chain(chain, wild))
    seconds = time.perf_counter() - t0
    assert 0 < lp.LAST_ANALYTIC_GIVEN_UP <= 1000
    assert seconds < 3.0, seconds                # (4.5 - 12 s before the give-up rule; ~0.1 s with it)
    assert np.all((out >= 0.0) & (out <= np.maximum(1.0, kk)[None, :])), out[-1][:3]
    with pytest.warns(RuntimeWarning, match="given up"):
        batch = pricer.price_chain_batch(chain, [sv.LOGSV_BTC_PARAMS, wild])
    assert lp.LAST_ANALYTIC_GIVEN_UP[0] == 0 and lp.LAST_ANALYTIC_GIVEN_UP[1] > 0      # per set: the sane neighbour is clean
    np.testing.assert_array_equal(np.stack(batch[0]), sane)          # a set's neighbours in the launch do not feel it
    np.testing.assert_array_equal(np.stack(batch[1]), out)
    np.testing.assert_array_equal(np.stack(pricer.price_chain(chain, sv.LOGSV_BTC_PARAMS)), sane)


def test_c5_reference_criterion_at_reference_scale(sv, golden):
    """Config C5's acceptance criterion, verbatim and at the reference's own scale: the reference accepts its analytic
    LogSV prices against Monte Carlo when |analytic - MC| <= 4 stderr on a 3-month slice (strikes 0.9 / 1.0 / 1.1,
    P / C / C, DF 0.98) priced with 40 000 paths x 91 steps (tests/test_logsv_characterization.py:346-407).  Here: the same
    slice, path count, step count and inequality -- no additive terms -- for all FIVE parameter sets of C5, GPU analytic
    against GPU Monte Carlo.  (At 2^23 paths the standard error falls below the truncation error of the reference's
    second-order expansion; that bias is measured and reported by tools/c5_bias.py -> profiles/r02_c5_bias.json, not
    absorbed into a tolerance here.)"""
    g = golden("analytic")
    chain = sv.OptionChain.slice_to_chain(ttm=0.25, forward=1.0, strikes=np.array([0.9, 1.0, 1.1]),
                                          optiontypes=np.array(["P", "C", "C"]), discfactor=0.98)
    pricer = sv.LogSVPricer()
    for i, tag in enumerate(("btc", "readme", "quick", "test", "fig3")):
        v = [float(a) for a in g[f"logsv_{tag}_params"]]
        params = sv.LogSvParams(sigma0=v[0], theta=v[1], kappa1=v[2], kappa2=v[3], beta=v[4], volvol=v[5])
        analytic = pricer.price_chain(chain, params)[0]
        mc, sd = pricer.model_mc_price_chain(chain, params, nb_path=40_000, nb_steps=360, seed=123 + i)   # 0.25 * 360 + 1 = 91
        assert np.all(np.isfinite(mc[0])) and np.all(sd[0] > 0.0)
        assert np.all(np.abs(analytic - mc[0]) <= 4.0 * sd[0]), (tag, np.abs(analytic - mc[0]) / sd[0])


def test_resident_fixed_randoms(sv, golden):
    """randoms uploaded once and re-used across calls give the same prices as per-call upload, and a
    second parameter set (a 'calibration iterate') prices on the same resident randoms"""
    g = golden("logsv_tiny_chain")
    p = P(g["params"])
    W0s, W1s = [g["W0_0"], g["W0_1"]], [g["W1_0"], g["W1_1"]]
    common = dict(ttms=g["ttms"], forwards=g["forwards"], discfactors=g["discfactors"], strikes_ttms=tuple(g["strikes"]),
                  optiontypes_ttms=tuple(g["types"]), vol_backbone_etas=np.ones(2))
    res = sv.upload_fixed_randoms(W0s, W1s, g["dts"])
    for _ in range(2):
        pr, sd = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, **common, **p)
        np.testing.assert_allclose(np.stack(pr), g["prices"], **ST)
        np.testing.assert_allclose(np.stack(sd), g["stderrs"], **ST)
    p2 = dict(p, volvol=1.2, beta=-0.1)
    a, _ = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, **common, **p2)
    b, _ = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=W0s, W1s=W1s, dts=g["dts"], **common, **p2)
    np.testing.assert_array_equal(np.stack(a), np.stack(b))
    # the fused C++ driver behind the resident path, on quadratic-variance payoffs, against the Python chain driver
    qv = dict(common, strikes_ttms=(np.array([0.2, 0.6, 1.0]),) * 2, optiontypes_ttms=(np.array(["C", "P", "C"]),) * 2,
              variable_type=sv.VariableType.Q_VAR)
    a, ea = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, **qv, **p2)
    b, eb = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=W0s, W1s=W1s, dts=g["dts"], **qv, **p2)
    np.testing.assert_array_equal(np.stack(a), np.stack(b))
    np.testing.assert_array_equal(np.stack(ea), np.stack(eb))
    # hipGraph replay of the fused driver: on by default, identical bits with it off, re-captured when the chain changes
    from stochvolmodels_amd.engine import option_type_codes
    assert res.graph_launches() >= 4
    args = lambda c, pp, vt: (c["ttms"], c["forwards"], c["discfactors"], [np.asarray(k, float) for k in c["strikes_ttms"]],   # noqa: E731
                              [option_type_codes(t) for t in c["optiontypes_ttms"]], pp["v0"], pp["theta"], pp["kappa1"],
                              pp["kappa2"], pp["beta"], pp["volvol"], np.ones(2), True, vt)
    for c, vt in ((common, 1), (qv, 2), (common, 1)):
        for pp in (p, p2):
            before = res.graph_launches()
            g_on = res.price_logsv_chain(*args(c, pp, vt), use_graph=True)
            assert res.graph_launches() == before + 1
            g_off = res.price_logsv_chain(*args(c, pp, vt), use_graph=False)
            assert res.graph_launches() == before + 1
            for a_, b_ in zip(g_on[0] + g_on[1], g_off[0] + g_off[1]):
                np.testing.assert_array_equal(a_, b_)
    with pytest.raises(ValueError):
        sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, **dict(qv, optiontypes_ttms=(
            np.array(["C", "X", "C"]),) * 2), **p2)
    res.free()


def test_fixed_randoms_drawn_on_device(sv):
    """draw_fixed_randoms_on_device: the chain's fixed randoms made in HBM by the counter-based generator.  They are the
    draws the on-device-RNG pricer consumes for the same (seed, first call of the process' counter): a chain priced on
    them equals logsv_mc_chain_pricer(seed=...) up to the two kernels' evaluation order (1e-9), repeatedly; and an MC
    calibration on them lands where the one on the reference's host-drawn arrays does, within Monte Carlo noise."""
    ttms = np.array([0.1, 0.3, 0.75])
    k = np.linspace(0.7, 1.3, 7)
    ty = np.where(k >= 1.0, "C", "P")
    common = dict(ttms=ttms, forwards=np.array([1.0, 1.01, 1.02]), discfactors=np.array([0.999, 0.99, 0.98]),
                  strikes_ttms=(k,) * 3, optiontypes_ttms=(ty,) * 3, vol_backbone_etas=np.ones(3))
    p = dict(v0=0.8376, theta=1.0413, kappa1=3.1844, kappa2=3.058, beta=0.1514, volvol=1.8458)
    n = 20011
    res = sv.draw_fixed_randoms_on_device(ttms, nb_path=n, nb_steps_per_year=360, seed=77)
    a1, e1 = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, **common, **p)
    a2, e2 = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, **common, **p)
    for u, v in zip(a1 + e1, a2 + e2):
        np.testing.assert_array_equal(u, v)
    b, eb = sv.logsv_mc_chain_pricer(nb_path=n, nb_steps_per_year=360, seed=77, **common, **p)   # (77, call 0): those draws
    np.testing.assert_allclose(np.concatenate(a1), np.concatenate(b), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(np.concatenate(e1), np.concatenate(eb), rtol=1e-12, atol=1e-12)
    res.free()


def test_parameter_sets_share_one_pass_over_the_randoms(sv):
    """logsv_mc_chain_pricer_fixed_randoms_batch / svmc_logsv_chain_price_fixed_sets: several parameter sets (the base point
    of an optimizer iterate and its finite-difference neighbours) stepped by ONE launch that reads the resident randoms
    once, every lane carrying all the sets' states.  Per set: the bits of a single-set call -- prices, standard errors
    and implied vols; 1, 2, 3, 6, 8 and 9 sets (9 = a launch of 8 and a single); LOG_RETURN with vol backbones and
    Q_VAR; a replay equals the capture."""
    ttms = np.array([0.1, 0.3, 0.75])
    k = np.linspace(0.7, 1.3, 7)
    ty = np.where(k >= 1.0, "C", "P")
    common = dict(ttms=ttms, forwards=np.array([1.0, 1.01, 1.02]), discfactors=np.array([0.999, 0.99, 0.98]),
                  strikes_ttms=(k,) * 3, optiontypes_ttms=(ty,) * 3)
    W = sv.get_randoms_for_chain_valuation(ttms, nb_path=6007, nb_steps_per_year=360, seed=5)
    res = sv.upload_fixed_randoms(*W)
    import pandas as pd
    rng = np.random.default_rng(9)
    sets = []
    for j in range(9):
        bb = None if j % 2 == 0 else pd.Series(1.0 + 0.1 * rng.standard_normal(3), index=ttms)
        sets.append(sv.LogSvParams(sigma0=0.8 + 0.02 * j, theta=1.0 + 0.01 * j, kappa1=3.0 + 0.1 * j, kappa2=3.0 - 0.1 * j,
                                   beta=0.15 - 0.03 * j, volvol=1.8 - 0.05 * j, vol_backbone=bb))

    def single(p, **kw):
        return sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, v0=p.sigma0, theta=p.theta, kappa1=p.kappa1,
                                                      kappa2=p.kappa2, beta=p.beta, volvol=p.volvol,
                                                      vol_backbone_etas=p.get_vol_backbone_etas(ttms=ttms), **common, **kw)

    for n_sets in (1, 2, 3, 6, 8, 9):
        for rep in range(2):                                        # capture, then replay
            out = sv.logsv_mc_chain_pricer_fixed_randoms_batch(params_list=sets[:n_sets], W0s=res, return_ivols=True, **common)
            assert len(out) == n_sets
            for p, got in zip(sets, out):
                want = single(p, return_ivols=True)
                for a, b in zip(got[0] + got[1] + got[2], want[0] + want[1] + want[2]):
                    np.testing.assert_array_equal(a, b, err_msg=f"{n_sets} sets, rep {rep}")
    qv = dict(common, strikes_ttms=(np.array([0.2, 0.6, 1.0]),) * 3, optiontypes_ttms=(np.array(["C", "P", "C"]),) * 3)
    out = sv.logsv_mc_chain_pricer_fixed_randoms_batch(params_list=sets[:4], W0s=res, variable_type=sv.VariableType.Q_VAR, **qv)
    for p, got in zip(sets, out):
        want = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, v0=p.sigma0, theta=p.theta, kappa1=p.kappa1,
                                                      kappa2=p.kappa2, beta=p.beta, volvol=p.volvol,
                                                      vol_backbone_etas=p.get_vol_backbone_etas(ttms=ttms),
                                                      variable_type=sv.VariableType.Q_VAR, **qv)
        for a, b in zip(got[0] + got[1], want[0] + want[1]):
            np.testing.assert_array_equal(a, b)
    res.free()


def test_frozen_randoms_regenerated_in_registers(sv):
    """draw_fixed_randoms_on_device() keeps NOTHING: the object names the stream (seed, call 0) and every pricing regenerates
    the normals in registers (svmc_logsv_chain_price_frozen_sets, logsv_chain_rng_sets_kernel<P>: one draw per step shared by
    the P parameter sets of a launch).  Per set the prices, standard errors and implied vols are those of
    logsv_mc_chain_pricer(seed=...) with that set's parameters BIT FOR BIT -- 1, 2, 3, 6, 8 and 9 sets (9 = a launch of 8 and
    a single), with vol backbones, in both measures, LOG_RETURN and Q_VAR, graph capture and replay, graphs off, one expiry
    -- and no HBM is held for randoms."""
    from stochvolmodels_amd.data.option_chain import black_ivols_native
    import pandas as pd
    ttms = np.array([0.1, 0.3, 0.75])
    k = np.linspace(0.7, 1.3, 7)
    ty = np.where(k >= 1.0, "C", "P")
    common = dict(ttms=ttms, forwards=np.array([1.0, 1.01, 1.02]), discfactors=np.array([0.999, 0.99, 0.98]),
                  strikes_ttms=(k,) * 3, optiontypes_ttms=(ty,) * 3)
    n, seed = 20011, 77
    rng = np.random.default_rng(9)
    sets = []
    for j in range(9):
        bb = None if j % 2 == 0 else pd.Series(1.0 + 0.1 * rng.standard_normal(3), index=ttms)
        sets.append(sv.LogSvParams(sigma0=0.8 + 0.02 * j, theta=1.0 + 0.01 * j, kappa1=3.0 + 0.1 * j, kappa2=3.0 - 0.1 * j,
                                   beta=0.15 - 0.03 * j, volvol=1.8 - 0.05 * j, vol_backbone=bb))

    def rng_route(p, chain=common, **kw):
        return sv.logsv_mc_chain_pricer(v0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta, volvol=p.volvol,
                                        vol_backbone_etas=p.get_vol_backbone_etas(ttms=chain["ttms"]), nb_path=n,
                                        nb_steps_per_year=360, seed=seed, **chain, **kw)

    res = sv.draw_fixed_randoms_on_device(ttms, nb_path=n, nb_steps_per_year=360, seed=seed)
    assert res.is_frozen and res.w0 == [] and res.w1 == []
    for spot in (True, False):
        for n_sets in (1, 2, 3, 6, 8, 9):
            for rep in range(2):                                    # capture, then replay
                out = sv.logsv_mc_chain_pricer_fixed_randoms_batch(params_list=sets[:n_sets], W0s=res, return_ivols=True,
                                                                   is_spot_measure=spot, **common)
                assert len(out) == n_sets
                for p, got in zip(sets, out):
                    want = rng_route(p, is_spot_measure=spot)
                    for a, b in zip(got[0] + got[1], want[0] + want[1]):
                        np.testing.assert_array_equal(a, b, err_msg=f"{n_sets} sets, rep {rep}, spot {spot}")
                    for iv, pr, t, f, d in zip(got[2], got[0], ttms, common["forwards"], common["discfactors"]):
                        np.testing.assert_allclose(iv, black_ivols_native(pr, float(t), float(f), k, ty, float(d)), rtol=1e-9,
                                                   equal_nan=True)
    # the single-set entry point (an objective evaluation), graph on and off
    p = sets[3]
    want = rng_route(p)
    for use_graph in (True, False):
        got = res.price_logsv_chain(ttms, common["forwards"], common["discfactors"], [k] * 3, [np.where(k >= 1.0, 0, 1).astype(np.int8)] * 3,
                                    p.sigma0, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, p.get_vol_backbone_etas(ttms=ttms), True,
                                    1, use_graph=use_graph)
        for a, b in zip(got[0] + got[1], want[0] + want[1]):
            np.testing.assert_array_equal(a, b)
    got = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, v0=p.sigma0, theta=p.theta, kappa1=p.kappa1,
                                                 kappa2=p.kappa2, beta=p.beta, volvol=p.volvol,
                                                 vol_backbone_etas=p.get_vol_backbone_etas(ttms=ttms), **common)
    for a, b in zip(got[0] + got[1], want[0] + want[1]):
        np.testing.assert_array_equal(a, b)
    # options on the quadratic variance
    qv = dict(common, strikes_ttms=(np.array([0.2, 0.6, 1.0]),) * 3, optiontypes_ttms=(np.array(["C", "P", "C"]),) * 3)
    out = sv.logsv_mc_chain_pricer_fixed_randoms_batch(params_list=sets[:4], W0s=res, variable_type=sv.VariableType.Q_VAR, **qv)
    for p, got in zip(sets, out):
        want = rng_route(p, chain=qv, variable_type=sv.VariableType.Q_VAR)
        for a, b in zip(got[0] + got[1], want[0] + want[1]):
            np.testing.assert_array_equal(a, b)
    res.free()
    # one expiry: the whole-chain kernel with m = 1 against the one-slice generator
    one = dict(ttms=ttms[:1], forwards=common["forwards"][:1], discfactors=common["discfactors"][:1], strikes_ttms=(k,),
               optiontypes_ttms=(ty,))
    res1 = sv.draw_fixed_randoms_on_device(ttms[:1], nb_path=n, nb_steps_per_year=360, seed=seed)
    out = sv.logsv_mc_chain_pricer_fixed_randoms_batch(params_list=sets[:3], W0s=res1, **one)
    for p, got in zip(sets, out):
        want = rng_route(p, chain=one)
        for a, b in zip(got[0] + got[1], want[0] + want[1]):
            np.testing.assert_array_equal(a, b)
    res1.free()
    # the materialised form is still there on request: the same draws in HBM, priced by the streamed kernels (rounding-level apart)
    hbm = sv.draw_fixed_randoms_on_device(ttms, nb_path=n, nb_steps_per_year=360, seed=seed, in_hbm=True)
    assert not hbm.is_frozen and len(hbm.w0) == 3
    a, _ = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=hbm, W1s=None, dts=None, v0=p.sigma0, theta=p.theta, kappa1=p.kappa1,
                                                  kappa2=p.kappa2, beta=p.beta, volvol=p.volvol, vol_backbone_etas=np.ones(3), **common)
    b, _ = rng_route(sv.LogSvParams(sigma0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta, volvol=p.volvol))
    np.testing.assert_allclose(np.concatenate(a), np.concatenate(b), rtol=1e-12, atol=1e-12)
    hbm.free()


@pytest.mark.parametrize("n", [777, 20011, 300_001])
def test_payoff_sums_of_all_sets_in_one_launch(sv, n):
    """the payoff pass of a multi-set evaluation is ONE launch (blockIdx.z = parameter set) and ONE column reduce where the
    chain's strike groups fit a launch, the per-set loop where they do not: either way set q's prices and standard errors are
    logsv_mc_chain_pricer's bits -- ragged chains, inverse options (their third accumulator, groups of 16), an expiry wider
    than a group, nine expiries (more groups than a launch takes), path counts either side of the four-trips rule"""
    rng = np.random.default_rng(n)
    sets = [sv.LogSvParams(sigma0=0.8 + 0.03 * j, theta=1.0, kappa1=3.0 + 0.2 * j, kappa2=3.0, beta=0.15 - 0.05 * j, volvol=1.8 - 0.1 * j)
            for j in range(5)]

    def chain_of(counts, inverse):
        m = len(counts)
        ttms = np.linspace(0.05, 0.4, m)
        fw = 1.0 + 0.01 * np.arange(m)
        ks = tuple(np.sort(f * rng.uniform(0.6, 1.5, c)) for f, c in zip(fw, counts))
        if inverse:
            tys = tuple(np.where(k >= f, np.where(np.arange(k.size) % 2 == 0, "IC", "C"), np.where(np.arange(k.size) % 3 == 0, "IP", "P"))
                        for k, f in zip(ks, fw))
        else:
            tys = tuple(np.where(k >= f, "C", "P") for k, f in zip(ks, fw))
        return dict(ttms=ttms, forwards=fw, discfactors=np.exp(-0.03 * ttms), strikes_ttms=ks, optiontypes_ttms=tys)

    cases = [("ragged", chain_of((5, 17, 1), False), sv.VariableType.LOG_RETURN),
             ("inverse, an expiry of 30", chain_of((4, 30, 9), True), sv.VariableType.LOG_RETURN),
             ("nine expiries", chain_of((3,) * 9, False), sv.VariableType.LOG_RETURN),
             ("wide, realised variance", chain_of((23, 2), False), sv.VariableType.Q_VAR)]
    for tag, chain, vt in cases:
        if vt == sv.VariableType.Q_VAR:
            chain = dict(chain, strikes_ttms=tuple(np.abs(k - 0.9) + 0.05 for k in chain["strikes_ttms"]))
        res = sv.draw_fixed_randoms_on_device(chain["ttms"], nb_path=n, nb_steps_per_year=120, seed=5)
        out = sv.logsv_mc_chain_pricer_fixed_randoms_batch(params_list=sets, W0s=res, variable_type=vt, **chain)
        for p, got in zip(sets, out):
            want = sv.logsv_mc_chain_pricer(v0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta, volvol=p.volvol,
                                            vol_backbone_etas=np.ones(len(chain["ttms"])), nb_path=n, nb_steps_per_year=120, seed=5,
                                            variable_type=vt, **chain)
            for a, b in zip(got[0] + got[1], want[0] + want[1]):
                np.testing.assert_array_equal(a, b, err_msg=f"{tag}, {n} paths")
        res.free()


def test_implied_vols_from_the_graph(sv, oracle):
    """the price -> implied-vol step of the calibration objective done by the last kernel of the replayed graph
    (svmc_logsv_chain_price_fixed_iv): same numbers as the host routine on the returned prices (the same solver, device
    libm against the host's), identical with the graph off (host route), the same NaN pattern for unattainable prices,
    prices untouched by asking for them; inverse quotes are inverted as price x forward, as on the host"""
    from stochvolmodels_amd.data.option_chain import black_ivols_native
    ttms = np.array([1 / 12, 0.25, 0.5, 1.0])
    k = np.linspace(0.55, 1.6, 13)
    ty = np.where(k >= 1.0, "C", "P")
    common = dict(ttms=ttms, forwards=np.array([1.0, 1.01, 1.02, 1.04]), discfactors=np.array([0.999, 0.99, 0.98, 0.96]),
                  strikes_ttms=(k,) * 4, optiontypes_ttms=(ty,) * 4, vol_backbone_etas=np.ones(4))
    p = dict(v0=0.8376, theta=1.0413, kappa1=3.1844, kappa2=3.058, beta=0.1514, volvol=1.8458)
    W = sv.get_randoms_for_chain_valuation(ttms, nb_path=20000, nb_steps_per_year=360, seed=3)
    res = sv.upload_fixed_randoms(*W)
    for pp in (p, dict(p, volvol=1.2, beta=-0.2), dict(p, v0=0.05, theta=0.05, volvol=0.3)):   # the last: far strikes unattainable
        pr0, sd0 = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, **common, **pp)
        pr, sd, iv = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, return_ivols=True, **common, **pp)
        for a, b in zip(pr0 + sd0, pr + sd):
            np.testing.assert_array_equal(a, b)
        for i in range(4):
            host = black_ivols_native(pr[i], float(ttms[i]), float(common["forwards"][i]), k, ty, float(common["discfactors"][i]))
            assert np.array_equal(np.isnan(iv[i]), np.isnan(host))
            ok = ~np.isnan(host)
            np.testing.assert_allclose(iv[i][ok], host[ok], rtol=1e-12)
        assert np.isfinite(np.concatenate(iv)).sum() >= (30 if pp["volvol"] > 1.0 else 8)
    assert np.isnan(np.concatenate(iv)).any()                       # the low-vol set loses its far strikes
    # graph off / per-call upload: the host route gives the same vols
    from stochvolmodels_amd.engine import option_type_codes
    args = (ttms, common["forwards"], common["discfactors"], [k] * 4, [option_type_codes(ty)] * 4, p["v0"], p["theta"],
            p["kappa1"], p["kappa2"], p["beta"], p["volvol"], np.ones(4), True, 1)
    on = res.price_logsv_chain(*args, use_graph=True, want_ivols=True)
    off = res.price_logsv_chain(*args, use_graph=False, want_ivols=True)
    np.testing.assert_allclose(np.concatenate(on[2]), np.concatenate(off[2]), rtol=1e-12)
    _, _, iv_host = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=W[0], W1s=W[1], dts=W[2], return_ivols=True, **common, **p)
    np.testing.assert_allclose(np.concatenate(iv_host), np.concatenate(on[2]), rtol=1e-12)
    # inverse quotes: the vanilla inversion of price x forward, from the graph's last kernel and from the host routine alike
    inv_types = (np.where(k >= 1.0, "IC", "IP"),) * 4
    pr_i, _, iv_i = sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, return_ivols=True,
                                                           **dict(common, optiontypes_ttms=inv_types), **p)
    chain_i = sv.OptionChain(ttms=ttms, forwards=common["forwards"], strikes_ttms=(k,) * 4, optiontypes_ttms=inv_types, ids=None,
                             discfactors=common["discfactors"])
    host_i = chain_i.compute_model_ivols_from_chain_data(pr_i)
    np.testing.assert_allclose(np.concatenate(iv_i), np.concatenate(host_i), rtol=1e-10, equal_nan=True)
    assert np.isfinite(np.concatenate(iv_i)).sum() >= 30
    res.free()


def test_mc_chain_implied_vols(sv):
    """ModelPricer.compute_mc_chain_implied_vols (reference model_pricer.py:216-241): MC price +/- 1.96 stderr -> Black
    vols; shape / ordering contract of the reference's tests/test_model_calibration_contracts.py:97-118"""
    kk = np.array([0.8, 0.9, 1.0, 1.1, 1.2])
    chain = sv.OptionChain.get_uniform_chain(ttms=np.array([0.25, 0.5]), ids=np.array(["3m", "6m"]),
                                             forwards=np.array([1.0, 1.0]), strikes=kk)
    p = sv.LogSvParams(sigma0=1.0, theta=1.0, kappa1=5.0, kappa2=5.0, beta=0.2, volvol=2.0)
    out = sv.LogSVPricer().compute_mc_chain_implied_vols(chain, p, nb_path=1 << 18, seed=3)
    prices, ups, downs, mid, up, down, stds = out
    assert len(prices) == len(mid) == 2 and mid[0].shape == (5,)
    for i in range(2):
        assert np.all(downs[i] <= prices[i]) and np.all(prices[i] <= ups[i])
        assert np.all(down[i] <= mid[i] + 1e-12) and np.all(mid[i] <= up[i] + 1e-12)
        assert np.all((mid[i] > 0.7) & (mid[i] < 1.3))               # ~100% vol model
    # the quickstart's analytic 6m ATM vol inside three widths of the Monte Carlo band itself (1.96 stderr each way)
    assert abs(mid[1][2] - 0.995757) <= 1.5 * (up[1][2] - down[1][2])


def test_analytic_qvar(sv, oracle, golden):
    """analytic calls on quadratic variance (40 000 psi-grid lanes per expiry) vs the reference, and vs the GPU Monte
    Carlo Q_VAR price at the reference's own scale and criterion (40 000 paths, |analytic - MC| <= 4 stderr).  The
    second-order affine expansion is an approximation: for the BTC set's SECOND expiry the Monte Carlo price sits below it
    by 0.4 % at the money to 6 % at the far strike -- measured at 2^22 paths (z = -9.6 .. -33) and as a mean z of -0.9 ..
    -3.0 over 24 seeds at 40 000 paths, identically under the round-2 Box-Muller stream and the round-3 inverse-CDF one
    (tools/r03/qvar_zscores.py, profiles/r03_qvar_expansion_bias.json).  A bare 4-stderr band around a price that is 3
    stderr off on average fails one seed in six whatever the generator -- so the statistical statement asserted here is
    VERDICT PARITY: the GPU's accept / reject map equals the reference's on the same stream (round 4 also asserted the band
    itself, widened by the measured 6.5 % truncation error: an additive term in a statistical assert, and a duplicate)."""
    g = golden("analytic_qvar")
    for tag in ("test", "btc"):
        v = [float(a) for a in g[f"{tag}_params"]]
        params = sv.LogSvParams(sigma0=v[0], theta=v[1], kappa1=v[2], kappa2=v[3], beta=v[4], volvol=v[5])
        kk = g[f"{tag}_strikes"]
        chain = sv.OptionChain(ttms=g["ttms"], forwards=g["forwards"], strikes_ttms=(kk, kk),
                               optiontypes_ttms=(np.array(["C"] * 8),) * 2, ids=None, discfactors=g["discfactors"])
        pricer = sv.LogSVPricer()
        an = pricer.price_chain(chain, params, variable_type=sv.VariableType.Q_VAR)
        np.testing.assert_allclose(np.stack(an), g[f"{tag}_prices"], rtol=0, atol=5e-6)
        if tag == "test":
            np.testing.assert_allclose(an[0], g["test_tight_prices"][0], rtol=0, atol=1e-8)
        # Monte Carlo at the reference's own scale (40 000 paths, its acceptance criterion |analytic - MC| <= 4 stderr,
        # tests/test_logsv_characterization.py:407), no additive terms; the expansion's bias on far OTM variance calls,
        # visible only at millions of paths, is reported by tools/c5_bias.py (profiles/r02_c5_bias.json)
        mc, sd = pricer.model_mc_price_chain(chain, params, variable_type=sv.VariableType.Q_VAR, nb_path=40_000,
                                             nb_steps=720, seed=8)
        diff, band = np.stack(mc) - np.stack(an), 4.0 * np.stack(sd)
        print(f"analytic vs MC Q_VAR [{tag}]: z = {np.round(diff / np.stack(sd), 2).tolist()}")
        # VERDICT PARITY with the bare criterion, no additive terms: per option, |analytic - MC| <= 4 stderr as the GPU answers
        # it (its analytic chain against its Monte Carlo) and as the reference answers it (its analytic prices -- the golden --
        # against the oracle's Monte Carlo on the same stream).  Asserted as (i) the two z-scores agree to DELTA everywhere and
        # (ii) the accept / reject maps are equal wherever the reference's |z| is further than 10 DELTA from the threshold -- an
        # option ON the knife edge (the BTC set's second expiry sits 0.02 of a z-unit from it) may fall either way with the
        # solver's last digits and says nothing about parity; (i) holds it to the reference all the same
        n, spy, x, s_, q, t0, step0, omc, osd = 40_000, 720, np.zeros(40_000), v[0] * np.ones(40_000), np.zeros(40_000), 0.0, 0, [], []
        for i, ttm in enumerate(g["ttms"]):
            nb, dt, _ = sv.set_time_grid(ttm - t0, spy)
            x, s_, q = oracle.logsv_terminal_rng(x, s_, q, nb, dt, v[1], v[2], v[3], v[4], v[5], 8, step_offset=step0)
            a, b = oracle.payoff(x, q, float(ttm), float(g["forwards"][i]), kk, np.array(["C"] * 8), float(g["discfactors"][i]), 2)
            omc.append(a), osd.append(b)
            t0, step0 = ttm, step0 + nb
        np.testing.assert_allclose(np.stack(mc), np.stack(omc), rtol=1e-11, atol=1e-15)
        gpu_map = np.abs(diff) <= band
        ref_map = np.abs(np.stack(omc) - g[f"{tag}_prices"]) <= 4.0 * np.stack(osd)
        z_ref = (np.stack(omc) - g[f"{tag}_prices"]) / np.stack(osd)
        z_gpu = diff / np.stack(sd)
        clear = np.abs(np.abs(z_ref) - 4.0) > 10.0 * VERDICT_DELTA
        print(f"analytic vs MC Q_VAR [{tag}]: verdict map pass {int(gpu_map.sum())} / fail {int((~gpu_map).sum())}; closest |z| to 4: "
              f"{np.min(np.abs(np.abs(z_ref) - 4.0)):.3f}; max |z_gpu - z_ref| {np.max(np.abs(z_gpu - z_ref)):.2e}; "
              f"{int((~clear).sum())} option(s) inside the dead band")
        assert np.max(np.abs(z_gpu - z_ref)) <= VERDICT_DELTA, (tag, np.max(np.abs(z_gpu - z_ref)))
        np.testing.assert_array_equal(gpu_map[clear], ref_map[clear],
                                      err_msg=f"Q_VAR {tag}: the accept / reject map differs from the reference's away from the threshold")
    with pytest.raises(ValueError):
        chain_p = sv.OptionChain.slice_to_chain(0.25, 1.0, np.array([0.04]), np.array(["P"]))
        sv.LogSVPricer().price_chain(chain_p, sv.LogSvParams(), variable_type=sv.VariableType.Q_VAR)


def test_config_c4_btc_style_chain(sv, oracle):
    """BASELINE config 4 (one rank's share): LogSV, 8 expiries.  (a) the irregular BTC-style maturities
    [1/52, 1/24, 1/12, 1/6, 1/4, 1/2, 3/4, 1] at spy = 1016 -- per-slice step counts follow int(dT*spy)+1 -- with
    forwards 67000 e^{0.05 T}, mixed C/P/IC/IP, path-wise against the CPU twin at 4096 paths; (b) the regular k/8
    chain at full per-rank size (2^21 paths x 1024 steps): per-slice put-call parity, martingale, determinism."""
    P_ = sv.LOGSV_BTC_PARAMS
    spy = 1016
    ttms = np.array([1 / 52, 1 / 24, 1 / 12, 1 / 6, 1 / 4, 1 / 2, 3 / 4, 1.0])
    fw = 67000.0 * np.exp(0.05 * ttms)
    dfs = np.exp(-0.05 * ttms)
    strikes = tuple(f * np.linspace(0.6, 1.6, 21) for f in fw)
    base = np.array((["P", "IP", "C", "IC"] * 6)[:21])
    types = tuple(base for _ in ttms)
    n, seed = 4096, 77
    pr, sd = sv.logsv_mc_chain_pricer(ttms=ttms, forwards=fw, discfactors=dfs, strikes_ttms=strikes, optiontypes_ttms=types,
                                      v0=P_.sigma0, theta=P_.theta, kappa1=P_.kappa1, kappa2=P_.kappa2, beta=P_.beta,
                                      volvol=P_.volvol, vol_backbone_etas=np.ones(8), nb_path=n, nb_steps_per_year=spy,
                                      seed=seed)
    x, s, q = np.zeros(n), P_.sigma0 * np.ones(n), np.zeros(n)
    t0, step0, total = 0.0, 0, 0
    for i, ttm in enumerate(ttms):
        nb, dt, _ = sv.set_time_grid(ttm - t0, spy)
        assert nb == int((ttm - t0) * spy) + 1
        x, s, q = oracle.logsv_terminal_rng(x, s, q, nb, dt, P_.theta, P_.kappa1, P_.kappa2, P_.beta, P_.volvol, seed,
                                            step_offset=step0)
        step0, t0, total = step0 + nb, ttm, total + nb
        opr, osd = oracle.payoff(x, q, float(ttm), float(fw[i]), strikes[i], types[i], float(dfs[i]))
        np.testing.assert_allclose(pr[i], opr, rtol=1e-12, atol=1e-13 * fw[i], err_msg=f"slice {i}")
        np.testing.assert_allclose(sd[i], osd, rtol=1e-12, atol=1e-13 * fw[i], err_msg=f"slice {i}")
    assert total == 1021                                           # sum over the 8 slices of int(dT*1016)+1: the
    #                                                                actual step count is what enters path-steps/s
    # (b) regular chain, full per-rank size
    ttms = np.arange(1, 9) / 8.0
    fw = 67000.0 * np.exp(0.05 * ttms)
    kk = tuple(f * np.linspace(0.6, 1.6, 21) for f in fw)
    both = tuple(np.concatenate([k, k]) for k in kk)
    cp = tuple(np.array(["C"] * 21 + ["P"] * 21) for _ in ttms)
    chain = sv.OptionChain(ttms=ttms, forwards=fw, strikes_ttms=both, optiontypes_ttms=cp, ids=None)
    pricer = sv.LogSVPricer()
    n = 1 << 21
    a, sa = pricer.model_mc_price_chain(chain, P_, nb_path=n, nb_steps=1016, seed=5)
    b, sb = pricer.model_mc_price_chain(chain, P_, nb_path=n, nb_steps=1016, seed=5)
    for i in range(8):
        np.testing.assert_array_equal(a[i], b[i])
        np.testing.assert_allclose(a[i][:21] - a[i][21:], fw[i] - kk[i], rtol=0, atol=1e-13 * fw[i])  # exact parity
        assert np.all(np.isfinite(a[i])) and np.all(sa[i] > 0)
    from stochvolmodels_amd.engine import get_engine
    x, s, q = get_engine(n).get_state()
    assert abs(np.mean(np.exp(x)) - 1.0) <= 4.0 * np.std(np.exp(x)) / np.sqrt(n)
    assert np.all(s > 0) and np.all(q >= 0)


def test_one_device_tail_equals_five_node_tail(sv):
    """round 6 ends a chain on one device with two launches -- the payoff kernel (its blocks sum their expiry's per-wave spot
    partials themselves) and chain_finish_kernel (a wave per quote: the column sums, price -> implied vol, results stored in
    pinned memory) -- where round 5 ran reduce, payoff, reduce, [implied vols,] copy.  Every sum is formed in the same order of
    additions, so LogSV (chain with inverse options, one expiry, Q_VAR), Heston (Euler, QE) and the frozen-randoms objective
    (one set, five sets, implied vols: that route kept the five-node tail, so it checks nothing but itself) must come out BIT FOR
    BIT the same either way, below and above the 2048-row switch of the in-kernel spot sums"""
    import json
    import subprocess
    import sys as _sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tail_nodes_worker.py")
    got = {}
    for nodes in ("2", "5"):
        env = dict(os.environ, SVMC_CHAIN_TAIL_NODES=nodes)
        r = subprocess.run([_sys.executable, worker], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[nodes] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert got["2"].keys() == got["5"].keys() and len(got["2"]) == 29
    diff = [k for k in got["2"] if got["2"][k] != got["5"][k]]
    assert not diff, diff


@pytest.mark.parametrize("n", [131072, 131073, 458752, 458753])
def test_few_waves_kernels_either_side_of_their_path_count(sv, oracle, n):
    """up to seven waves per SIMD (458752 paths on an MI355X) the on-device-RNG LogSV generators run as logsv_rng_lat_kernel /
    logsv_chain_rng_lat_kernel (256-thread blocks, the pipelined time loop), above as the full-launch kernels; 131072 paths are
    the last launch whose payoff blocks sum the spot partials themselves (2048 rows), 131073 the first with a reduce launch ahead:
    the same statements -- path by path the CPU twin's numbers on the same stream either side of both switches, for a chain
    (both kernels' slice loops, odd slice boundaries) and for one expiry, prices, standard errors and the state the launch
    leaves behind"""
    from stochvolmodels_amd.engine import get_engine
    P_ = sv.LOGSV_BTC_PARAMS
    seed, spy = 31, 60
    k = np.array([0.8, 1.0, 1.25])
    ty = np.array(["P", "C", "IC"])
    for ttms in (np.array([0.12, 0.3, 0.55]), np.array([0.3])):
        m = len(ttms)
        fw, df = 1.0 + 0.02 * np.arange(m), np.exp(-0.04 * ttms)
        pr, sd = sv.logsv_mc_chain_pricer(ttms=ttms, forwards=fw, discfactors=df, strikes_ttms=(k,) * m, optiontypes_ttms=(ty,) * m,
                                          v0=P_.sigma0, theta=P_.theta, kappa1=P_.kappa1, kappa2=P_.kappa2, beta=P_.beta,
                                          volvol=P_.volvol, vol_backbone_etas=np.ones(m), nb_path=n, nb_steps_per_year=spy, seed=seed)
        x, s, q = np.zeros(n), P_.sigma0 * np.ones(n), np.zeros(n)
        t0, step0 = 0.0, 0
        for i, ttm in enumerate(ttms):
            nb, dt, _ = sv.set_time_grid(ttm - t0, spy)
            x, s, q = oracle.logsv_terminal_rng(x, s, q, nb, dt, P_.theta, P_.kappa1, P_.kappa2, P_.beta, P_.volvol, seed,
                                                step_offset=step0)
            step0, t0 = step0 + nb, ttm
            opr, osd = oracle.payoff(x, q, float(ttm), float(fw[i]), k, ty, float(df[i]))
            np.testing.assert_allclose(pr[i], opr, rtol=1e-11, atol=1e-13)
            np.testing.assert_allclose(sd[i], osd, rtol=1e-11, atol=1e-13)
        gx, gs, gq = get_engine(n).get_state()
        np.testing.assert_allclose(gx, x, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(gs, s, rtol=1e-10)
        np.testing.assert_allclose(gq, q, rtol=1e-10)


def test_c_host_calibration_objective(sv, tmp_path):
    """examples/calibration_objective.c: the objective and the gradient evaluation of an MC calibration from plain C on FROZEN
    randoms (svmc_logsv_chain_price_frozen_sets, nothing resident) -- per set the Python host's logsv_mc_chain_pricer(seed) bit
    for bit, the implied vols those of the host inversion; the program itself checks its base point against the chain driver"""
    import json
    import os
    import subprocess
    from stochvolmodels_amd.data.option_chain import black_ivols_native
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "calibration_objective")
    libdir = os.path.join(root, "stochvolmodels_amd")
    subprocess.run(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "calibration_objective.c"),
                    "-o", exe, "-L" + libdir, "-lsvmc", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-lm"], check=True)
    n = 30011
    run = subprocess.run([exe, str(n), "77", "20"], check=True, capture_output=True, text=True)
    out = json.loads(run.stdout.replace("NaN", "null"))
    assert out["base_equals_chain_driver"] is True and out["steps"] == 364 and out["one_set_ms"] > 0 and out["seven_sets_ms"] > 0
    ttms = np.array([1 / 12, 0.25, 0.5, 1.0])
    k = 0.7 + 0.05 * np.arange(13)
    ty = np.where(np.arange(13) >= 6, "C", "P")
    chain = dict(ttms=ttms, forwards=np.ones(4), discfactors=np.ones(4), strikes_ttms=(k,) * 4, optiontypes_ttms=(ty,) * 4)
    base = np.array([0.8376, 1.0413, 3.1844, 3.058, 0.1514, 1.8458])
    prices = np.array(out["prices"], dtype=float).reshape(7, 4, 13)
    stderrs = np.array(out["stderrs"], dtype=float).reshape(7, 4, 13)
    ivols = np.array([np.nan if v is None else v for v in out["ivols"]], dtype=float).reshape(7, 4, 13)
    for q in (0, 1, 4, 6):
        p = base.copy()
        if q:
            p[q - 1] *= 1.0 + 1e-4
        pr, sd = sv.logsv_mc_chain_pricer(v0=p[0], theta=p[1], kappa1=p[2], kappa2=p[3], beta=p[4], volvol=p[5],
                                          vol_backbone_etas=np.ones(4), nb_path=n, nb_steps_per_year=360, seed=77, **chain)
        np.testing.assert_array_equal(np.array(pr), prices[q])
        np.testing.assert_array_equal(np.array(sd), stderrs[q])
        for i in range(4):
            np.testing.assert_allclose(ivols[q, i], black_ivols_native(prices[q, i], float(ttms[i]), 1.0, k, ty, 1.0), rtol=1e-9,
                                       equal_nan=True)


def test_c_host_example(sv, tmp_path):
    """the drop-in boundary is a C ABI: examples/price_chain.c (plain C, gcc, no Python, no torch) prices a chain through
    the fused drivers svmc_logsv_chain_price / svmc_heston_chain_price; the Python host with the same seed must give
    the same prices (same kernels, same order of launches)"""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "price_chain")
    libdir = os.path.join(root, "stochvolmodels_amd")
    subprocess.run(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "price_chain.c"),
                    "-o", exe, "-L" + libdir, "-lsvmc", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-lm"],
                   check=True)
    out = json.loads(subprocess.run([exe, "65536", "123"], check=True, capture_output=True, text=True).stdout)
    P_ = sv.LOGSV_BTC_PARAMS
    ttms, fw, df = np.array([0.1, 0.25]), np.array([1.0, 1.01]), np.array([0.99, 0.98])
    kk = np.array([0.8, 1.0, 1.2])
    strikes = (kk, 1.01 * kk)
    types = (np.array(["P", "C", "C"]), np.array(["IP", "IC", "C"]))
    pr, sd = sv.logsv_mc_chain_pricer(ttms=ttms, forwards=fw, discfactors=df, strikes_ttms=strikes, optiontypes_ttms=types,
                                      v0=P_.sigma0, theta=P_.theta, kappa1=P_.kappa1, kappa2=P_.kappa2, beta=P_.beta,
                                      volvol=P_.volvol, vol_backbone_etas=np.ones(2), nb_path=65536, nb_steps_per_year=120,
                                      seed=123)
    np.testing.assert_array_equal(np.concatenate(pr), out["logsv_prices"])
    np.testing.assert_array_equal(np.concatenate(sd), out["logsv_stderrs"])
    # ... and the same chain on state arrays the C host owns (svmc_session_create_on): the same prices, and the terminal state of
    # the first paths where the host left its arrays -- the Python engine's, which is driven through that very entry point
    np.testing.assert_array_equal(out["logsv_prices_on_caller_state"], out["logsv_prices"])
    from stochvolmodels_amd.engine import get_engine
    gx, gs, gq = get_engine(65536).get_state()
    np.testing.assert_array_equal(gx[:4], out["terminal_x_head"])
    np.testing.assert_array_equal(gs[:4], out["terminal_sigma_head"])
    np.testing.assert_array_equal(gq[:4], out["terminal_qvar_head"])
    pr, sd = sv.heston_mc_chain_pricer(ttms=ttms[:1], forwards=fw[:1], discfactors=df[:1], strikes_ttms=strikes[:1],
                                       optiontypes_ttms=types[:1], v0=0.04, theta=0.04, kappa=4.0, rho=-0.5, volvol=0.4,
                                       nb_path=65536, seed=123)
    np.testing.assert_array_equal(pr[0], out["heston_prices"])
    np.testing.assert_array_equal(sd[0], out["heston_stderrs"])
    # the calibration inner loop from C: the same device-drawn randoms (call id 1) through the Python fixed-randoms API
    eng = _engine(65536)
    w0p, w1p = eng.fill_normals(12, 123, call_id=1)
    W0, W1 = eng.download(w0p, 12 * 65536).reshape(12, -1), eng.download(w1p, 12 * 65536).reshape(12, -1)
    for it, volvol in enumerate((P_.volvol, 1.2)):
        pf, _ = sv.logsv_mc_chain_pricer_fixed_randoms(
            ttms=ttms[:1], forwards=fw[:1], discfactors=df[:1], strikes_ttms=strikes[:1], optiontypes_ttms=types[:1],
            W0s=[W0], W1s=[W1], dts=[0.1 / 12], v0=P_.sigma0, theta=P_.theta, kappa1=P_.kappa1, kappa2=P_.kappa2,
            beta=P_.beta, volvol=volvol, vol_backbone_etas=np.ones(1))
        np.testing.assert_array_equal(pf[0], out["fixed_prices"][3 * it:3 * it + 3])
    from stochvolmodels_amd.data.option_chain import infer_black_ivols
    iv = infer_black_ivols(np.array(out["fixed_prices"][1:2]), 0.1, 1.0, np.array([1.0]), ["C"], 0.99)
    np.testing.assert_allclose(out["atm_call_ivol"], iv, rtol=1e-12)


def test_reference_heston_and_logsv_mc_ci_tests(sv):
    """the reference's own end-to-end MC tests, run verbatim against this package:
    tests/test_heston_characterization.py:269-292 (analytic Heston inside the seeded 40 000-path MC confidence interval)
    and tests/test_logsv_characterization.py:605-635 (un-fixed LogSV MC, 512 paths: finite, non-negative)"""
    chain = sv.OptionChain.slice_to_chain(ttm=0.25, forward=1.0, strikes=np.array([0.9, 1.0, 1.1]),
                                          optiontypes=np.array(["P", "C", "C"]), discfactor=0.98, id="3m")
    params = sv.HestonParams(v0=0.04, theta=0.05, kappa=2.0, rho=-0.5, volvol=0.4)
    analytic = np.asarray(sv.HestonPricer().price_chain(chain, params)[0])
    sv.set_seed(123)
    mc_prices, mc_errors = sv.HestonPricer().model_mc_price_chain(chain, params, nb_path=40_000)
    mc_prices, mc_errors = np.asarray(mc_prices[0]), np.asarray(mc_errors[0])
    assert np.all(np.isfinite(mc_prices)) and np.all(np.isfinite(mc_errors)) and np.all(mc_errors > 0.0)
    assert np.all(np.abs(analytic - mc_prices) <= 4.0 * mc_errors)

    chain = sv.OptionChain.slice_to_chain(ttm=0.02, forward=1.0, strikes=np.array([0.9, 1.0, 1.1]),
                                          optiontypes=np.array(["P", "C", "C"]), discfactor=0.99, id="short")
    p = sv.LogSvParams(sigma0=0.2, theta=0.22, kappa1=3.0, kappa2=12.0, beta=-0.3, volvol=0.4)
    prices, errors = sv.logsv_mc_chain_pricer(ttms=chain.ttms, forwards=chain.forwards, discfactors=chain.discfactors,
                                              strikes_ttms=chain.strikes_ttms, optiontypes_ttms=chain.optiontypes_ttms,
                                              v0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta,
                                              volvol=p.volvol, vol_backbone_etas=p.get_vol_backbone_etas(chain.ttms),
                                              nb_path=512, nb_steps_per_year=360)
    assert np.all(np.isfinite(prices[0])) and np.all(np.asarray(prices[0]) >= 0.0)
    assert np.all(np.isfinite(errors[0])) and np.all(np.asarray(errors[0]) >= 0.0)


def test_payoff_randomised_against_numpy_semantics(sv, oracle):
    """compute_mc_vars_payoff on 60 random slices -- ragged sizes, all four payoff codes, both variable types, NaN / +-inf
    injected into x and qvar -- against the NumPy restatement of the reference (nanmean / nanstd / where semantics)"""
    rng = np.random.default_rng(2024)
    import warnings
    for trial in range(60):
        n = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 257, 1000, 4097, 20011]))
        x = 0.5 * rng.standard_normal(n) - 0.1
        q = 0.3 * np.exp(0.7 * rng.standard_normal(n))
        if trial % 3 == 1 and n > 3:
            idx = rng.choice(n, size=max(1, n // 50), replace=False)
            x[idx] = rng.choice([np.nan, -np.inf], size=idx.size)          # +inf makes every spot NaN: tested in golden
            q[rng.choice(n, size=max(1, n // 70), replace=False)] = np.nan
        k = int(rng.integers(1, 40))
        vt = int(rng.choice([1, 2]))
        fwd = float(rng.uniform(0.5, 3.0))
        strikes = (fwd if vt == 1 else 0.3) * rng.uniform(0.4, 1.8, k)
        types = rng.choice(["C", "P", "IC", "IP"], size=k)
        df, ttm = float(rng.uniform(0.8, 1.0)), float(rng.uniform(0.05, 2.0))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            epr, esd = oracle.np_payoff(x, q, ttm, fwd, strikes, types, df, vt)
        pr, sd = sv.compute_mc_vars_payoff(x0=x, sigma0=np.ones(n), qvar0=q, ttm=ttm, forward=fwd, strikes_ttm=strikes,
                                           optiontypes_ttm=types, discfactor=df, variable_type=sv.VariableType(vt))
        np.testing.assert_allclose(pr, epr, rtol=1e-12, atol=1e-13, err_msg=f"trial {trial} n={n} vt={vt}")
        np.testing.assert_allclose(sd, esd, rtol=1e-12, atol=1e-13, err_msg=f"trial {trial} n={n} vt={vt}")


@pytest.mark.parametrize("vt", [1, 2])
def test_payoff_every_group_width(sv, oracle, vt):
    """one payoff kernel per group width (1..22 strikes of plain chains, 1..16 with inverse options) and payoff variable:
    every instantiation once, and the widths that spill into a second group, against the NumPy restatement -- with NaN and
    -inf log-returns in the sample, which take the saturated branches of the in-kernel exp"""
    import warnings
    rng = np.random.default_rng(77)
    n = 4099
    x = 0.4 * rng.standard_normal(n) - 0.05
    q = 0.3 * np.exp(0.5 * rng.standard_normal(n))
    x[[5, 900]] = np.nan
    x[[17, 2048]] = -np.inf
    q[33] = np.nan
    fwd, ttm, df = 1.3, 0.75, 0.97
    for kinds, widths in ((["C", "P"], range(1, 28)), (["C", "P", "IC", "IP"], range(1, 20))):
        for k in widths:
            strikes = (fwd if vt == 1 else 0.3) * np.linspace(0.5, 1.6, k)
            types = np.array([kinds[j % len(kinds)] for j in range(k)])
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                epr, esd = oracle.np_payoff(x, q, ttm, fwd, strikes, types, df, vt)
            pr, sd = sv.compute_mc_vars_payoff(x0=x, sigma0=np.ones(n), qvar0=q, ttm=ttm, forward=fwd, strikes_ttm=strikes,
                                               optiontypes_ttm=types, discfactor=df, variable_type=sv.VariableType(vt))
            np.testing.assert_allclose(pr, epr, rtol=1e-12, atol=1e-13, err_msg=f"k={k} kinds={kinds} vt={vt}")
            np.testing.assert_allclose(sd, esd, rtol=1e-12, atol=1e-13, err_msg=f"k={k} kinds={kinds} vt={vt}")


# ---------------------------------------------------------------------------------------------------
# rough LogSV
# ---------------------------------------------------------------------------------------------------
def _rough_inputs(g, tag):
    m = len(g["ttms"])
    sigma0, theta, kappa1, kappa2, beta, orthog = (float(a) for a in g["params"])
    return dict(ttms=g["ttms"], forwards=g["forwards"], discfactors=g["discfactors"],
                strikes_ttms=[g[f"strikes_{i}"] for i in range(m)], optiontypes_ttms=[g[f"types_{i}"] for i in range(m)],
                sigma0=sigma0, theta=theta, kappa1=kappa1, kappa2=kappa2, beta=beta, orthog_vol=orthog,
                weights=g[f"{tag}_weights"], nodes=g[f"{tag}_nodes"])


@pytest.mark.parametrize("tag", ["h010", "h045", "h050"])
def test_rough_logsv_fixed_randoms_vs_reference(sv, golden, tag):
    g = golden("rough")
    kw = _rough_inputs(g, tag)
    nb_path = int(g[f"{tag}_nb_path"])
    Z0, Z1, grids = sv.get_randoms_for_rough_vol_chain_valuation(ttms=kw["ttms"], nb_path=nb_path,
                                                                 nb_steps_per_year=360, seed=10)
    pr, sd = sv.rough_logsv_mc_chain_pricer_fixed_randoms(Z0=Z0, Z1=Z1, timegrids=grids, **kw)
    for i in range(len(kw["ttms"])):
        np.testing.assert_allclose(pr[i], g[f"{tag}_prices_{i}"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(sd[i], g[f"{tag}_stderrs_{i}"], rtol=1e-12, atol=1e-13)
        if tag == "h010":   # the reference's committed regression prices, at the reference's own tolerance
            np.testing.assert_allclose(pr[i], g[f"reference_regression_prices_{i}"], rtol=1e-7)
    # terminal state of the last expiry is still resident
    from stochvolmodels_amd.engine import get_engine
    eng = get_engine(nb_path)
    ls, _, y = eng.get_state()
    np.testing.assert_allclose(ls[:128], g[f"{tag}_log_s_head"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(eng.get_factors(kw["nodes"].size)[:, :128], g[f"{tag}_vol_head"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(y[:128], g[f"{tag}_y_head"], rtol=1e-12, atol=1e-11)


def test_rough_logsv_device_rng_and_pricer_route(sv, oracle, golden):
    g = golden("rough")
    kw = _rough_inputs(g, "h010")
    n, seed = 20000, 777
    sv.set_seed(5)
    pr, sd = sv.rough_logsv_mc_chain_pricer(nb_path=n, nb_steps_per_year=360, seed=seed, **kw)
    # oracle on the same counter-based stream (stream tag 3, call id 0 after set_seed)
    grids = [np.linspace(0.0, t, int(t * 360) + 2) for t in kw["ttms"]]
    Z0, Z1 = oracle.fill_normals(seed, n, grids[-1].size - 1, call_id=0, stream=3)
    po, so = oracle.rough_logsv_chain_fixed_randoms(kw["ttms"], kw["forwards"], kw["discfactors"], kw["strikes_ttms"],
                                                    kw["optiontypes_ttms"], Z0, Z1, kw["sigma0"], kw["theta"],
                                                    kw["kappa1"], kw["kappa2"], kw["beta"], kw["orthog_vol"],
                                                    kw["weights"], kw["nodes"], grids)
    for i in range(len(grids)):
        np.testing.assert_allclose(pr[i], po[i], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(sd[i], so[i], rtol=1e-12, atol=1e-13)
    # Q_VAR payoffs on the same paths, and the two draws statistically consistent with the fixed-randoms golden
    pq, _ = sv.rough_logsv_mc_chain_pricer(nb_path=n, nb_steps_per_year=360, seed=seed,
                                           variable_type=sv.VariableType.Q_VAR,
                                           **{**kw, "strikes_ttms": [np.array([0.3, 0.5])] * 4,
                                              "optiontypes_ttms": [np.array(["C", "P"])] * 4})
    assert all(np.all(np.isfinite(p)) and np.all(p >= 0) for p in pq)
    pqo, _ = oracle.rough_logsv_chain_fixed_randoms(kw["ttms"], kw["forwards"], kw["discfactors"],
                                                    [np.array([0.3, 0.5])] * 4, [np.array(["C", "P"])] * 4, Z0, Z1,
                                                    kw["sigma0"], kw["theta"], kw["kappa1"], kw["kappa2"], kw["beta"],
                                                    kw["orthog_vol"], kw["weights"], kw["nodes"], grids, variable_type=2)
    for i in range(len(grids)):
        np.testing.assert_allclose(pq[i], pqo[i], rtol=1e-12, atol=1e-13)
    for i in range(len(grids)):
        err = np.sqrt(sd[i] ** 2 / n + g[f"h010_stderrs_{i}"] ** 2 / 10000)    # second return = payoff std here
        assert np.all(np.abs(pr[i] - g[f"h010_prices_{i}"]) <= 4.5 * err + 1e-12)

    # LogSVPricer.model_mc_price_chain(use_rough_mc=True, seed=...) reproduces the reference's route (:390-411)
    chain = sv.OptionChain(ttms=kw["ttms"], forwards=kw["forwards"], discfactors=kw["discfactors"],
                           strikes_ttms=tuple(kw["strikes_ttms"]), optiontypes_ttms=tuple(kw["optiontypes_ttms"]),
                           ids=np.array([f"t{i}" for i in range(4)]))
    params = sv.LogSvParams(sigma0=kw["sigma0"], theta=kw["theta"], kappa1=kw["kappa1"], kappa2=kw["kappa2"],
                            beta=kw["beta"], volvol=kw["orthog_vol"], H=0.1, weights=kw["weights"], nodes=kw["nodes"])
    pr2, _ = sv.LogSVPricer().model_mc_price_chain(option_chain=chain, params=params, nb_path=10000, nb_steps=360,
                                                   use_rough_mc=True, seed=10)
    for i in range(4):
        np.testing.assert_allclose(pr2[i], g[f"h010_prices_{i}"], rtol=1e-12, atol=1e-13)
    with pytest.raises(AssertionError):
        sv.LogSVPricer().model_mc_price_chain(option_chain=chain, params=params, nb_path=100, use_rough_mc=True)
    p05 = sv.LogSvParams(H=0.5)
    p05.approximate_kernel(T=1.0)
    assert p05.nodes.tolist() == [1e-3] and p05.weights.tolist() == [1.0]
    # approximate_kernel for H <= 0.49 (the quadrature rule of rough_logsv/rough_kernel.py; host test:
    # tests/test_host_logic.py::test_rough_kernel_quadrature_rule): a drop-in rough chain pricing from H alone
    auto = sv.LogSvParams(sigma0=kw["sigma0"], theta=kw["theta"], kappa1=kw["kappa1"], kappa2=kw["kappa2"],
                          beta=kw["beta"], volvol=kw["orthog_vol"], H=0.1)
    auto.approximate_kernel(T=float(chain.ttms[-1]))
    assert auto.nodes.shape == auto.weights.shape == (3,)
    pr3, _ = sv.LogSVPricer().model_mc_price_chain(option_chain=chain, params=auto, nb_path=10000, nb_steps=360,
                                                   use_rough_mc=True, seed=10)
    if np.allclose(auto.nodes, kw["nodes"], rtol=1e-6) and np.allclose(auto.weights, kw["weights"], rtol=1e-6):
        for i in range(4):                     # the golden case used this very rule: the same prices
            np.testing.assert_allclose(pr3[i], g[f"h010_prices_{i}"], rtol=1e-12, atol=1e-10)
    assert all(np.all(np.isfinite(p_)) and np.all(p_ >= 0.0) for p_ in pr3)


# ---------------------------------------------------------------------------------------------------
# calibration loop: the optimizer runs on the host, every objective evaluation on the GPU
# ---------------------------------------------------------------------------------------------------
def _calibration_chain(sv, g, prefix=""):
    k, ty = g["strikes"], g["types"]
    mids = [g[f"{prefix}mid_0"], g[f"{prefix}mid_1"]]
    return sv.OptionChain(ttms=g["ttms"], forwards=g["forwards"], strikes_ttms=(k, k), optiontypes_ttms=(ty, ty),
                          discfactors=np.ones(2), ids=np.array(["t0", "t1"]),
                          bid_ivs=tuple(m - 0.005 for m in mids), ask_ivs=tuple(m + 0.005 for m in mids))


def _vec(p):
    return np.array([p.sigma0, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol])


@pytest.mark.parametrize("tag", ["mc5", "mc4c", "rough4", "an4"])
def test_logsv_calibration_vs_reference(sv, golden, tag):
    """same chain, start point, fixed randoms, SLSQP options as the reference's run.  The two objectives agree to
    ~1e-9 per evaluation, so the optimizer paths coincide up to its own stopping tolerance (ftol 1e-8 on an
    objective of order 1e-6..1e-4): fitted parameters within 2e-3 relative + 2e-3 absolute, and the objective
    at our optimum no worse than at the reference's."""
    g = golden("calibration")
    chain = _calibration_chain(sv, g)
    CT, CE, KT = sv.LogsvModelCalibrationType, sv.CalibrationEngine, sv.ConstraintsType
    s = g["start"]
    start = dict(sigma0=s[0], theta=s[1], kappa1=s[2], kappa2=s[3], beta=s[4], volvol=s[5])
    kw = {"mc5": dict(calibration_engine=CE.MC, model_calibration_type=CT.PARAMS5, nb_path=4000, nb_steps=360, seed=10),
          "mc4c": dict(calibration_engine=CE.MC, model_calibration_type=CT.PARAMS4,
                       constraints_type=KT.INVERSE_MARTINGALE_MOMENT4, is_vega_weighted=False, nb_path=4000,
                       nb_steps=360, seed=7),
          "rough4": dict(calibration_engine=CE.ROUGH_MC, model_calibration_type=CT.PARAMS4, nb_path=2000, nb_steps=360,
                         seed=10),
          "an4": dict(calibration_engine=CE.ANALYTIC, model_calibration_type=CT.PARAMS4,
                      constraints_type=KT.MMA_MARTINGALE)}[tag]
    if tag == "rough4":
        start.update(H=0.1, nodes=g["rough_nodes"], weights=g["rough_weights"])
    pricer = sv.LogSVPricer()
    fit = pricer.calibrate_model_params_to_chain(option_chain=chain, params0=sv.LogSvParams(**start), disp=False, **kw)
    ref = g[f"{tag}_fit"]
    tol = dict(rtol=2e-3, atol=2e-3) if tag != "an4" else dict(rtol=2e-2, atol=1e-2)   # an4: RK45 rtol 1e-3 in the reference
    np.testing.assert_allclose(_vec(fit), ref, **tol)
    assert pricer.last_calibration["n_eval"] > 5
    if tag == "mc5":
        # the MC engine's gradient from one replay per iterate (all bumped vectors stepped on one pass over the randoms):
        # bit-identical objective values, hence the same fit as with SLSQP's own differencing
        assert pricer.last_calibration["n_gradient_batches"] > 2
        plain = sv.LogSVPricer()
        fit0 = plain.calibrate_model_params_to_chain(option_chain=chain, params0=sv.LogSvParams(**start), disp=False,
                                                     batched_gradient=False, **kw)
        assert plain.last_calibration["n_gradient_batches"] == 0
        np.testing.assert_allclose(_vec(fit), _vec(fit0), rtol=1e-12, atol=1e-12)
    if tag in ("mc5", "rough4"):
        # the same calibration on fixed randoms drawn in HBM instead of by NumPy: another sample of the same estimator
        fit_dev = sv.LogSVPricer().calibrate_model_params_to_chain(option_chain=chain, params0=sv.LogSvParams(**start),
                                                                   disp=False, device_randoms=True, **kw)
        np.testing.assert_allclose(_vec(fit_dev), _vec(fit), rtol=0.25, atol=0.25)       # 4000 paths: MC noise
    if tag == "an4":
        # the analytic engine hands SLSQP its forward-difference gradient from ONE batch of launches per iterate (the bumped
        # parameter vectors advance together, logsv_chain_pricer_batch): same evaluation points as SLSQP's own differences,
        # bit-identical prices, hence the same optimizer path
        assert pricer.last_calibration["n_gradient_batches"] > 2
        n_eval_batched = pricer.last_calibration["n_eval"]
        plain = sv.LogSVPricer()
        fit0 = plain.calibrate_model_params_to_chain(option_chain=chain, params0=sv.LogSvParams(**start), disp=False,
                                                     batched_gradient=False, **kw)
        assert plain.last_calibration["n_gradient_batches"] == 0
        np.testing.assert_allclose(_vec(fit), _vec(fit0), rtol=1e-12, atol=1e-12)
        assert abs(n_eval_batched - plain.last_calibration["n_eval"]) <= 2
    if tag == "mc4c":       # the constraints hold at the optimum
        assert fit.kappa2 - 2.0 * fit.beta >= -1e-8
        assert fit.kappa1 + fit.kappa2 * fit.theta - 1.5 * (fit.beta ** 2 + fit.volvol ** 2) >= -1e-8


@pytest.mark.parametrize("tag", ["an", "mc"])
def test_logsv_varswap_fit_calibration_vs_reference(sv, golden, tag):
    """PARAMS_WITH_VARSWAP_FIT: SLSQP over (beta, volvol) only, the vol backbone refitted to the chain's variance-swap
    strikes for every candidate -- same chain, start point, randoms and optimizer options as the reference's run
    (tests/golden/make_golden.py g_varswap; the host pieces alone are in tests/test_varswap_golden.py)."""
    g = golden("varswap")
    mids = [g[f"chain_mid_{i}"] for i in range(4)]
    chain = sv.OptionChain(ttms=g["chain_ttms"], forwards=g["chain_forwards"],
                           strikes_ttms=tuple(g[f"chain_strikes_{i}"] for i in range(4)),
                           optiontypes_ttms=tuple(g[f"chain_types_{i}"] for i in range(4)), discfactors=np.ones(4),
                           ids=np.array(["t0", "t1", "t2", "t3"]), bid_ivs=tuple(m - 0.005 for m in mids),
                           ask_ivs=tuple(m + 0.005 for m in mids))
    s = g["start"]
    start = sv.LogSvParams(sigma0=s[0], theta=s[1], kappa1=s[2], kappa2=s[3], beta=s[4], volvol=s[5])
    CE = sv.CalibrationEngine
    kw = dict(an=dict(calibration_engine=CE.ANALYTIC), mc=dict(calibration_engine=CE.MC, nb_path=4000, nb_steps=360, seed=10))[tag]
    pricer = sv.LogSVPricer()
    fit = pricer.calibrate_model_params_to_chain(option_chain=chain, params0=start, disp=False,
                                                 model_calibration_type=sv.LogsvModelCalibrationType.PARAMS_WITH_VARSWAP_FIT, **kw)
    tol = dict(rtol=2e-3, atol=2e-3) if tag == "mc" else dict(rtol=2e-2, atol=1e-2)    # an: RK45 rtol 1e-3 in the reference
    np.testing.assert_allclose(_vec(fit), g[f"{tag}_fit"], **tol)
    np.testing.assert_allclose(fit.vol_backbone.to_numpy(), g[f"{tag}_backbone"], **tol)
    assert _vec(fit)[:4].tolist() == s[:4].tolist()                        # sigma0, theta, kappa1, kappa2 are not touched
    assert pricer.last_calibration["n_eval"] > 5


def test_heston_calibration_vs_reference(sv, golden):
    g = golden("calibration")
    chain = _calibration_chain(sv, g, "heston_")
    fit = sv.HestonPricer().calibrate_model_params_to_chain(option_chain=chain, params0=None, disp=False)
    got = np.array([fit.v0, fit.theta, fit.kappa, fit.rho, fit.volvol])
    np.testing.assert_allclose(got, g["heston_fit"], rtol=2e-3, atol=2e-3)
    assert 2.0 * fit.kappa * fit.theta - fit.volvol ** 2 >= -1e-8


def test_calibration_errors(sv, golden):
    g = golden("calibration")
    chain = _calibration_chain(sv, g)
    p0 = sv.LogSvParams()
    with pytest.raises(NotImplementedError):
        sv.LogSVPricer().calibrate_model_params_to_chain(option_chain=chain, params0=p0, disp=False,
                                                         model_calibration_type=sv.LogsvModelCalibrationType.PARAMS6)
    bare = sv.OptionChain(ttms=g["ttms"], forwards=g["forwards"], strikes_ttms=(g["strikes"],) * 2,
                          optiontypes_ttms=(g["types"],) * 2, discfactors=np.ones(2), ids=np.array(["a", "b"]))
    with pytest.raises((ValueError, TypeError)):
        sv.LogSVPricer().calibrate_model_params_to_chain(option_chain=bare, params0=p0, disp=False)


def test_rough_chain_one_launch_equals_expiry_by_expiry(sv):
    """svmc_rough_logsv_chain: all expiries of a rough chain side by side in one launch must return the bits of the
    expiry-by-expiry launches -- supplied randoms and device draw, 1-3 factors, LOG_RETURN and Q_VAR, odd path count"""
    from stochvolmodels_amd.pricers import logsv_pricer as lp
    ttms = np.array([0.05, 0.1, 0.25, 0.4])
    kk = np.linspace(0.8, 1.25, 6)
    n, spy = 5003, 120
    chain = dict(ttms=ttms, forwards=np.array([1.0, 1.005, 1.01, 1.02]), discfactors=np.array([0.999, 0.995, 0.99, 0.98]),
                 strikes_ttms=(kk,) * 4, optiontypes_ttms=(np.where(kk >= 1.0, "C", "P"),) * 4)
    Z0, Z1, grids = sv.get_randoms_for_rough_vol_chain_valuation(ttms, nb_path=n, nb_steps_per_year=spy, seed=4)
    for H in (0.1, 0.45, 0.5):
        p = sv.LogSvParams(sigma0=0.8, theta=1.0, kappa1=3.0, kappa2=3.0, beta=0.15, volvol=1.8, H=H)
        p.approximate_kernel(T=float(ttms[-1]))
        pars = dict(sigma0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta, orthog_vol=p.volvol,
                    weights=p.weights, nodes=p.nodes)
        for vt, strikes in ((sv.VariableType.LOG_RETURN, chain["strikes_ttms"]), (sv.VariableType.Q_VAR, (0.6 * kk,) * 4)):
            kw = dict(chain, strikes_ttms=strikes, variable_type=vt, **pars)
            out = {}
            for flag in (True, False):
                lp.ROUGH_CHAIN_ONE_LAUNCH = flag
                try:
                    out[flag] = (sv.rough_logsv_mc_chain_pricer_fixed_randoms(Z0=Z0, Z1=Z1, timegrids=grids, **kw),
                                 sv.rough_logsv_mc_chain_pricer(nb_path=n, nb_steps_per_year=spy, seed=11, **kw))
                finally:
                    lp.ROUGH_CHAIN_ONE_LAUNCH = True
            for a, b in zip(out[True], out[False]):
                for x, y in zip(a[0] + a[1], b[0] + b[1]):
                    np.testing.assert_array_equal(x, y, err_msg=f"H = {H}, {vt}")
            assert np.all(np.isfinite(np.concatenate(out[True][0][0])))


def test_rough_cabi_continuation_and_errors(sv, oracle, golden):
    """svmc_rough_logsv_terminal called directly: two half-range calls continuing from the resident state equal one
    full-range call bit for bit (device RNG: the step offset carries the counter), and argument errors map to
    ValueError"""
    g = golden("rough")
    nodes, weights = g["h010_nodes"], g["h010_weights"]
    v0 = np.full(3, 0.377 / weights.sum())
    n, nb, h = 4097, 40, 1.0 / 360
    args = (nodes, weights, v0, 0.347, 1.29, 1.93, 0.8, 3.0)
    eng = _engine(n, offset=77)
    eng.rough_logsv(nb, h, *args, seed=9, call_id=2, from_origin=True)
    full = (*eng.get_state(), eng.get_factors(3).copy())
    eng.rough_logsv(nb // 2, h, *args, seed=9, call_id=2, from_origin=True)
    eng.rough_logsv(nb - nb // 2, h, *args, seed=9, call_id=2, step_offset=nb // 2, from_origin=False)
    two = (*eng.get_state(), eng.get_factors(3))
    assert np.array_equal(full[0], two[0]) and np.array_equal(full[2], two[2]) and np.array_equal(full[3], two[3])
    # the same numbers from the CPU twin
    Z0, Z1 = oracle.fill_normals(9, n, nb, call_id=2, path_offset=77, stream=3)
    ls, y = np.zeros(n), np.zeros(n)
    vol = np.ascontiguousarray(np.repeat(v0[:, None], n, axis=1))
    P = oracle._p
    oracle.lib().svo_rough_logsv_terminal_w(n, nb, h, 3, P(np.ascontiguousarray(nodes)), P(np.ascontiguousarray(weights)),
                                            P(v0), 0.347, 1.29, 1.93, 0.8, 3.0, P(ls), P(vol), P(y), P(Z0), P(Z1), n)
    np.testing.assert_allclose(full[0], ls, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(full[3], vol, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(full[2], y, rtol=1e-12, atol=1e-12)
    with pytest.raises(ValueError):
        eng.rough_logsv(nb, h, np.ones(4), np.ones(4), np.ones(4), 0.3, 1.0, 1.0, 0.5, 1.0)        # 4 factors
    with pytest.raises(ValueError):
        eng.rough_logsv(nb, h, *args[:-2], 1.5, 3.0)                                                 # |rho| > 1
    with pytest.raises(ValueError):
        eng.rough_logsv(nb, h, *args, z0_ptr=eng.x.ptr, z1_ptr=None)                                 # Z0 without Z1
    with pytest.raises(NotImplementedError):
        sv.rough_logsv_mc_chain_pricer_fixed_randoms(
            ttms=np.array([0.1]), forwards=np.ones(1), discfactors=np.ones(1), strikes_ttms=(np.ones(1),),
            optiontypes_ttms=(np.array(["C"]),), Z0=np.zeros((40, 8)), Z1=np.zeros((40, 8)), sigma0=0.3, theta=0.3,
            kappa1=1.0, kappa2=1.0, beta=0.1, orthog_vol=1.0, weights=np.ones(4), nodes=np.ones(4),
            timegrids=[np.linspace(0, 0.1, 37)])


def test_logsv_slice_w_equals_terminal_w_plus_reductions(sv):
    """the fused streamed slice (svmc_logsv_slice_w) against svmc_logsv_terminal_w + snapshot + svmc_spot_sums"""
    n, nb = 5000, 33
    p = sv.LOGSV_BTC_PARAMS
    a, b = _engine(n), _engine(n)
    for e in (a, b):
        e.fill_state(0.0, p.sigma0, 0.0)
        e.reserve_snapshots(2)
    wa, wb = a.fill_normals(nb, 5), b.fill_normals(nb, 5)
    sa, _ = a.alloc_sums(2, "spot")
    sb, _ = b.alloc_sums(2, "spot")
    a.logsv_slice_w(nb, 1 / 360, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, 0.9, False, *wa, 1.02, 0, 1, sa)
    b.logsv_w(nb, 1 / 360, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, 0.9, False, *wb)
    b.finish_slice(1.02, 0, 1, sb)
    for x, y in zip(a.get_state(), b.get_state()):
        assert np.array_equal(x, y)
    assert np.array_equal(a.download(a.snapshot_ptr(0), 2 * n), b.download(b.snapshot_ptr(0), 2 * n))
    np.testing.assert_allclose(a.download(sa, 2), b.download(sb, 2), rtol=1e-14)      # different reduction trees


def test_threads_price_concurrently(sv):
    """the Python host from several threads at once: every thread gets its own engine from the cache (state buffers,
    reduction scratch, pinned download buffer), so four threads pricing chains of the SAME size with different seeds, 12
    calls each, return exactly what the same calls return one after the other"""
    import threading
    p = sv.LOGSV_BTC_PARAMS
    ttms = np.array([0.1, 0.25, 0.5])
    kk = np.linspace(0.7, 1.3, 7)
    kw = dict(ttms=ttms, forwards=np.array([1.0, 1.01, 1.02]), discfactors=np.array([0.99, 0.98, 0.97]), strikes_ttms=(kk,) * 3,
              optiontypes_ttms=(np.where(kk >= 1.0, "C", "IP"),) * 3, v0=p.sigma0, theta=p.theta, kappa1=p.kappa1,
              kappa2=p.kappa2, beta=p.beta, volvol=p.volvol, vol_backbone_etas=np.ones(3), nb_path=30_011,
              nb_steps_per_year=200)
    serial = {(t, i): sv.logsv_mc_chain_pricer(seed=1000 * t + i, **kw) for t in range(4) for i in range(12)}
    got, errs = {}, []

    def work(t):
        try:
            for i in range(12):
                got[(t, i)] = sv.logsv_mc_chain_pricer(seed=1000 * t + i, **kw)
        except Exception as exc:                             # noqa: BLE001
            errs.append(exc)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errs, errs
    assert len(got) == 48
    for key, (pr, sd) in serial.items():
        for a, b in zip(pr + sd, got[key][0] + got[key][1]):
            np.testing.assert_array_equal(a, b, err_msg=str(key))


def test_two_engines_on_their_own_streams(sv):
    """the ABI takes one HIP stream per call: two engines driven on two non-blocking streams, launches interleaved
    from the host, give exactly what each gives alone on the default stream"""
    import ctypes as C
    from stochvolmodels_amd import _lib
    from stochvolmodels_amd.engine import HipEngine
    L = _lib.load()
    p = sv.LOGSV_BTC_PARAMS
    n, nb = 1 << 16, 200

    def run(eng, seed):
        eng.fill_state(0.0, p.sigma0, 0.0)
        for k in range(4):                                  # four back-to-back slices of 50 steps
            eng.logsv_rng(nb // 4, 1 / 360, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, 1.0, True, seed, 0, k * (nb // 4))

    ref = []
    for seed in (1, 2):
        e = HipEngine(n)
        run(e, seed)
        ref.append(e.get_state())
        e.close()
    streams = [C.c_void_p(), C.c_void_p()]
    for s in streams:
        assert L.svmc_stream_create(C.byref(s)) == 0
    engines = [HipEngine(n, stream=s) for s in streams]
    for e in engines:
        e.fill_state(0.0, p.sigma0, 0.0)
    for k in range(4):                                      # interleave the two streams' launches
        for e, seed in zip(engines, (1, 2)):
            e.logsv_rng(nb // 4, 1 / 360, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, 1.0, True, seed, 0, k * (nb // 4))
    for e, r in zip(engines, ref):
        for a, b in zip(e.get_state(), r):
            assert np.array_equal(a, b)
        e.close()
    for s in streams:
        assert L.svmc_stream_destroy(s) == 0


@pytest.mark.parametrize("vt", [1, 2])
def test_payoff_sums_chain_equals_per_expiry(sv, vt):
    """svmc_payoff_sums_chain (all expiries in one launch pair, 20 chunks of 8 strikes per launch) gives the bits of
    per-expiry svmc_payoff_sums: 9 expiries with ragged strike counts (0, 1, 8, 9, 21, 33, ...: 26 chunks -> two
    launches), all four payoff codes"""
    from stochvolmodels_amd.engine import option_type_codes, payoff_shifts
    rng = np.random.default_rng(5)
    n, m = 20001, 9
    counts = [21, 0, 1, 8, 9, 33, 21, 16, 40]
    eng = _engine(n)
    eng.reserve_snapshots(2 * m)
    p = sv.LOGSV_BTC_PARAMS
    eng.fill_state(0.0, p.sigma0, 0.0)
    forwards = 1.0 + 0.01 * np.arange(m)
    ttms = 0.05 * (1 + np.arange(m))
    spot_ptr, _ = eng.alloc_sums(2 * m, "spot")
    for i in range(m):
        eng.logsv_slice_rng(7, 0.05 / 7, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, 1.0, True, 3, 0, 7 * i,
                            float(forwards[i]), i, m + i, spot_ptr + 16 * i)
    strikes = [np.sort(rng.uniform(0.02, 0.3, c)) if vt == 2 else f * np.sort(rng.uniform(0.6, 1.5, c))
               for c, f in zip(counts, forwards)]
    types = [rng.choice(["C", "P"] if vt == 2 else ["C", "P", "IC", "IP"], c) for c in counts]
    codes = [option_type_codes(t) if len(t) else np.zeros(0, dtype=np.int8) for t in types]
    shifts = [payoff_shifts(k, c, float(f), vt) for k, c, f in zip(strikes, codes, forwards)]
    total = sum(counts)
    a_ptr, _ = eng.alloc_sums(3 * total + 1, "a")
    b_ptr, _ = eng.alloc_sums(3 * total + 1, "b")
    off = 0
    for i in range(m):
        if counts[i]:
            eng.payoff_sums(eng.snapshot_ptr(i), eng.snapshot_ptr(m + i) if vt == 2 else None, float(forwards[i]),
                            float(ttms[i]), spot_ptr + 16 * i, strikes[i], codes[i], shifts[i], vt, a_ptr + 8 * off)
        off += 3 * counts[i]
    eng.payoff_sums_chain(range(m), range(m, 2 * m) if vt == 2 else None, forwards, ttms, spot_ptr, strikes, codes, shifts,
                          vt, b_ptr)
    a, b = eng.download(a_ptr, 3 * total), eng.download(b_ptr, 3 * total)
    assert np.array_equal(a, b)
    assert np.all(a[2::3] == n)                       # every path counted for every strike
    eng.close()


@pytest.mark.parametrize("m,vt,spot", [(8, 1, True), (19, 2, False), (1, 1, True)])
def test_whole_chain_stepping_equals_slice_by_slice(sv, m, vt, spot):
    """svmc_logsv_chain_rng (all expiries in one stepping launch, 16 slices per launch) against the slice-by-slice
    launches: same prices, stderrs and terminal state, bit for bit -- ragged step counts, vol backbone, both
    measures, quadratic-variance payoffs"""
    from stochvolmodels_amd.engine import get_engine
    from stochvolmodels_amd.pricers import logsv_pricer as lp
    rng = np.random.default_rng(m)
    ttms = np.cumsum(rng.uniform(0.01, 0.08, m))
    fw = 1.0 + 0.02 * rng.standard_normal(m)
    kk = [np.array([0.05, 0.2, 0.5]) if vt == 2 else f * np.array([0.8, 1.0, 1.2]) for f in fw]
    ty = [np.array(["C", "P", "C"]) if vt == 2 else np.array(["P", "IC", "C"]) for _ in range(m)]
    p = sv.LOGSV_BTC_PARAMS
    kw = dict(ttms=ttms, forwards=fw, discfactors=np.full(m, 0.99), strikes_ttms=kk, optiontypes_ttms=ty, v0=p.sigma0,
              theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta, volvol=p.volvol,
              vol_backbone_etas=1.0 + 0.1 * rng.uniform(-1, 1, m), is_spot_measure=spot, nb_path=30001,
              nb_steps_per_year=333, variable_type=sv.VariableType(vt), seed=99)
    out = {}
    for flag in (True, False):
        lp.WHOLE_CHAIN_STEPPING = flag
        try:
            pr, sd = sv.logsv_mc_chain_pricer(**kw)
        finally:
            lp.WHOLE_CHAIN_STEPPING = True
        out[flag] = (np.concatenate(pr), np.concatenate(sd), get_engine(30001).get_state())
    np.testing.assert_array_equal(out[True][0], out[False][0])
    np.testing.assert_array_equal(out[True][1], out[False][1])
    for a, b in zip(out[True][2], out[False][2]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("scheme,vt", [("euler", 1), ("qe", 2), ("qe", 1)])
def test_heston_whole_chain_stepping_equals_slice_by_slice(sv, scheme, vt):
    """svmc_heston_chain_rng against slice-by-slice svmc_heston_slice_rng: identical bits (18 expiries -> two launches)"""
    from stochvolmodels_amd.engine import get_engine
    from stochvolmodels_amd.pricers import heston_pricer as hp
    m = 18
    rng = np.random.default_rng(3)
    ttms = np.cumsum(rng.uniform(0.01, 0.06, m))
    fw = 1.0 + 0.02 * rng.standard_normal(m)
    kk = [np.array([0.01, 0.04, 0.09]) if vt == 2 else f * np.array([0.8, 1.0, 1.2]) for f in fw]
    ty = [np.array(["C", "P", "C"]) if vt == 2 else np.array(["IP", "C", "IC"]) for _ in range(m)]
    kw = dict(ttms=ttms, forwards=fw, discfactors=np.full(m, 0.98), strikes_ttms=kk, optiontypes_ttms=ty, v0=0.05, theta=0.04,
              kappa=3.0, rho=-0.6, volvol=0.7, nb_path=20011, scheme=scheme, nb_steps_per_year=250,
              variable_type=sv.VariableType(vt), seed=5)
    out = {}
    for flag in (True, False):
        hp.WHOLE_CHAIN_STEPPING = flag
        try:
            pr, sd = sv.heston_mc_chain_pricer(**kw)
        finally:
            hp.WHOLE_CHAIN_STEPPING = True
        out[flag] = (np.concatenate(pr), np.concatenate(sd), get_engine(20011).get_state())
    np.testing.assert_array_equal(out[True][0], out[False][0])
    np.testing.assert_array_equal(out[True][1], out[False][1])
    for a, b in zip(out[True][2], out[False][2]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("tag", ["base", "btc"])
def test_heston_analytic_qvar_vs_reference(sv, golden, tag):
    """heston_chain_pricer(variable_type=Q_VAR): closed-form MGF on the 40 000-point psi grid (heston_mgf_grid_kernel)
    and the quadratic-variance call transform (mgf_qvar_slice_kernel) against the reference; MC agrees within 4 stderr"""
    g = golden("heston_qvar")
    v0, theta, kappa, rho, volvol = (float(a) for a in g[f"{tag}_params"])
    kk, ty = g[f"{tag}_strikes"], g[f"{tag}_types"]
    kw = dict(v0=v0, theta=theta, kappa=kappa, volvol=volvol, rho=rho, ttms=g["ttms"], forwards=g["forwards"],
              strikes_ttms=(kk,) * 3, optiontypes_ttms=(ty,) * 3, discfactors=g["discfactors"])
    pr = sv.heston_chain_pricer(variable_type=sv.VariableType.Q_VAR, **kw)
    np.testing.assert_allclose(np.stack(pr), g[f"{tag}_prices"], rtol=1e-10, atol=1e-12)
    chain = sv.OptionChain(ttms=g["ttms"], forwards=g["forwards"], strikes_ttms=(kk,) * 3, optiontypes_ttms=(ty,) * 3,
                           discfactors=g["discfactors"], ids=np.array(["a", "b", "c"]))
    pr2 = sv.HestonPricer().price_chain(chain, sv.HestonParams(v0=v0, theta=theta, kappa=kappa, rho=rho, volvol=volvol),
                                        variable_type=sv.VariableType.Q_VAR)
    np.testing.assert_array_equal(np.stack(pr2), np.stack(pr))
    if tag == "base":
        # Monte Carlo agrees.  (Not asserted for BTC_HESTON_PARAMS: there the reference's own transform breaks down --
        # exp(zeta) overflows on the psi grid, its prices hit the 1e-10 floor and sit 30 % below Monte Carlo; the GPU
        # path reproduces the reference's numbers, which is what parity asks for.)
        # the reference's scale (40 000 paths) and criterion (4 stderr, tests/test_logsv_characterization.py:407), nothing added
        mc, sd = sv.heston_mc_chain_pricer(nb_path=40_000, variable_type=sv.VariableType.Q_VAR, scheme="qe",
                                           nb_steps_per_year=720, seed=8, **kw)
        for i in range(3):
            assert np.all(np.abs(mc[i] - pr[i]) <= 4.0 * sd[i]), (i, np.abs(mc[i] - pr[i]) / sd[i])
    with pytest.raises(ValueError):                # the reference prices calls only on this variable
        sv.heston_chain_pricer(variable_type=sv.VariableType.Q_VAR, **dict(kw, optiontypes_ttms=(np.array(["P"] * 8),) * 3))


def test_price_slice_and_vanilla_consistent_with_chain(sv):
    """ModelPricer.price_slice / price_vanilla (reference model_pricer.py:156-195; its own consistency test is
    tests/test_logsv_characterization.py:101-143): one slice / one option through the chain pricer"""
    kk = np.array([0.9, 1.0, 1.1])
    ty = np.array(["P", "C", "C"])
    for pricer, params in ((sv.LogSVPricer(), sv.LOGSV_BTC_PARAMS), (sv.HestonPricer(), sv.HestonParams())):
        chain = sv.OptionChain.slice_to_chain(ttm=0.25, forward=1.0, strikes=kk, optiontypes=ty, discfactor=0.99)
        ref = pricer.price_chain(option_chain=chain, params=params)[0]
        pr, iv = pricer.price_slice(params=params, ttm=0.25, forward=1.0, strikes=kk, optiontypes=ty, discfactor=0.99)
        np.testing.assert_array_equal(pr, ref)
        assert np.all(np.isfinite(iv)) and np.all(iv > 0)
        p1, v1 = pricer.price_vanilla(params=params, ttm=0.25, forward=1.0, strike=1.0, optiontype="C", discfactor=0.99)
        np.testing.assert_allclose([p1, v1], [pr[1], iv[1]], rtol=1e-12)


def test_reference_quickstart_known_answers(sv):
    """the reference's offline quickstart (examples/getting_started/quickstart.py:11-47), statement for statement on
    this package, against the four numbers it asserts itself (rtol 5e-6): prices AND Black implied vols -- the only
    published anchors for the price -> vol inversion the reference delegates to a third-party package"""
    params = sv.LogSvParams(sigma0=1.0, theta=1.0, kappa1=5.0, kappa2=5.0, beta=0.2, volvol=2.0)
    pricer = sv.LogSVPricer()
    vanilla_price, vanilla_ivol = pricer.price_vanilla(params=params, ttm=0.25, forward=1.0, strike=1.0, optiontype="C")
    chain = sv.OptionChain.get_uniform_chain(ttms=np.array([0.25, 0.5]), ids=np.array(["3m", "6m"]),
                                             forwards=np.array([1.0, 1.0]), strikes=np.array([0.8, 0.9, 1.0, 1.1, 1.2]))
    chain_prices, chain_ivols = pricer.compute_chain_prices_with_vols(option_chain=chain, params=params)
    assert all(np.all(np.isfinite(v)) for v in (*chain_prices, *chain_ivols))
    assert [v.shape for v in chain_prices] == [(5,), (5,)] and [v.shape for v in chain_ivols] == [(5,), (5,)]
    np.testing.assert_allclose(vanilla_price, 0.197331, rtol=5.0e-6, atol=1.0e-8)
    np.testing.assert_allclose(vanilla_ivol, 0.999577, rtol=5.0e-6, atol=1.0e-8)
    np.testing.assert_allclose(chain_prices[1][2], 0.275202, rtol=5.0e-6, atol=1.0e-8)
    np.testing.assert_allclose(chain_ivols[1][2], 0.995757, rtol=5.0e-6, atol=1.0e-8)


def test_randomised_chain_sweep(sv, oracle):
    """tests/fuzz_parity.py inside the suite: 40 random chains -- ragged strike counts (1..11 per expiry, 1..6
    expiries), all payoff codes, both measures, both payoff variables, vol backbones, odd path counts, LogSV / Heston
    Euler / QE -- priced by the whole-chain kernels, slice by slice (bit-identical) and by the CPU oracle on the same
    counter-based randoms (1e-8 relative, identical NaN / inf pattern)"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(root, "tests", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    assert fuzz.main(n_cases=40, seed=20240927) < 1e-8


def test_bench_line_two_ranks(tmp_path):
    """the driver's N > 1 invocation of bench.py in its PLAIN form -- `python bench.py --gpus 2`, no launcher around it:
    bench.py starts its own ranks under torch.distributed.run -- on the hardware there is: two ranks sharing this GPU
    (gloo; RCCL refuses two ranks on one device).  The line must be the C4 workload (2^21 paths per GPU, 8 x 128 steps,
    8 x 21 strikes) and carry roofline (from the loaded library's own instruction histogram, not stale), cpu_baseline, the
    N = 1 share rate and the whole-job-on-one-GPU leg, the RCCL-route leg (here: its refusal, reported) and the stream-
    ordered and host-synchronised collectives must have produced identical bits."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SVMC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--cpu-sample-paths", "4096"], capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    line = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["scaling"] == "weak" and line["dtype"] == "f64"
    assert line["self_launched"] is True and line["gc_frozen_before_timed_region"] is True
    assert line["config"]["workload"].startswith("C4") and line["config"]["paths_per_gpu"] == 1 << 21
    assert line["config"]["paths_total"] == 1 << 22 and line["config"]["time_steps"] == 1024 and line["config"]["strikes"] == 168
    assert line["value"] > 0 and line["unit"] == "path-steps/s"
    r = line["roofline"]
    assert r["kernel"] == "logsv_chain_rng_kernel" and r["bound"] == "valu_issue" and 0.0 < r["frac"] < 1.0
    assert r["stale"] is False and r["frac"] <= r["frac_in_stream_int32_cost"] < 1.05 and 40 < r["insts_per_wave_step"] < 60
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
    assert line["n1_share_value"] > 0 and line["c4_full_one_gpu"]["paths"] == 1 << 22
    assert line["stream_ordered_equals_strict_sync"] is True and line["comm"] == "TorchComm" and line["backend"] == "gloo"
    # two ranks on one device: the RCCL leg says why it did not run; with a GPU per rank it carries a rate
    rr = line["rccl_route"]
    assert ("skipped" in rr) or (rr["value"] > 0 and line["rccl_ranks_seen"] == 2 and rr["prices_equal_torch_route"] is True)
    _check_self_proving_fields(line, 2)


def _check_self_proving_fields(line, world):
    """what makes an N > 1 line prove itself (bench.py docstring): the sharded job equals the whole job on one device to
    reduction-order rounding, every rank returned the same bits, the group saw every rank, the kernel time of every rank"""
    sc = line["self_check"]
    assert "self_check_failed" not in line, line["self_check_failed"]
    assert sc["ranks_agree_bitwise"] is True and sc["tolerance"] == 1e-12
    assert 0.0 <= line["sharded_vs_one_gpu_max_rel_dev"] <= 1e-12 and 0.0 <= sc["sharded_vs_one_gpu_max_rel_dev_stderr"] <= 1e-12
    assert line["group_ranks_seen"] == world
    k = line["kernel_ms_over_ranks"]
    assert len(k["per_rank"]) == world and 0.0 < k["min"] <= k["max"]
    r = line["roofline"]
    assert 500.0 < r["clock_mhz_in_kernel"] <= 2500.0 and r["frac"] <= r["frac_at_sustained_clock"] < 1.05


def test_bench_line_one_gpu():
    """the default single-GPU line (C2) at a few steps: the fused C driver's leg (bit-equal prices, interleaved with the Python
    route), the clock measured inside the kernel and the fraction against it, the counters only when they are this library's"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SVMC_DIST_BACKEND")}
    env["SVMC_BENCH_PREWARM"] = "3"
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-sample-paths",
                          "4096", "--no-streamed", "--no-extra-legs"], capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    line = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["workload"].startswith("C2") and line["config"]["paths_total"] == 1 << 20
    c = line["c_abi_route"]
    assert c["prices_equal_python_route"] is True and c["max_rel_dev_vs_python_route"] == 0.0 and c["calls"] >= 3
    assert 0.5 < c["c_over_python_interleaved"] < 1.5 and c["value"] > 1e10      # (wall-clock ratios: sanity bounds, not a perf gate)
    r = line["roofline"]
    assert r["kernel"] == "logsv_rng_kernel" and r["stale"] is False and 500.0 < r["clock_mhz_in_kernel"] <= 2500.0
    assert r["frac"] <= r["frac_at_sustained_clock"] < 1.0 and r["frac_at_sustained_clock"] <= r["frac_in_stream_at_sustained_clock"] < 1.1
    assert ("counters" in r) == (r["insts_per_wave_step_counters"] is not None)
    assert line["cpu_baseline"]["kind"] == "port" and "self_check_failed" not in line


def test_bench_line_eight_ranks_share_one_gpu():
    """the rehearsal of the driver's 8-GPU run on the hardware there is: `python bench.py --gpus 8` starts its own eight
    ranks (gloo: they share this GPU), 8-way shard_range, eight local ranks -> devices, eight torch.distributed.run
    children; the line must carry the self-proving fields and the command must return 0"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SVMC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", SVMC_BENCH_PREWARM="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--paths-per-gpu", "131072",
                          "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                         timeout=900, cwd=root)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    line = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["config"]["paths_total"] == 8 * 131072 and line["config"]["parallelism"] == "path-sharded x8"
    assert line["self_launched"] is True and line["backend"] == "gloo" and "skipped" in line["rccl_route"]
    assert line["n1_share_value"] > 0 and line["c4_full_one_gpu"]["paths"] == 8 * 131072
    _check_self_proving_fields(line, 8)
    print("8 ranks on one GPU: sharded vs one-GPU job", line["sharded_vs_one_gpu_max_rel_dev"],
          "kernel ms over ranks", line["kernel_ms_over_ranks"])


def test_bench_fails_fast_when_a_rank_never_arrives():
    """a rank that hangs before the rendezvous (fault injection) must end the command with a non-zero status and that
    rank's message on stderr within its deadline -- not hang until the driver's timeout"""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SVMC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", SVMC_BENCH_PREWARM="1",
               SVMC_BENCH_FAULT="hang_init:1", SVMC_BENCH_INIT_TIMEOUT="8", SVMC_BENCH_RENDEZVOUS_TIMEOUT="30")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--paths-per-gpu", "4096", "--steps", "1",
                          "--warmup", "0", "--no-cpu-baseline", "--no-extra-legs"], capture_output=True, text=True, env=env,
                         timeout=600, cwd=root)
    took = time.time() - t0
    assert run.returncode != 0, run.stdout[-1000:]
    assert "did not finish within" in run.stderr and "rank 1" in run.stderr, run.stderr[-3000:]
    assert took < 300, took


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 8` on a box with fewer GPUs says so instead of hanging in a rendezvous"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SVMC_DIST_BACKEND")}
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], capture_output=True, text=True,
                         env=env, timeout=300, cwd=root)
    assert run.returncode != 0 and "GPU(s) visible" in (run.stdout + run.stderr)


def _run_bench(argv, env_extra, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "SVMC_DIST_BACKEND")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, capture_output=True, text=True, env=env,
                         timeout=timeout, cwd=root)
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    return run, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize("fault", ["nccl_init", "nccl_hang"])
def test_bench_ladder_lands_on_gloo_with_a_number(fault):
    """first contact with a node whose RCCL does not come up must still produce a line: two ranks (sharing this GPU), the nccl
    rung failing in its probe children -- by an exception ("nccl_init") or by HANGING until the parent kills the child
    ("nccl_hang") -- the rccl rung failing on its own (two ranks on one device), the job running on the gloo rung: exit status 0,
    a number, the rung and the reasons in the line, the self-proving fields intact, and the single-process route beside it"""
    run, line = _run_bench(["--gpus", "2", "--paths-per-gpu", "65536", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                           dict(SVMC_BENCH_SHARE_DEVICES="1", SVMC_DIST_FORCE_PROBE="1", SVMC_BENCH_FAULT=fault, SVMC_BENCH_PREWARM="1",
                                SVMC_BENCH_INIT_TIMEOUT="20" if fault == "nccl_hang" else "90"))
    assert run.returncode == 0 and line is not None, run.stdout[-1500:] + run.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["comm"] == "TorchComm" and line["backend"] == "gloo"
    lad = line["comm_ladder"]
    assert lad["rung"] == "gloo" and lad["control_plane"] == "gloo"
    assert [p[0] for p in lad["probes"]] == ["nccl", "rccl", "gloo"] and [p[1] for p in lad["probes"]] == [False, False, True]
    why = line["comm_fallback_reason"]
    assert "nccl: rank 0:" in why and "rccl: rank 0:" in why
    assert ("fault injection" in why) if fault == "nccl_init" else ("did not finish within 20 s (killed)" in why)
    _check_self_proving_fields(line, 2)
    sp = line["single_process_route"]
    assert "error" not in sp, sp
    assert sp["reduce"] == "host" and sp["devices"] == [0, 0] and sp["shards_agree_bitwise"] is True and sp["value"] > 0
    assert 0.0 <= sp["max_rel_dev_vs_launched_ranks"] <= 1e-12


def test_bench_single_process_mode():
    """`python bench.py --gpus N --single-process`: one process, N shards (here on the one device), the C4 workload, the
    host transport picked because the shards share a device; eight shards at a reduced path count"""
    run, line = _run_bench(["--gpus", "8", "--single-process", "--devices", "0,0,0,0,0,0,0,0", "--paths-per-gpu", "131072", "--steps", "3",
                            "--warmup", "1"], dict(SVMC_BENCH_PREWARM="1"))
    assert run.returncode == 0 and line is not None, run.stdout[-1500:] + run.stderr[-3000:]
    assert line["n_gpus"] == 8 and line["config"]["workload"].startswith("C4") and line["config"]["paths_total"] == 8 * 131072
    assert line["comm"] == "single-process multi-session" and line["reduce"] == "host" and "share a device" in line["reduce_fallback_reason"]
    assert line["shards_agree_bitwise"] is True and len(line["shard_ms_last_call"]) == 8 and line["value"] > 0
    assert len(line["prices_seed_777"]) == 168
    run, _ = _run_bench(["--gpus", "64", "--single-process"], {})
    assert run.returncode != 0 and "GPU(s) visible" in (run.stdout + run.stderr)


def test_bench_default_line_ends_with_the_secondary_block():
    """the driver's N = 1 line: C2 in front, and as its LAST key `secondary` -- C1, C3 (Euler / QE x both sets), C5 (analytic batch,
    2^23-path Monte Carlo per set, the 4-stderr verdict), the frozen-randoms calibration objective and C2 at 2^21 paths -- every
    parity scalar at rounding level, the whole block in seconds"""
    run, line = _run_bench(["--steps", "5", "--warmup", "2", "--cpu-sample-paths", "4096"], dict(SVMC_BENCH_PREWARM="3"))
    assert run.returncode == 0 and line is not None, run.stdout[-1500:] + run.stderr[-3000:]
    assert list(line)[-1] == "secondary" and "roofline_valu_flop_estimate" not in line
    sec = line["secondary"]
    assert "error" not in sec, sec
    assert sec["c1"]["dev"] <= 1e-11 and sec["c1"]["psps"] > 1e9
    for tag in ("euler_base", "euler_btc", "qe_base", "qe_btc"):
        assert sec["c3"][tag]["dev"] <= 1e-10 and sec["c3"][tag]["psps"] > 1e11, (tag, sec["c3"][tag])
    c5 = sec["c5"]
    assert len(c5["mc_ms"]) == 5 and max(c5["dev"]) <= 1e-10 and c5["analytic_dev_btc"] <= 1e-6 and c5["analytic_batch_ms"] < 20.0
    # at 2^23 paths the 4-stderr band is narrower than the truncation error of the reference's second-order expansion for part of
    # every chain (profiles/r02_c5_bias.json): the counts are the line's information, not a gate; the kappa2 = 12 set fails most
    assert all(0 <= v <= 84 for v in c5["pass_of_84"]) and c5["pass_of_84"][3] == min(c5["pass_of_84"]) < 84
    f3 = sec["f3_frozen"]
    assert f3["bit_equal_to_mc_chain_pricer"] is True and f3["hbm_bytes_for_randoms"] == 0 and f3["one_set_with_ivols_ms"] < 1.0
    assert sec["c2_at_2e21_paths"]["psps"] > 0.9 * line["value"]
    assert sec["seconds"] < 30.0
    assert len(__import__("json").dumps(sec)) < 2000                 # the block fits the part of the line a truncating log keeps


@pytest.mark.parametrize("launcher", ["plain", "torchrun"])
@pytest.mark.parametrize("rungs,want", [("nccl,rccl,gloo", "nccl"), ("rccl,gloo", "rccl"), ("gloo", "gloo")])
def test_fallback_ladder_rungs_on_hardware_with_one_rank(tmp_path, rungs, want, launcher):
    """every rung of dist.init_with_fallback brought up for real on this box with ONE rank (two ranks cannot share a device under
    RCCL): the probe child of the rung (its own rendezvous, an RCCL communicator, one all-reduce), then the rung in the process
    itself -- torch's nccl group beside the gloo control plane, or libsvmc's ncclCommInitRank -- and a chain priced through it
    equals the ungrouped result bit for bit.  launcher = "torchrun": the same under `python -m torch.distributed.run` -- the
    driver's form -- whose environment (TORCHELASTIC_USE_AGENT_STORE: "the launcher serves the rendezvous store") must not leak into
    the probe children, which rendezvous on a port of their own (they would wait for a store nobody serves, and a healthy RCCL
    would be reported as hanging)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import json, os, sys, numpy as np\n"
        f"sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})\n"
        "import stochvolmodels_amd as sv\n"
        "from stochvolmodels_amd import dist\n"
        "from cases import LOGSV_CASE\n"
        "base, _ = sv.logsv_mc_chain_pricer(**LOGSV_CASE)\n"
        "comm, rep = dist.init_with_fallback(probe_timeout=150)\n"
        "got, _ = sv.logsv_mc_chain_pricer(**LOGSV_CASE)\n"
        "seen = comm.ranks_seen() if hasattr(comm, 'ranks_seen') else None\n"
        "print(json.dumps({'rep': rep, 'equal': bool(all(np.array_equal(a, b) for a, b in zip(got, base))), 'seen': seen}))\n"
        "import torch.distributed as td\n"
        "td.destroy_process_group()\n")
    env = {k: v for k, v in os.environ.items() if k not in ("SVMC_DIST_BACKEND", "RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR",
                                                            "MASTER_PORT")}
    env.update(SVMC_DIST_SINGLE_RANK_GROUP="1", SVMC_DIST_RUNGS=rungs, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = str(29731 + len(rungs) + (40 if launcher == "torchrun" else 0))
    if launcher == "torchrun":
        script = tmp_path / "ladder_one_rank.py"
        script.write_text(code)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", port, str(script)]
    else:
        env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        cmd = [sys.executable, "-c", code]
    run = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-3000:]
    out = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    rep = out["rep"]
    assert rep["rung"] == want and rep["comm_fallback_reason"] is None and rep["control_plane"] == "gloo", rep
    assert rep["comm"] == ("RcclComm" if want == "rccl" else "TorchComm") and rep["probes"][0]["ok"] is True
    assert out["equal"] is True and (out["seen"] == 1 if want == "rccl" else True)
