"""
Host-side mirror of the reference API: types, validation, the step-count rules, seeding, error conventions.
No GPU needed.  Reference behaviours cited per test.
"""
import numpy as np
import pandas as pd
import pytest

import stochvolmodels_amd as sv
from stochvolmodels_amd.engine import option_type_codes, payoff_shifts
from stochvolmodels_amd.mc_chain import variable_type_code
from stochvolmodels_amd.utils import funcs


def test_enums_match_reference_values():
    # utils/config.py:8-23
    assert [t.value for t in sv.OptionType] == ["C", "P", "IC", "IP"]
    assert (sv.VariableType.LOG_RETURN.value, sv.VariableType.Q_VAR.value, sv.VariableType.SIGMA.value) == (1, 2, 3)


def test_time_grid_rule(golden):
    # utils/funcs.py:44-47; anchors SURVEY.md 8(a1)
    assert sv.set_time_grid(1.0, 1023)[:2] == (1024, 2.0 ** -10)
    assert sv.set_time_grid(0.25, 360)[:2] == (91, 0.0027472527472527475)
    assert sv.set_time_grid(0.02, 360)[:2] == (8, 0.0025)
    for ttm, spy, n, dt in golden("time_grid")["cases"]:
        nb, d, grid = sv.set_time_grid(ttm, int(spy))
        assert (nb, d) == (int(n), dt) and grid.shape == (nb + 1,) and grid[0] == 0.0


def test_time_grid_steps_is_the_grid_rule_without_the_grid(golden):
    """the chain drivers' (nb_steps, dt) must be set_time_grid's, bit for bit: the reference-generated cases and a sweep"""
    cases = [(float(t), int(s)) for t, s, _, _ in golden("time_grid")["cases"]]
    rng = np.random.default_rng(3)
    cases += [(float(rng.random() * 3.0), int(rng.integers(1, 4000))) for _ in range(3000)]
    cases += [(k / 8.0, 1016) for k in range(1, 9)] + [(0.125, 1016), (1.0 / 52, 1016), (1e-5, 360), (0.0, 360)]
    for ttm, spy in cases:
        nb, dt, _ = funcs.set_time_grid(ttm, spy)
        assert funcs.time_grid_steps(ttm, spy) == (nb, dt), (ttm, spy)


def test_payoff_finalize_chain_equals_the_per_expiry_calls():
    """svmc_payoff_finalize_chain (all strikes of a chain, one discount factor per strike) against svmc_payoff_finalize
    expiry by expiry: the same arithmetic per strike, hence the same bits -- NaN sums and empty counts included"""
    from stochvolmodels_amd.engine import payoff_finalize, payoff_finalize_chain
    rng = np.random.default_rng(1)
    counts = [13, 1, 21, 5]
    sums = rng.random(3 * sum(counts)) + 1.0
    sums[2::3] = 1e5
    sums[5], sums[3], sums[11] = 0.0, np.nan, 0.0              # 0 / 0 and NaN must come out as NumPy's nanmean gives them
    shifts = [rng.random(k) for k in counts]
    dfs = rng.random(len(counts))
    p, e = payoff_finalize_chain(sums, np.concatenate(shifts), np.repeat(dfs, counts), 1e5)
    lo = 0
    for k, sh, df in zip(counts, shifts, dfs):
        a, b = payoff_finalize(sums[3 * lo:3 * (lo + k)], sh, df, 1e5)
        np.testing.assert_array_equal(p[lo:lo + k], a)
        np.testing.assert_array_equal(e[lo:lo + k], b)
        lo += k


def test_option_codes_and_errors():
    np.testing.assert_array_equal(option_type_codes(np.array(["C", "P", "IC", "IP"])), [0, 1, 2, 3])
    with pytest.raises(ValueError, match="payoff"):            # utils/mc_payoffs.py:84
        option_type_codes(np.array(["C", "XX"]))
    with pytest.raises(NotImplementedError):                    # utils/mc_payoffs.py:69-70
        variable_type_code(sv.VariableType.SIGMA)
    assert variable_type_code(sv.VariableType.Q_VAR) == 2


def test_payoff_shifts_are_intrinsic_at_forward():
    k = np.array([0.8, 1.2, 0.8, 1.2])
    s = payoff_shifts(k, np.array([0, 0, 1, 3], dtype=np.int8), 1.0, 1)
    np.testing.assert_allclose(s, [0.2, 0.0, 0.0, 0.2])
    assert np.all(payoff_shifts(k, np.array([0, 1, 2, 3], dtype=np.int8), 1.0, 2) == 0)


def test_option_chain_validation():
    # data/option_chain.py:147-215
    ok = sv.OptionChain.slice_to_chain(ttm=0.25, forward=1.0, strikes=np.array([0.9, 1.0]), optiontypes=np.array(["P", "C"]),
                                       discfactor=0.98)
    np.testing.assert_allclose(ok.discount_rates, -np.log(0.98) / 0.25)
    assert ok.ids[0] == "0.25"
    with pytest.raises(ValueError):
        sv.OptionChain(ttms=np.array([0.5, 0.25]), forwards=np.ones(2), strikes_ttms=(np.ones(1),) * 2,
                       optiontypes_ttms=(np.array(["C"]),) * 2, ids=None)
    with pytest.raises(ValueError, match="unsupported optiontypes"):
        sv.OptionChain.slice_to_chain(0.25, 1.0, np.array([1.0]), np.array(["X"]))
    with pytest.raises(ValueError):
        sv.OptionChain.slice_to_chain(0.25, 1.0, np.array([-1.0]), np.array(["C"]))
    with pytest.raises(ValueError):
        sv.OptionChain.slice_to_chain(0.25, -1.0, np.array([1.0]), np.array(["C"]))
    uni = sv.OptionChain.get_uniform_chain(ttms=np.array([0.1, 0.2, 0.3]), ids=np.array(["a", "b", "c"]))
    assert uni.forwards.shape == (3,) and list(uni.optiontypes_ttms[0]) == ["P", "C", "C"]
    u21 = sv.OptionChain.to_uniform_strikes(ok, num_strikes=21)
    assert u21.strikes_ttms[0].shape == (21,) and u21.optiontypes_ttms[0][-1] == "C"


def test_logsv_params():
    # pricers/logsv/logsv_params.py:87-98, :143-162
    p = sv.LogSvParams(sigma0=0.2, theta=0.2, kappa1=1.0, kappa2=None)
    assert p.kappa2 == 5.0
    np.testing.assert_array_equal(p.get_vol_backbone_etas(np.array([0.1, 0.5])), np.ones(2))
    p.set_vol_backbone(pd.Series([0.9, 1.1, 1.2], index=[0.25, 0.5, 1.0]))
    np.testing.assert_array_equal(p.get_vol_backbone_etas(np.array([0.1, 0.25, 0.3, 1.0])), [0.9, 0.9, 1.1, 1.2])
    assert sv.LOGSV_BTC_PARAMS.volvol == 1.8458 and sv.BTC_HESTON_PARAMS.volvol == 2.0
    assert sv.HestonParams().kappa == 4.0
    q = sv.LogSvParams.copy(sv.LOGSV_BTC_PARAMS)
    assert q == sv.LOGSV_BTC_PARAMS and q is not sv.LOGSV_BTC_PARAMS
    with pytest.raises(AssertionError):
        sv.LogSvParams(H=0.6)


def test_fixed_randoms_contract():
    # tests/test_logsv_characterization.py:583-602 of the reference
    kw = dict(ttms=np.array([0.25]), nb_path=8, nb_steps_per_year=12, seed=7)
    a0, a1, adt = sv.get_randoms_for_chain_valuation(**kw)
    b0, b1, bdt = sv.get_randoms_for_chain_valuation(**kw)
    np.testing.assert_array_equal(a0[0], b0[0])
    np.testing.assert_array_equal(a1[0], b1[0])
    assert adt == bdt and a0[0].shape == (4, 8)
    np.random.seed(91)
    expected = np.random.random()
    np.random.seed(91)
    sv.get_randoms_for_chain_valuation(**kw)
    assert np.random.random() == expected
    rs = np.random.RandomState(7)
    np.testing.assert_array_equal(a0[0], rs.normal(0, 1, size=(4, 8)))      # W0 first, then W1
    np.testing.assert_array_equal(a1[0], rs.normal(0, 1, size=(4, 8)))


def test_seed_bookkeeping():
    sv.set_seed(5)
    assert funcs.next_rng_call() == (5, 0)
    assert funcs.next_rng_call() == (5, 1)
    assert funcs.next_rng_call(seed=99) == (99, 0)              # explicit seed: pure replay, no state consumed
    assert funcs.next_rng_call() == (5, 2)
    sv.set_seed(5)
    assert funcs.next_rng_call() == (5, 0)


def test_base_class_contract():
    # reference tests/test_model_calibration_contracts.py:141-155: defaults raise NotImplementedError
    class Dummy(sv.ModelPricer):
        pass
    d = Dummy()
    for call in (lambda: d.model_mc_price_chain(None, None), lambda: d.simulate_terminal_values(None),
                 lambda: d.simulate_vol_paths(None)):
        with pytest.raises(NotImplementedError):
            call()


def test_mc_pdf_filters_invalid_paths_and_normalises(capsys):
    """the contract of ModelPricer.get_log_return_mc_pdf (reference tests/test_model_calibration_contracts.py:121-139):
    NaN and out-of-range paths are counted on stdout and left out, the density is finite, non-negative and sums to one;
    a pricer whose simulate_terminal_values returns the Monte Carlo triple (x, vol, qvar) is read through x"""
    class Flat(sv.ModelPricer):
        def simulate_terminal_values(self, params, **kwargs):
            return np.array([-0.2, -0.1, 0.0, 0.1, 0.2, np.nan, np.inf, -np.inf])

    class Triple(sv.ModelPricer):
        def simulate_terminal_values(self, params, **kwargs):
            x = np.array([-0.2, -0.1, 0.0, 0.1, 0.2, np.nan, np.inf, -np.inf])
            return x, np.ones(8), np.zeros(8)
    grid = np.linspace(-0.5, 0.5, 51)
    dens = []
    for cls in (Flat, Triple):
        density = cls().get_log_return_mc_pdf(ttm=0.25, params=None, x_grid=grid, nb_path=8)
        assert np.all(np.isfinite(density)) and np.all(density >= 0.0)
        np.testing.assert_allclose(np.sum(density), 1.0, rtol=0.0, atol=1e-14)
        out = capsys.readouterr().out
        assert "num -inf = 1" in out and "num +inf = 1" in out and "num nans = 1" in out
        dens.append(density)
    np.testing.assert_array_equal(dens[0], dens[1])
    from scipy.stats import gaussian_kde
    want = gaussian_kde(np.array([-0.2, -0.1, 0.0, 0.1, 0.2]))(grid)
    np.testing.assert_allclose(dens[0], want / want.sum(), rtol=1e-14)


def test_lazy_exports():
    assert "logsv_mc_chain_pricer_fixed_randoms" in dir(sv)
    with pytest.raises(AttributeError):
        sv.not_a_symbol


def test_black_implied_vol_round_trip():
    """the host-side Black-76 inversion used by compute_mc_chain_implied_vols (parity with the reference's third-party
    routine is unpinned; this only checks self-consistency)"""
    from stochvolmodels_amd.data.option_chain import black_price, infer_black_ivols
    k = np.array([0.7, 0.9, 1.0, 1.1, 1.4])
    vols = np.array([0.9, 0.6, 0.5, 0.55, 0.8])
    types = np.array(["P", "P", "C", "C", "C"])
    pr = black_price(1.02, k, 0.5, vols, types == "C", 0.97)
    np.testing.assert_allclose(infer_black_ivols(pr, 0.5, 1.02, k, types, 0.97), vols, rtol=1e-12)
    chain = sv.OptionChain.slice_to_chain(0.5, 1.02, k, types, discfactor=0.97)
    np.testing.assert_allclose(chain.compute_model_ivols_from_chain_data([pr])[0], vols, rtol=1e-12)
    assert np.isnan(infer_black_ivols(np.array([2.0]), 0.5, 1.0, np.array([1.0]), np.array(["C"]))[0])
    # inverse quotes: the Black-76 value of (S - K)^+ / S is the vanilla value over the forward -> same vols from price / F
    inv = np.where(types == "C", "IC", "IP")
    np.testing.assert_allclose(infer_black_ivols(pr / 1.02, 0.5, 1.02, k, inv, 0.97), vols, rtol=1e-12)
    chain_inv = sv.OptionChain.slice_to_chain(0.5, 1.02, k, inv, discfactor=0.97)
    np.testing.assert_allclose(chain_inv.compute_model_ivols_from_chain_data([pr / 1.02])[0], vols, rtol=1e-12)
    with pytest.raises(ValueError):
        infer_black_ivols(pr[:1], 0.5, 1.0, k[:1], np.array(["X"]))


def test_validate_optimization_result():
    """reference pricers/model_pricer.py:48-80 (tests/test_model_calibration_contracts.py covers the same cases)"""
    from types import SimpleNamespace as NS
    from stochvolmodels_amd.utils.calibration import CalibrationError, validate_optimization_result
    b = ((0.0, 1.0), (None, 2.0))
    assert validate_optimization_result(NS(success=True, x=[0.5, 1.0], message="ok"), b).tolist() == [0.5, 1.0]
    assert validate_optimization_result(NS(success=True, x=[1.0 + 5e-11, -9.0], message="ok"), b)[0] > 1.0   # slack
    for bad in (NS(success=False, x=[0.5, 1.0], message="boom"), NS(success=True, x=None, message="m"),
                NS(success=True, x=["a", 1.0], message="m"), NS(success=True, x=[0.5], message="m"),
                NS(success=True, x=[[0.5, 1.0]], message="m"), NS(success=True, x=[np.nan, 1.0], message="m"),
                NS(success=True, x=[-1e-3, 1.0], message="m"), NS(success=True, x=[0.5, 2.1], message="m")):
        with pytest.raises(CalibrationError):
            validate_optimization_result(bad, b)


def test_calibration_parser_and_constraints():
    import stochvolmodels_amd as sv
    from stochvolmodels_amd.pricers.logsv_pricer import _calibration_constraints, _calibration_parser
    p0 = sv.LogSvParams(sigma0=0.3, theta=0.4, kappa1=2.0, kappa2=3.0, beta=0.1, volvol=1.0, H=0.3,
                        nodes=np.array([1.0]), weights=np.array([2.0]))
    names, parse = _calibration_parser(sv.LogsvModelCalibrationType.PARAMS4, p0)
    assert names == ("sigma0", "theta", "beta", "volvol")
    p = parse(np.array([0.5, 0.6, -0.2, 0.9]))
    assert (p.sigma0, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, p.H) == (0.5, 0.6, 2.0, 3.0, -0.2, 0.9, 0.3)
    assert p.nodes is p0.nodes and p.weights is p0.weights
    names, parse = _calibration_parser(sv.LogsvModelCalibrationType.PARAMS5, p0)
    p = parse(np.array([0.5, 0.5, 2.0, -0.2, 0.9]))
    assert names == ("sigma0", "theta", "kappa1", "beta", "volvol") and p.kappa2 == 4.0     # kappa1 / theta
    with pytest.raises(NotImplementedError):                     # as the reference
        _calibration_parser(sv.LogsvModelCalibrationType.PARAMS6, p0)
    names_vs, _ = _calibration_parser(sv.LogsvModelCalibrationType.PARAMS_WITH_VARSWAP_FIT, p0)
    assert names_vs == ("beta", "volvol")                        # its parse() is covered in tests/test_varswap_golden.py
    assert _calibration_constraints(parse, sv.ConstraintsType.UNCONSTRAINT) is None
    c = _calibration_constraints(parse, sv.ConstraintsType.INVERSE_MARTINGALE_MOMENT4)
    x = np.array([0.5, 0.5, 2.0, -0.2, 0.9])
    assert c[0]["type"] == "ineq" and np.isclose(c[0]["fun"](x), 4.0 + 0.4)
    assert np.isclose(c[1]["fun"](x), 2.0 + 4.0 * 0.5 - 1.5 * (0.04 + 0.81))
    assert np.isclose(_calibration_constraints(parse, sv.ConstraintsType.MMA_MARTINGALE)["fun"](x), 4.2)


def test_option_chain_calibration_helpers():
    import stochvolmodels_amd as sv
    from stochvolmodels_amd.data.option_chain import black_price, black_vega
    k = np.array([0.9, 1.0, 1.1])
    chain = sv.OptionChain(ttms=np.array([0.25, 1.0]), forwards=np.array([1.0, 1.05]), strikes_ttms=(k, k),
                           optiontypes_ttms=(np.array(["P", "C", "C"]),) * 2, discfactors=np.ones(2),
                           ids=np.array(["a", "b"]), bid_ivs=(np.array([0.3, 0.2, 0.25]),) * 2,
                           ask_ivs=(np.array([0.32, 0.22, 0.27]),) * 2)
    x, y = chain.get_chain_data_as_xy()
    assert x[0] is chain.ttms and np.allclose(y[0], [0.31, 0.21, 0.26])
    assert np.allclose(chain.get_chain_atm_vols(), [0.21, 0.235])
    v = chain.get_chain_vegas()
    h = 1e-5
    fd = (black_price(1.05, k, 1.0, y[1] + h, np.array([False, True, True])) -
          black_price(1.05, k, 1.0, y[1] - h, np.array([False, True, True]))) / (2 * h)
    assert np.allclose(v[1], fd, rtol=1e-6)
    assert np.allclose(chain.get_chain_vegas(is_unit_ttm_vega=True)[0], black_vega(1.0, k, 1.0, y[0]))


def test_native_black_implied_vols_vs_bisection():
    """svmc_black_implied_vols (host routine of libsvmc, no GPU involved) against the NumPy bisection and the truth"""
    from stochvolmodels_amd.data.option_chain import black_ivols_native, black_price, infer_black_ivols
    rng = np.random.default_rng(0)
    n_checked = 0
    for _ in range(200):
        F, T, df = rng.uniform(0.5, 100), rng.uniform(0.01, 3), rng.uniform(0.9, 1)
        k = F * np.exp(rng.normal(0, 0.4, 25))
        ty = rng.choice(["C", "P"], 25)
        vol = rng.uniform(0.02, 3, 25)
        pr = black_price(F, k, T, vol, ty == "C", df)
        a, b = black_ivols_native(pr, T, F, k, ty, df), infer_black_ivols(pr, T, F, k, ty, df)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        otm = (k >= F) == (ty == "C")
        tv = np.where(otm, pr / df, pr / df - np.abs(F - k)) / F       # time value per unit forward
        m = ~np.isnan(a) & (tv > 1e-7)                                  # below that the INPUT price is rounding noise
        np.testing.assert_allclose(a[m], b[m], rtol=1e-8)
        np.testing.assert_allclose(a[m], vol[m], rtol=1e-7)
        n_checked += int(m.sum())
    assert n_checked > 4000
    # far tail: prices down to 1e-300 are still inverted on the log scale
    a = black_ivols_native(np.array([1.3377398071023282e-297 * 50.0]), 1.0, 50.0, np.array([50.0 * np.exp(1.1555)]), ["C"])
    assert np.isfinite(a[0]) and 0.02 < a[0] < 0.05
    # contract: NaN outside the attainable band / for NaN prices
    out = black_ivols_native(np.array([0.0, np.nan, 2.0, 0.05]), 1.0, 1.0, np.array([1.0, 1.0, 1.0, 1.0]), ["C"] * 4)
    assert np.isnan(out[:3]).all() and abs(out[3] - 0.12538) < 1e-4
    # inverse quotes: the vanilla inversion of price x forward; unknown codes raise like the payoffs do
    F, kk, vv = 40.0, np.array([30.0, 40.0, 55.0]), np.array([0.6, 0.5, 0.7])
    pr = black_price(F, kk, 0.75, vv, np.array([False, True, True]), 0.95) / F
    np.testing.assert_allclose(black_ivols_native(pr, 0.75, F, kk, ["IP", "IC", "IC"], 0.95), vv, rtol=1e-9)
    with pytest.raises(ValueError):
        black_ivols_native(np.array([0.1]), 1.0, 1.0, np.array([1.0]), ["XX"])


def test_build_keeps_basic_blocks_aligned():
    """the stepping loop's speed depends on its placement in instruction memory (DESIGN.md section 5, 3.80 vs 4.08 ms
    for identical instructions): the build must keep aligning basic blocks"""
    from stochvolmodels_amd import build
    f = build.flags()
    assert "--align-all-blocks=4" in f and f[f.index("--align-all-blocks=4") - 1] == "-mllvm"
    assert "--offload-arch=gfx950" in f


def test_rough_kernel_quadrature_rule(golden):
    """LogSvParams.approximate_kernel for every H (reference pricers/logsv/logsv_params.py:96-118): the European
    quadrature rule of rough_logsv/rough_kernel.py -- L-BFGS-B over the log-nodes of the L2 error with the optimal
    weights eliminated -- against the reference's own nodes and weights (tests/golden/rough_kernel.npz)"""
    from stochvolmodels_amd.pricers.logsv.logsv_params import LogSvParams
    from stochvolmodels_amd.pricers.rough_logsv.rough_kernel import european_rule, l2_error_optimal_weights
    g = golden("rough_kernel")
    for i, (H, N, T) in enumerate(g["cases"]):
        nodes, weights = european_rule(float(H), int(N), float(T))
        np.testing.assert_allclose(nodes, g[f"nodes_{i}"], rtol=1e-6, err_msg=f"nodes H={H} N={N} T={T}")
        np.testing.assert_allclose(weights, g[f"weights_{i}"], rtol=1e-6, err_msg=f"weights H={H} N={N} T={T}")
        assert nodes.shape == weights.shape == (int(N),) and np.all(nodes > 0) and np.all(weights > 0)
    for j in range(5):
        H, T = g[f"params_H_T_{j}"]
        p = LogSvParams(sigma0=0.8, theta=1.0, kappa1=3.0, kappa2=3.0, beta=0.15, volvol=1.8, H=float(H))
        p.approximate_kernel(T=float(T))
        np.testing.assert_allclose(p.nodes, g[f"params_nodes_{j}"], rtol=1e-6)
        np.testing.assert_allclose(p.weights, g[f"params_weights_{j}"], rtol=1e-6)
        assert p.nodes.size == (1 if H > 0.49 else 2 if H > 0.4 else 3)
    # the analytic gradient of the objective against central differences
    H, T, x = 0.2, 1.0, np.array([0.05, 1.3, 40.0])
    err, grad, _ = l2_error_optimal_weights(H, T, x, want_grad=True)
    for k in range(3):
        h = 1e-5 * x[k]
        up, dn = x.copy(), x.copy()
        up[k] += h
        dn[k] -= h
        fd = (l2_error_optimal_weights(H, T, up)[0] - l2_error_optimal_weights(H, T, dn)[0]) / (2 * h)
        assert abs(fd - grad[k]) <= 1e-4 * abs(grad[k]) + 1e-10          # finite-difference accuracy


def test_bench_workloads_and_roofline_arithmetic():
    """bench.py's host-side pieces (no GPU): the C2 / C4 workloads are BASELINE's, the CPU baseline leg
    runs on a tiny sample, and the VALU-issue roofline is the arithmetic DESIGN.md section 7 states"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import stochvolmodels_amd as sv
    c2, c4 = bench.make_workload("c2", sv), bench.make_workload("c4", sv)
    assert c2["grids"] == [(1024, 2.0 ** -10)] and c2["n_strikes"] == 21 and c2["nb_total"] == 1024
    assert [g[0] for g in c4["grids"]] == [128] * 8 and c4["n_strikes"] == 168 and c4["nb_total"] == 1024
    assert all(abs(g[1] - 0.125 / 128) < 1e-18 for g in c4["grids"])
    np.testing.assert_allclose(c4["forwards"], 67000.0 * np.exp(0.05 * np.arange(1, 9) / 8))
    for k, f, t in zip(c4["strikes"], c4["forwards"], c4["types"]):
        np.testing.assert_allclose(k, f * np.linspace(0.6, 1.6, 21))
        assert list(t) == ["P" if v < f else "C" for v in k]
    base = bench.cpu_baseline(c4, 256, sv.LOGSV_BTC_PARAMS)
    assert base["kind"] == "port" and base["cores"] == 1 and base["value"] > 0 and "8 expiries" in base["sample"]
    # the VALU-issue roof: the time loop's instruction histogram (per trip = two time steps) priced per opcode class
    classes = {"fp64": 71, "int32": 29, "int32_3op": 4, "quarter": 2}
    isa = {"kernels": {"logsv_rng_kernel": {"classes": classes, "valu": 106, "lds": 10}}, "stale": False,
           "source": "stochvolmodels_amd/libsvmc.isa.json", "lib_sha256": "ab" * 32}
    pmc = {"logsv_rng_kernel": {"valu_insts_per_wave_step": 53.2, "config": {"paths": 1 << 20, "steps": 1024}, "hbm_bytes": 59.0e6},
           "matches_loaded_library": True}
    r = bench.kernel_rooflines("logsv_rng_kernel", 2.0, 50, 1 << 20, c2, isa, pmc, 2400.0)["roofline"]
    cyc = (71 * 4 + 29 * 2 + 4 * 4 + 2 * 16) / 2.0                  # 195 issue cycles per wave-step
    want = cyc * (2 ** 20 / 64) * 1024 / 2.0e-3
    assert r["bound"] == "valu_issue" and r["issue_cycles_per_wave_step"] == cyc and abs(r["achieved"] - want) < 1e-9 * want
    assert r["peak"] == 1024 * 2.4e9 and abs(r["frac"] - want / (1024 * 2.4e9)) < 1e-12 and r["insts_per_wave_step"] == 53.0
    in_stream = (71 * 4 + 29 * 3.9 + 4 * 4 + 2 * 16) / 2.0
    assert abs(r["frac_in_stream_int32_cost"] - r["frac"] * in_stream / cyc) < 1e-12
    # round 6: the implementation-independent readings beside it -- fp64-class instructions x 2 flop x 64 lanes against the 78.6
    # TFLOP/s fp64 vector peak; without the in-kernel clock stamps the measured cycles per wave-step are not invented
    wave_steps = (2 ** 20 / 64) * 1024
    assert r["fp64_insts_per_wave_step"] == 35.5
    assert abs(r["fp64_fma_frac"] - 35.5 * 2 * 64 * wave_steps / 2.0e-3 / 78.6e12) < 1e-12
    assert r["cycles_per_wave_step_measured"] is None and r["lds_busy_frac"] is None
    stamps = [0, 0, 2_200_000, 100_000, 0, 0, 2_200_000, 100_000]     # 2.2e6 shader ticks in 1e5 ticks of the 100 MHz clock: 2200 MHz
    r2 = bench.kernel_rooflines("logsv_rng_kernel", 2.0, 50, 1 << 20, c2, isa, pmc, 2400.0, stamps)["roofline"]
    assert abs(r2["clock_mhz_in_kernel"] - 2200.0) < 1e-9
    assert abs(r2["cycles_per_wave_step_measured"] - 2.0e-3 * 2.2e9 * 1024 / wave_steps) < 1e-9
    assert r["clock_mhz_sensor"] == 2400.0 and r["traffic"] == 59.0e6 and r["stale"] is False
    assert r["insts_per_wave_step_counters"] == 53.2
    assert r["clock_mhz_in_kernel"] is None and r["frac_at_sustained_clock"] is None        # no stamps handed in
    # the in-kernel clock: first block 2.0e6 shader cycles in 1e5 ticks of the 100 MHz clock = 2000 MHz, last block 2200 MHz
    stamps = [10, 1000, 10 + 2_000_000, 1000 + 100_000, 50, 7000, 50 + 1_100_000, 7000 + 50_000]
    rc = bench.kernel_rooflines("logsv_rng_kernel", 2.0, 50, 1 << 20, c2, isa, pmc, None, stamps)["roofline"]
    assert abs(rc["clock_probe"]["first_block"]["mhz"] - 2000.0) < 1e-9 and abs(rc["clock_probe"]["last_block"]["mhz"] - 2200.0) < 1e-9
    assert abs(rc["clock_mhz_in_kernel"] - 2100.0) < 1e-9 and abs(rc["clock_probe"]["first_block"]["wave_lifetime_ms"] - 1.0) < 1e-12
    assert abs(rc["frac_at_sustained_clock"] - rc["frac"] * 2400.0 / 2100.0) < 1e-12
    assert abs(rc["frac_in_stream_at_sustained_clock"] - rc["frac_in_stream_int32_cost"] * 2400.0 / 2100.0) < 1e-12
    assert bench.clock_from_stamps([0] * 8) == {}                                               # a launch that never stamped
    # the N > 1 self-check's deviation measure: relative, zero references only match zeros, NaN patterns must agree
    assert bench.max_rel_dev([np.array([1.0, 2.0])], [np.array([1.0, 2.0 * (1 + 1e-13)])]) < 1.1e-13
    assert bench.max_rel_dev([np.array([0.0, 2.0])], [np.array([0.0, 2.0])]) == 0.0
    assert bench.max_rel_dev([np.array([1e-30, 2.0])], [np.array([0.0, 2.0])]) == float("inf")
    assert bench.max_rel_dev([np.array([np.nan, 2.0])], [np.array([1.0, 2.0])]) == float("inf")
    assert bench.max_rel_dev([np.array([1.0])], [np.array([1.0, 2.0])]) == float("inf")
    rr = bench.kernel_rooflines("logsv_rng_kernel", 2.0, 50, 1 << 20, c2, isa, pmc, 2400.0)
    assert rr["roofline_hbm"]["algorithmic_bytes"] == 32.0 * 2 ** 20
    # a library that is not the one the histogram describes: stale; counters of another build are not quoted
    stale = bench.kernel_rooflines("logsv_rng_kernel", 2.0, 50, 1 << 20, c2, dict(isa, stale=True),
                                   dict(pmc, matches_loaded_library=False), None)["roofline"]
    assert stale["stale"] is True and stale["insts_per_wave_step_counters"] is None
    # without a histogram for the kernel the line falls back to the HBM roof (marked stale) instead of inventing a count
    none = bench.kernel_rooflines("logsv_rng_kernel", 2.0, 50, 1 << 20, c2, {"kernels": {}, "stale": True, "source": None,
                                                                             "lib_sha256": ""}, {}, None)["roofline"]
    assert none["bound"] == "hbm" and none["stale"] is True
    # the histogram the build wrote beside the library names the library it was read from
    from stochvolmodels_amd import _lib, build
    build.build()
    live = bench.load_isa(_lib.LIB_PATH)
    assert live["stale"] is False and live["kernels"]["logsv_rng_kernel"]["valu"] > 60
    assert live["kernels"]["logsv_chain_rng_kernel"]["classes"]["quarter"] == 2
    # ... and the register / scratch / LDS footprint of every kernel of the build: the stepping kernels must not spill
    import json
    meta = json.load(open(os.path.join(os.path.dirname(_lib.LIB_PATH), "libsvmc.isa.json")))["metadata"]
    stepping = [k for k in meta if "rng_kernel" in k]
    assert len(stepping) >= 6 and len([k for k in meta if "payoff_group_kernel" in k]) >= 60
    spilling = {k: v for k, v in meta.items() if v["scratch_bytes"] != 0}
    assert not spilling, spilling                              # no kernel of the library touches scratch memory


def test_batched_gradient_uses_scipys_difference_points():
    """ImpliedVolObjective.gradient (the analytic calibration's batched forward differences) must evaluate the objective
    exactly where SLSQP's own differencing would (scipy.optimize approx_derivative, '2-point', abs_step = sqrt(eps),
    steps flipped or shortened at the box): same points -> same gradient -> same optimizer path.  Checked against scipy
    itself on random points, incl. points on and next to the bounds."""
    from scipy.optimize._numdiff import approx_derivative
    from stochvolmodels_amd.utils.calibration import ImpliedVolObjective
    rng = np.random.default_rng(5)
    n = 5
    for trial in range(200):
        lb = rng.uniform(-2.0, 0.0, n)
        ub = lb + rng.uniform(1e-9 if trial % 7 == 0 else 0.1, 3.0, n)
        x0 = rng.uniform(lb, ub)
        if trial % 3 == 0:
            x0[rng.integers(n)] = ub[rng.integers(n)] if trial % 2 else lb[rng.integers(n)]
            x0 = np.clip(x0, lb, ub)
        if trial % 5 == 0:
            j = rng.integers(n)
            x0[j] = ub[j] - rng.uniform(0, 2e-8)                                   # closer to the bound than one step
            x0 = np.clip(x0, lb, ub)
        seen = []

        def fun(x):
            seen.append(np.array(x, dtype=float))
            return float(np.sum(np.sin(x) * np.arange(1, n + 1)))

        f0 = fun(x0)
        seen.clear()
        g_scipy = approx_derivative(fun, x0, method="2-point", abs_step=float(np.sqrt(np.finfo(float).eps)), f0=f0,
                                    bounds=(lb, ub))
        pts_scipy = [p.copy() for p in seen]
        obj = ImpliedVolObjective(None, np.zeros(1), np.ones(1), model_vols_batch=lambda pts: [np.array([0.0])] * len(pts),
                                  bounds=list(zip(lb, ub)))
        h = obj.fd_steps(x0)
        for i in range(n):
            xi = x0.copy()
            xi[i] += h[i]
            np.testing.assert_array_equal(xi, pts_scipy[i], err_msg=f"trial {trial} component {i}")
        # and the gradient assembled the same way from the same values
        obj._value = lambda vols: vols                                              # the batch hands the values straight through
        obj.model_vols_batch = lambda pts: [fun(p) for p in pts]
        obj._last = (x0.tobytes(), f0)
        np.testing.assert_array_equal(obj.gradient(x0), g_scipy)


def test_bench_watchdog_ends_a_stuck_phase():
    """bench.py's Watchdog (no GPU involved): a phase that overruns its deadline ends the process with status 1 and the
    phase's name and a traceback on stderr -- even while the main thread sits in a call that never returns; a phase that
    finishes in time and disarms leaves the process alone"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import importlib.util, sys, time\n"
            f"spec = importlib.util.spec_from_file_location('bench', r'{os.path.join(root, 'bench.py')}')\n"
            "bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)\n"
            "wd = bench.Watchdog(7)\n"
            "wd.arm(30.0, 'a phase that ends in time'); wd.disarm()\n"
            "wd.arm(2.0, 'a collective that never returns')\n"
            "if sys.argv[1] == 'hang': time.sleep(60)\n"
            "wd.disarm(); print('finished')\n")
    ok = subprocess.run([sys.executable, "-c", code, "fine"], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0 and "finished" in ok.stdout
    stuck = subprocess.run([sys.executable, "-c", code, "hang"], capture_output=True, text=True, timeout=120)
    assert stuck.returncode == 1 and "finished" not in stuck.stdout
    assert "rank 7" in stuck.stderr and "a collective that never returns" in stuck.stderr and "did not finish within 2 s" in stuck.stderr
    assert "time.sleep" in stuck.stderr or "Timeout" in stuck.stderr or "line" in stuck.stderr          # the traceback


def test_dlpack_capsule_keeps_and_releases_its_owner():
    """engine.DeviceArray.__dlpack__ (no GPU needed for the plumbing): the capsule is a "dltensor" of a kDLROCM float64 2-d tensor
    that keeps the array alive; a capsule nobody consumed releases it when it dies, a consumed one when the consumer calls the
    tensor's deleter"""
    import ctypes as C
    import gc

    from stochvolmodels_amd import engine

    class FakeArray:
        shape, device, ptr = (3, 4), 2, 0x7000

    before = len(engine._DLPACK_ALIVE)
    cap = engine._dlpack_capsule(FakeArray())
    assert len(engine._DLPACK_ALIVE) == before + 1
    C.pythonapi.PyCapsule_GetName.restype = C.c_char_p
    C.pythonapi.PyCapsule_GetName.argtypes = [C.py_object]
    assert C.pythonapi.PyCapsule_GetName(cap) == b"dltensor"
    m = next(v[0] for v in engine._DLPACK_ALIVE.values() if v[3].__class__ is FakeArray)
    t = m.dl_tensor
    assert (t.data, t.device.device_type, t.device.device_id, t.ndim, t.dtype.code, t.dtype.bits, t.dtype.lanes) == \
        (0x7000, 10, 2, 2, 2, 64, 1) and [t.shape[0], t.shape[1]] == [3, 4] and not t.strides
    del cap, m, t
    gc.collect()
    assert len(engine._DLPACK_ALIVE) == before                       # unconsumed: released with the capsule
    cap = engine._dlpack_capsule(FakeArray())
    m = next(v[0] for v in engine._DLPACK_ALIVE.values() if v[3].__class__ is FakeArray)
    m.deleter(C.pointer(m))                                          # what a consumer does when it is done with the tensor
    assert len(engine._DLPACK_ALIVE) == before


def test_chain_marshalling_is_keyed_by_content():
    """the objective of a calibration prices one chain hundreds of times: its arrays are marshalled for the C ABI once per
    CONTENT (engine.DeviceRandoms._marshalled_chain) and the option-type codes are kept per array content
    (engine.option_type_codes) -- a caller that changes an array in place gets a new entry, never stale pointers"""
    from stochvolmodels_amd.engine import DeviceRandoms, option_type_codes
    ty = np.array(["C", "P", "IC", "IP"])
    c1, c2 = option_type_codes(ty), option_type_codes(ty.copy())
    assert c1 is c2 and c1.tolist() == [0, 1, 2, 3] and not c1.flags.writeable
    ty[1] = "C"
    assert option_type_codes(ty).tolist() == [0, 0, 2, 3]
    assert option_type_codes(np.array([["C", "P"], ["IP", "IC"]])).tolist() == [[0, 1], [3, 2]]
    for _ in range(2):                                                   # a bad code raises every time, cached or not
        with pytest.raises(ValueError):
            option_type_codes(np.array(["C", "X"]))
    assert option_type_codes(["P", "C"]).tolist() == [1, 0]              # plain sequences still work

    res = DeviceRandoms.frozen([10, 20], [0.01, 0.01], nb_path=100, n_local=100, col0=0, seed=1)
    ttms, fw, df = np.array([0.1, 0.3]), np.array([1.0, 1.01]), np.array([0.99, 0.98])
    ks = [np.array([0.9, 1.0, 1.1]), np.array([0.8, 1.2])]
    cs = [option_type_codes(np.array(["P", "C", "C"])), option_type_codes(np.array(["P", "C"]))]
    a = res._marshalled_chain(ttms, fw, df, ks, cs)
    assert res._marshalled_chain(ttms.copy(), fw, df, [k.copy() for k in ks], cs) is a
    assert a["total"] == 5 and a["offs"].tolist() == [0, 3, 5] and a["slices"] == [slice(0, 3), slice(3, 5)]
    assert a["keep"][3].tolist() == [0.9, 1.0, 1.1, 0.8, 1.2] and a["keep"][4].tolist() == [1, 0, 0, 1, 0]
    ks[0][1] = 1.05                                                      # in place: another chain, the kept copy untouched
    b = res._marshalled_chain(ttms, fw, df, ks, cs)
    assert b is not a and b["keep"][3][1] == 1.05 and a["keep"][3][1] == 1.0
