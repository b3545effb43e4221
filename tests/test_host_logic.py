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
    with pytest.raises(NotImplementedError):
        infer_black_ivols(pr[:1], 0.5, 1.0, k[:1], np.array(["IC"]))
