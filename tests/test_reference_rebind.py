"""
INTEGRATION.md, route B, exercised for real: the reference package (imported from /root/reference in the build
container, NumPy mode through the identity-njit stand-ins) with its module-level Monte Carlo entry points rebound to
this package's, driven through the REFERENCE's own LogSVPricer / HestonPricer objects, OptionChain, parameter
dataclasses and enums.  The engine is the CPU test double (no GPU here), so what this pins is the host seam: foreign
types in, reference-shaped results out.  Skipped where the reference checkout is absent (the GPU box).
"""
import os
import sys

import numpy as np
import pytest

REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


@pytest.fixture()
def rebound(monkeypatch, oracle):
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "golden", "_shims"), REF, here):
        if p not in sys.path:
            monkeypatch.syspath_prepend(p)
    import stochvolmodels as ref
    import stochvolmodels.pricers.heston_pricer as ref_heston
    import stochvolmodels.pricers.logsv_pricer as ref_logsv
    import stochvolmodels.utils.mc_payoffs as ref_payoffs
    import stochvolmodels_amd as sv
    import stochvolmodels_amd.pricers.heston_pricer as amd_heston
    import stochvolmodels_amd.pricers.logsv_pricer as amd_logsv
    from fake_engine import FakeEngine
    engines = {}

    def fake_get_engine(n_path, path_offset=0, device=None):
        return engines.setdefault((n_path, path_offset), FakeEngine(n_path, path_offset))

    monkeypatch.setattr(amd_logsv, "get_engine", fake_get_engine)
    monkeypatch.setattr(amd_heston, "get_engine", fake_get_engine)
    for name in ("logsv_mc_chain_pricer", "logsv_mc_chain_pricer_fixed_randoms", "simulate_logsv_x_vol_terminal"):
        monkeypatch.setattr(ref_logsv, name, getattr(amd_logsv, name))
    monkeypatch.setattr(ref_heston, "heston_mc_chain_pricer", amd_heston.heston_mc_chain_pricer)
    return ref, sv, ref_logsv, ref_heston, ref_payoffs


def _chains(ref, sv):
    ttms, fw, df = np.array([0.08, 0.25]), np.array([1.0, 1.02]), np.array([0.995, 0.98])
    kk = (np.array([0.9, 1.0, 1.1]), np.array([0.85, 1.02, 1.2]))
    ty = (np.array(["P", "C", "C"]), np.array(["IP", "IC", "C"]))
    mk = lambda mod: mod.OptionChain(ttms=ttms, forwards=fw, discfactors=df, strikes_ttms=kk, optiontypes_ttms=ty,   # noqa: E731
                                     ids=np.array(["a", "b"]))
    return mk(ref), mk(sv)


def test_reference_pricer_objects_drive_the_rebound_path(rebound):
    ref, sv, ref_logsv, ref_heston, _ = rebound
    chain_ref, chain_amd = _chains(ref, sv)
    # LogSV: the reference's pricer, params, chain and enum; the nb_steps default rule included
    for vt_name in ("LOG_RETURN", "Q_VAR"):
        kw = dict(nb_path=777, nb_steps=None if vt_name == "LOG_RETURN" else 150)
        chains = (chain_ref, chain_amd)
        if vt_name == "Q_VAR":
            q = dict(strikes_ttms=(np.array([0.1, 0.4]),) * 2, optiontypes_ttms=(np.array(["C", "P"]),) * 2)
            chains = tuple(m.OptionChain(ttms=c.ttms, forwards=c.forwards, discfactors=c.discfactors, ids=c.ids, **q)
                           for m, c in ((ref, chain_ref), (sv, chain_amd)))
        sv.set_seed(11)
        a, ea = ref.LogSVPricer().model_mc_price_chain(option_chain=chains[0], params=ref_logsv.LOGSV_BTC_PARAMS,
                                                       variable_type=getattr(ref.VariableType, vt_name), **kw)
        sv.set_seed(11)
        b, eb = sv.LogSVPricer().model_mc_price_chain(option_chain=chains[1], params=sv.LOGSV_BTC_PARAMS,
                                                      variable_type=getattr(sv.VariableType, vt_name), **kw)
        assert isinstance(a, list) and len(a) == 2 and a[0].dtype == np.float64
        for x, y in zip(list(a) + list(ea), list(b) + list(eb)):
            np.testing.assert_array_equal(x, y)
    # Heston through the reference's HestonPricer
    sv.set_seed(3)
    a, _ = ref.HestonPricer().model_mc_price_chain(option_chain=chain_ref, params=ref_heston.HestonParams(), nb_path=555)
    sv.set_seed(3)
    b, _ = sv.HestonPricer().model_mc_price_chain(option_chain=chain_amd, params=sv.HestonParams(), nb_path=555)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    # the reference's error contract survives the rebinding
    with pytest.raises(NotImplementedError):
        ref.LogSVPricer().model_mc_price_chain(option_chain=chain_ref, params=ref_logsv.LOGSV_BTC_PARAMS, nb_path=64,
                                               variable_type=ref.VariableType.SIGMA)


def test_reference_fixed_randoms_objective_path(rebound, oracle):
    """the call the reference's MC calibration objective makes (logsv_pricer.py:244-265): its own
    get_randoms_for_chain_valuation output handed to the rebound fixed-randoms pricer, against the reference's own
    pricer on the same randoms"""
    ref, sv, ref_logsv, _, _ = rebound
    import importlib
    import stochvolmodels.pricers.logsv_pricer as fresh
    chain_ref, _ = _chains(ref, sv)
    p = ref_logsv.LOGSV_BTC_PARAMS
    W0s, W1s, dts = fresh.get_randoms_for_chain_valuation(ttms=chain_ref.ttms, nb_path=400, nb_steps_per_year=100, seed=9)
    kw = dict(ttms=chain_ref.ttms, forwards=chain_ref.forwards, discfactors=chain_ref.discfactors,
              strikes_ttms=chain_ref.strikes_ttms, optiontypes_ttms=chain_ref.optiontypes_ttms, W0s=W0s, W1s=W1s, dts=dts,
              v0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta, volvol=p.volvol,
              vol_backbone_etas=p.get_vol_backbone_etas(ttms=chain_ref.ttms))
    got, got_e = ref_logsv.logsv_mc_chain_pricer_fixed_randoms(**kw)            # rebound -> this package
    original = importlib.reload(importlib.import_module("stochvolmodels.pricers.logsv_pricer"))
    want, want_e = original.logsv_mc_chain_pricer_fixed_randoms(**kw)           # the reference's own NumPy-mode code
    for x, y in zip(list(got) + list(got_e), list(want) + list(want_e)):
        np.testing.assert_allclose(x, y, rtol=1e-12, atol=1e-15)
