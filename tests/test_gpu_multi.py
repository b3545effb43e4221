"""
Single-process multi-device sessions (include/svmc.h svmc_multi_*; stochvolmodels_amd/multi.py) on the hardware there is:
R shards of one job on ONE device through the host transport (the same code path as R devices: one session, one stream and
one host thread per shard, the two all-reduces of a chain through pinned host memory), against the one-session job.

The randoms are indexed by the global path id, so the terminal STATE of the sharded job is the one-session state bit for bit
(paths do not interact); prices differ by the order of the final additions only (1e-12).  SURVEY.md 8(e).
"""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sv():
    import stochvolmodels_amd as sv
    from stochvolmodels_amd import _lib
    _lib.load()
    return sv


def _small_chain():
    ttms, fw, df = np.array([0.1, 0.25, 0.5]), np.array([1.0, 1.01, 1.03]), np.array([0.99, 0.98, 0.96])
    kk = np.linspace(0.7, 1.3, 7)
    strikes = tuple(f * kk for f in fw)
    types = (np.where(kk >= 1.0, "C", "P"), np.array(["IP", "IC", "C", "P", "IC", "IP", "C"]), np.where(kk >= 1.0, "C", "P"))
    return ttms, fw, df, strikes, types


def _rel(a, b):
    a, b = np.concatenate([np.ravel(v) for v in a]), np.concatenate([np.ravel(v) for v in b])
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-300)))


@pytest.mark.parametrize("shards", [1, 2, 3, 8])
@pytest.mark.parametrize("vt", [1, 2])
def test_multi_session_equals_one_session(sv, shards, vt):
    from stochvolmodels_amd.engine import get_engine
    from stochvolmodels_amd.multi import MultiDeviceSession
    p = sv.LOGSV_BTC_PARAMS
    ttms, fw, df, strikes, types = _small_chain()
    if vt == 2:
        strikes = tuple(np.linspace(0.3, 1.5, 7) for _ in fw)
        types = tuple(np.where(k >= 0.8, "C", "P") for k in strikes)
    n, seed, spy = 100_003, 4242, 200                          # a path count no shard count divides
    variable_type = sv.VariableType.LOG_RETURN if vt == 1 else sv.VariableType.Q_VAR
    args = dict(ttms=ttms, forwards=fw, discfactors=df, strikes_ttms=strikes, optiontypes_ttms=types, v0=p.sigma0,
                theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta, volvol=p.volvol,
                vol_backbone_etas=np.array([1.0, 0.95, 1.05]), nb_path=n, nb_steps_per_year=spy, variable_type=variable_type)
    pr1, sd1 = sv.logsv_mc_chain_pricer(seed=seed, **args)
    state1 = get_engine(n).get_state()
    ms = MultiDeviceSession([0] * shards, n, 3, 21, reduce="host")
    try:
        info = ms.info()
        assert info["n_shards"] == shards and info["reduce"] == "host"
        assert [s["path_offset"] for s in info["shards"]] == [n * r // shards for r in range(shards)]
        assert sum(s["n_path"] for s in info["shards"]) == n
        prm, sdm = ms.price_logsv_chain(ttms, fw, df, strikes, types, p.sigma0, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol,
                                        args["vol_backbone_etas"], True, spy, vt, seed, 0)
        assert ms.info()["shards_agree"] is True
        for a, b in zip(ms.get_state(), state1):
            np.testing.assert_array_equal(a, b)                # the same paths, whoever stepped them
        if shards == 1:
            for a, b in zip(prm + sdm, pr1 + sd1):
                np.testing.assert_array_equal(a, b)
        assert _rel(prm, pr1) <= 1e-12 and _rel(sdm, sd1) <= 1e-12, (_rel(prm, pr1), _rel(sdm, sd1))
        # Heston QE through the same multi-session
        h = dict(v0=0.04, theta=0.04, kappa=4.0, rho=-0.5, volvol=0.4)
        hp1, hs1 = sv.heston_mc_chain_pricer(ttms=ttms, forwards=fw, discfactors=df, strikes_ttms=strikes, optiontypes_ttms=types,
                                             nb_path=n, scheme="qe", seed=seed, variable_type=variable_type, **h)
        hpm, hsm = ms.price_heston_chain(ttms, fw, df, strikes, types, h["v0"], h["theta"], h["kappa"], h["rho"], h["volvol"], 1,
                                         360, vt, seed, 0)
        assert ms.info()["shards_agree"] is True
        assert _rel(hpm, hp1) <= 1e-12 and _rel(hsm, hs1) <= 1e-12
    finally:
        ms.close()


def test_devices_keyword_of_the_pricers(sv):
    """logsv_mc_chain_pricer(devices=...) / heston / model_mc_price_chain(devices=...): the drop-in user's route to several
    GPUs from one interpreter; seeds behave as on one device (same seed -> same prices, set_seed + call counter)"""
    p = sv.LOGSV_BTC_PARAMS
    ttms, fw, df, strikes, types = _small_chain()
    kw = dict(ttms=ttms, forwards=fw, discfactors=df, strikes_ttms=strikes, optiontypes_ttms=types, v0=p.sigma0, theta=p.theta,
              kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta, volvol=p.volvol, vol_backbone_etas=np.ones(3), nb_path=1 << 16,
              nb_steps_per_year=120)
    one, one_sd = sv.logsv_mc_chain_pricer(seed=11, **kw)
    two, two_sd = sv.logsv_mc_chain_pricer(seed=11, devices=[0, 0], reduce="host", **kw)
    again, _ = sv.logsv_mc_chain_pricer(seed=11, devices=[0, 0], reduce="host", **kw)
    assert _rel(two, one) <= 1e-12 and _rel(two_sd, one_sd) <= 1e-12
    for a, b in zip(two, again):
        np.testing.assert_array_equal(a, b)
    chain = sv.OptionChain(ttms=ttms, forwards=fw, strikes_ttms=strikes, optiontypes_ttms=types, ids=None, discfactors=df)
    a, _ = sv.LogSVPricer().model_mc_price_chain(chain, p, nb_path=1 << 16, nb_steps=120, seed=5)
    b, _ = sv.LogSVPricer().model_mc_price_chain(chain, p, nb_path=1 << 16, nb_steps=120, seed=5, devices=[0, 0, 0], reduce="host")
    assert _rel(b, a) <= 1e-12
    hp = sv.HestonParams()
    a, _ = sv.HestonPricer().model_mc_price_chain(chain, hp, nb_path=1 << 16, seed=5)
    b, _ = sv.HestonPricer().model_mc_price_chain(chain, hp, nb_path=1 << 16, seed=5, devices=2, reduce="host") \
        if _device_count() >= 2 else sv.HestonPricer().model_mc_price_chain(chain, hp, nb_path=1 << 16, seed=5, devices=[0, 0],
                                                                            reduce="host")
    assert _rel(b, a) <= 1e-12
    with pytest.raises(ValueError):
        sv.logsv_mc_chain_pricer(seed=1, devices=[0], comm=object(), **kw)
    from stochvolmodels_amd import multi
    multi.close_all()


def _device_count():
    import ctypes as C
    from stochvolmodels_amd import _lib
    n = C.c_int()
    _lib.check(_lib.load().svmc_device_count(C.byref(n)))
    return n.value


def test_multi_session_errors_and_rccl_refusal(sv):
    """argument errors come back as the reference's exception types; RCCL on shards that share a device is refused with the
    reason (RCCL takes one rank per device), AUTO falls back to the host transport; a shard that fails mid-call releases the
    others (the call returns, it does not hang)"""
    from stochvolmodels_amd._lib import SvmcError
    from stochvolmodels_amd.multi import MultiDeviceSession
    with pytest.raises(ValueError):
        MultiDeviceSession([0, 0], 1, 1, 1)                    # fewer paths than shards
    with pytest.raises(ValueError):
        MultiDeviceSession([99], 1024, 1, 1)                   # no such device
    with pytest.raises(SvmcError) as exc:
        MultiDeviceSession([0, 0], 1024, 1, 1, reduce="rccl")
    assert "share a device" in str(exc.value)
    ms = MultiDeviceSession([0, 0], 4096, 2, 16, reduce="auto")
    try:
        assert ms.info()["reduce"] == "host"
        ttms, fw, df, strikes, types = _small_chain()
        with pytest.raises(ValueError):                        # 3 expiries into a 2-expiry multi-session
            ms.price_logsv_chain(ttms, fw, df, strikes, types, 0.8, 1.0, 3.0, 3.0, 0.1, 1.8, np.ones(3), True, 120, 1, 1, 0)
        bad = (np.array(["C", "P", "XX", "C", "P", "C", "P"]),)
        with pytest.raises(ValueError, match="unknown option payoff code"):
            ms.price_logsv_chain(ttms[:1], fw[:1], df[:1], strikes[:1], bad, 0.8, 1.0, 3.0, 3.0, 0.1, 1.8, np.ones(1), True, 120, 1, 1, 0)
        with pytest.raises(ValueError):                        # ttms not increasing: every shard fails the same check
            ms.price_logsv_chain(ttms[[1, 0]], fw[:2], df[:2], strikes[:2], types[:2], 0.8, 1.0, 3.0, 3.0, 0.1, 1.8, np.ones(2),
                                 True, 120, 1, 1, 0)
        # ... and the multi-session is usable afterwards
        pr, sd = ms.price_logsv_chain(ttms[:2], fw[:2], df[:2], strikes[:2], types[:2], 0.8, 1.0, 3.0, 3.0, 0.1, 1.8, np.ones(2),
                                      True, 120, 1, 1, 0)
        assert all(np.all(np.isfinite(a)) for a in pr)
    finally:
        ms.close()


def test_a_shard_failing_between_the_meeting_points_releases_the_others(sv, monkeypatch):
    """a shard whose all-reduce fails AFTER the first meeting point of a chain (fault injection: no argument check can produce
    that) -- the others are released from the second one instead of waiting for ever, the call raises naming the shard, and
    the NEXT call on the same multi-session is right: every job restarts the exchange slots at parity 0 on every shard (a
    failed job leaves the shards' counters apart; shards on different parities would read each other's stale slots)"""
    from stochvolmodels_amd._lib import SvmcError
    from stochvolmodels_amd.multi import MultiDeviceSession
    p = sv.LOGSV_BTC_PARAMS
    ttms, fw, df, strikes, types = _small_chain()
    args = (ttms, fw, df, strikes, types, p.sigma0, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, np.ones(3), True, 200, 1, 99, 0)
    ref, ref_sd = sv.logsv_mc_chain_pricer(ttms=ttms, forwards=fw, discfactors=df, strikes_ttms=strikes, optiontypes_ttms=types,
                                           v0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta, volvol=p.volvol,
                                           vol_backbone_etas=np.ones(3), nb_path=50_000, nb_steps_per_year=200, seed=99)
    for shard, k in ((1, 1), (2, 0), (0, 3)):              # second all-reduce of the first chain; first of the first; of the second chain
        monkeypatch.setenv("SVMC_MULTI_FAULT", f"{shard},{k}")
        ms = MultiDeviceSession([0, 0, 0], 50_000, 3, 21, reduce="host")
        monkeypatch.delenv("SVMC_MULTI_FAULT")
        try:
            seen_failure = False
            for call in range(4):
                try:
                    pr, sd = ms.price_logsv_chain(*args)
                except SvmcError as exc:
                    assert not seen_failure and f"shard {shard} " in str(exc) and "fault injection" in str(exc), str(exc)
                    seen_failure = True
                    continue
                assert _rel(pr, ref) <= 1e-12 and _rel(sd, ref_sd) <= 1e-12 and ms.info()["shards_agree"] is True, (shard, k, call)
            assert seen_failure
        finally:
            ms.close()


def test_c4_eight_shards_of_full_size_on_one_device(sv):
    """C4 as the driver's 8-GPU run shards it -- 2^24 paths, 8 expiries x 128 steps, 8 x 21 strikes, EIGHT shards of 2^21 --
    through the multi-session on the one device there is (the shards run back to back on it), against the one-session job
    of 2^24 paths: prices and standard errors to 1e-12, the shards' returned bits identical, every path's terminal state
    identical to the one-session job's"""
    from stochvolmodels_amd.engine import get_engine
    p = sv.LOGSV_BTC_PARAMS
    n, seed, spy = 1 << 24, 20240604, 1016
    ttms = np.arange(1, 9) / 8.0
    fw, df = 67000.0 * np.exp(0.05 * ttms), np.exp(-0.05 * ttms)
    strikes = tuple(f * np.linspace(0.6, 1.6, 21) for f in fw)
    types = tuple(np.where(k >= f, "C", "P") for k, f in zip(strikes, fw))
    kw = dict(ttms=ttms, forwards=fw, discfactors=df, strikes_ttms=strikes, optiontypes_ttms=types, v0=p.sigma0, theta=p.theta,
              kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta, volvol=p.volvol, vol_backbone_etas=np.ones(8), nb_path=n,
              nb_steps_per_year=spy, seed=seed)
    pr1, sd1 = sv.logsv_mc_chain_pricer(**kw)
    x1 = get_engine(n).get_state()[0]
    from stochvolmodels_amd.multi import get_multi_session, close_all
    pr8, sd8 = sv.logsv_mc_chain_pricer(devices=[0] * 8, reduce="host", **kw)
    ms = get_multi_session([0] * 8, n, 8, 168, reduce="host")
    info = ms.info()
    assert info["shards_agree"] is True and [s["n_path"] for s in info["shards"]] == [1 << 21] * 8
    assert [s["path_offset"] for s in info["shards"]] == [r << 21 for r in range(8)]
    dev_p, dev_s = _rel(pr8, pr1), _rel(sd8, sd1)
    print(f"C4 8 x 2^21 shards on one device vs one session of 2^24: prices {dev_p:.2e}, stderrs {dev_s:.2e}; "
          f"shard ms {[round(s['last_call_ms'], 1) for s in info['shards']]}")
    assert dev_p <= 1e-12 and dev_s <= 1e-12
    np.testing.assert_array_equal(ms.get_state()[0], x1)
    close_all()


def test_c_host_multi_example(sv, tmp_path):
    """examples/price_chain_multi.c: a plain-C host, several shards from one process, no launcher"""
    exe = str(tmp_path / "price_chain_multi")
    libdir = os.path.join(ROOT, "stochvolmodels_amd")
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "price_chain_multi.c"), "-o", exe, "-L" + libdir, "-lsvmc",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-lm"], check=True)
    n_dev = _device_count()
    out1 = json.loads(subprocess.run([exe, "1", "65536", "123", "host"], check=True, capture_output=True, text=True, timeout=300).stdout)
    # three shards: on distinct devices where there are three, else all on device 0 (then AUTO must pick the host transport)
    out3 = json.loads(subprocess.run([exe, "3", "65536", "123"], check=True, capture_output=True, text=True, timeout=300).stdout)
    assert out3["shards_agree"] is True and out3["n_shards"] == 3
    assert out3["reduce"] == ("rccl" if n_dev >= 3 else "host")
    if out3["reduce"] == "rccl":
        assert out3["rccl_ranks_seen"] == 3
    P_ = sv.LOGSV_BTC_PARAMS
    ttms, fw, df = np.array([0.1, 0.25]), np.array([1.0, 1.01]), np.array([0.99, 0.98])
    kk = np.array([0.8, 1.0, 1.2])
    strikes = (kk, 1.01 * kk)
    types = (np.array(["P", "C", "C"]), np.array(["IP", "IC", "C"]))
    pr, sd = sv.logsv_mc_chain_pricer(ttms=ttms, forwards=fw, discfactors=df, strikes_ttms=strikes, optiontypes_ttms=types,
                                      v0=P_.sigma0, theta=P_.theta, kappa1=P_.kappa1, kappa2=P_.kappa2, beta=P_.beta,
                                      volvol=P_.volvol, vol_backbone_etas=np.ones(2), nb_path=65536, nb_steps_per_year=120, seed=123)
    np.testing.assert_array_equal(np.concatenate(pr), out1["logsv_prices"])       # one shard = the single-session driver
    np.testing.assert_array_equal(np.concatenate(sd), out1["logsv_stderrs"])
    np.testing.assert_allclose(out3["logsv_prices"], out1["logsv_prices"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(out3["logsv_stderrs"], out1["logsv_stderrs"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(out3["heston_qe_prices"], out1["heston_qe_prices"], rtol=1e-12, atol=0)


@pytest.mark.skipif("_device_count() < 2")
def test_multi_session_over_rccl_on_distinct_devices(sv):
    """with two or more GPUs in the box: the RCCL transport (ncclCommInitAll) against the host transport, bit for bit up to
    the collective's own order of additions"""
    from stochvolmodels_amd.multi import MultiDeviceSession
    n_dev = _device_count()
    p = sv.LOGSV_BTC_PARAMS
    ttms, fw, df, strikes, types = _small_chain()
    out = {}
    for mode in ("rccl", "host"):
        ms = MultiDeviceSession(n_dev, 1 << 18, 3, 21, reduce=mode)
        try:
            out[mode] = ms.price_logsv_chain(ttms, fw, df, strikes, types, p.sigma0, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol,
                                             np.ones(3), True, 200, 1, 77, 0)
            info = ms.info()
            assert info["shards_agree"] is True and info["reduce"] == mode
            if mode == "rccl":
                assert info["rccl_ranks_seen"] == n_dev
        finally:
            ms.close()
    assert _rel(out["rccl"][0], out["host"][0]) <= 1e-12 and _rel(out["rccl"][1], out["host"][1]) <= 1e-12
