"""shared inputs of the distributed tests"""
import numpy as np

_KK = np.array([0.8, 1.0, 1.2])
LOGSV_CASE = dict(
    ttms=np.array([0.05, 0.1, 0.25]), forwards=np.array([1.0, 1.01, 1.02]), discfactors=np.array([0.999, 0.99, 0.98]),
    strikes_ttms=tuple(f * _KK for f in (1.0, 1.01, 1.02)),
    optiontypes_ttms=(np.array(["P", "C", "C"]), np.array(["IP", "IC", "C"]), np.array(["P", "C", "IC"])),
    v0=0.8376, theta=1.0413, kappa1=3.1844, kappa2=3.058, beta=0.1514, volvol=1.8458,
    vol_backbone_etas=np.array([1.0, 0.95, 1.05]), is_spot_measure=True, nb_path=1001, nb_steps_per_year=120, seed=77)
HESTON_CASE = dict(
    ttms=np.array([0.05, 0.1]), forwards=np.array([1.0, 1.01]), discfactors=np.array([0.999, 0.99]),
    strikes_ttms=(_KK, 1.01 * _KK), optiontypes_ttms=(np.array(["P", "C", "C"]), np.array(["IP", "IC", "C"])),
    v0=0.04, theta=0.04, kappa=4.0, rho=-0.5, volvol=0.4, nb_path=1001, seed=78, scheme="qe", nb_steps_per_year=100)
ROUGH_CASE = dict(
    ttms=np.array([0.05, 0.1]), forwards=np.array([1.0, 1.01]), discfactors=np.array([0.999, 0.99]),
    strikes_ttms=(_KK, 1.01 * _KK), optiontypes_ttms=(np.array(["P", "C", "C"]), np.array(["IP", "IC", "C"])),
    sigma0=0.377, theta=0.347, kappa1=1.29, kappa2=1.93, beta=2.45, orthog_vol=1.81,
    weights=np.array([0.777, 1.554, 8.516]), nodes=np.array([0.0772, 5.19, 108.46]))


def bounded_put_check(x, strikes, types, ref_prices, forward=1.0):
    """Monte Carlo puts WITHOUT the reference's forward recentring against reference prices (calls mapped through
    put-call parity, discount factor 1): (|mc - ref| / stderr per strike, stderr).  A put payoff is bounded, so its
    standard error is honest even where E[S^2] is infinite (Feller-violating Heston sets) -- the recentred estimator of
    utils/mc_payoffs.py:61-63 subtracts a sample mean of S_T whose variance does not exist there, noise that the
    per-strike stderr it reports cannot see."""
    import numpy as np
    strikes = np.asarray(strikes, dtype=float)
    ref_put = np.where(np.asarray(types) == "P", ref_prices, ref_prices - (forward - strikes))
    pay = np.maximum(strikes[None, :] - forward * np.exp(np.asarray(x))[:, None], 0.0)
    sd = pay.std(axis=0) / np.sqrt(pay.shape[0])
    return np.abs(pay.mean(axis=0) - ref_put), sd
