"""
Chain driver shared by the LogSV and Heston Monte Carlo pricers.

Restates the expiry loop of logsv_mc_chain_pricer / logsv_mc_chain_pricer_fixed_randoms /
heston_mc_chain_pricer (reference pricers/logsv_pricer.py:806-867, :1100-1162,
pricers/heston_pricer.py:285-331): one path set, terminal state carried slice to slice, payoff reduction
per slice.  Re-ordered for the GPU / multi-GPU case:

  phase 1  for every expiry ONE kernel: advance the resident state, snapshot the terminal x (and qvar for Q_VAR
           chains) in HBM and reduce the block partials of [sum F*exp(x), count]  -- no host round trip
  phase 2  per-expiry [sum F*exp(x), count]              -> ONE all-reduce over ranks (2*M doubles)
  phase 3  per-strike [sum d, sum d^2, count] of ALL expiries in one launch pair -> ONE all-reduce (3*sum K doubles)
  phase 4  D2H of the sums, host finalisation (utils/mc_payoffs.py:85-88)

The driver is engine-agnostic: `engine` is a HipEngine in the product (GPU only, no fallback); tests may
pass a double to exercise the sharding logic on CPU.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np

from .engine import LOG_RETURN, Q_VAR, option_type_codes, payoff_finalize, payoff_finalize_chain, payoff_shifts


def variable_type_code(variable_type) -> int:
    code = int(getattr(variable_type, "value", variable_type))   # this package's enum, the reference's, or an int
    if code not in (LOG_RETURN, Q_VAR):
        raise NotImplementedError  # VariableType.SIGMA, reference utils/mc_payoffs.py:69-70
    return code


def price_chain_on_engine(engine, comm, n_path_total: int, ttms: np.ndarray, forwards: np.ndarray,
                          discfactors: np.ndarray, strikes_ttms: Sequence[np.ndarray],
                          optiontypes_ttms: Sequence[np.ndarray], variable_type,
                          advance_slice: Callable[[int, float, int, object, int], None],
                          finalize: Callable = payoff_finalize, advance_chain: Callable[[bool, int], None] = None
                          ) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """advance_slice(i, forward_i, snap_row, qvar_row | None, spot_ptr) must leave the state advanced over slice i,
    the terminal x (qvar) in snapshot row snap_row (qvar_row) and [sum F*exp(x), count] at spot_ptr.  A generator that
    can step the whole chain in one launch passes advance_chain(need_qvar, spot_ptr) instead: it must fill snapshot
    rows 0..m-1 (x), m..2m-1 (qvar, when need_qvar) and the 2m spot sums."""
    vt = variable_type_code(variable_type)
    m = len(ttms)
    if not (len(forwards) == len(discfactors) == len(strikes_ttms) == len(optiontypes_ttms) == m):
        raise ValueError("chain arrays must have one entry per maturity")
    need_q = vt == Q_VAR

    # phase 1: stepping + snapshot + local spot sums, state resident.  Queued FIRST: the host-side preparation of the
    # payoff pass (strike arrays, option codes, recentring shifts) then runs while the GPU steps
    engine.reserve_snapshots(m * (2 if need_q else 1))
    spot_ptr, spot_handle = comm.alloc(engine, 2 * m, "spot")
    if advance_chain is not None:
        advance_chain(need_q, spot_ptr)
    else:
        for i in range(m):
            advance_slice(i, float(forwards[i]), i, (m + i) if need_q else None, spot_ptr + 16 * i)
    strikes = [np.ascontiguousarray(np.asarray(k, dtype=np.float64)) for k in strikes_ttms]
    codes = [option_type_codes(t) for t in optiontypes_ttms]          # ValueError on unknown codes
    shifts = [payoff_shifts(k, c, float(f), vt) for k, c, f in zip(strikes, codes, forwards)]

    # phase 2: forward recentring needs the GLOBAL mean of the terminal spots
    comm.all_reduce_sum(engine, spot_handle)

    # phase 3: per-strike payoff sums
    offs = np.concatenate([[0], np.cumsum([3 * len(k) for k in strikes])]).astype(int)
    sums_ptr, sums_handle = comm.alloc(engine, int(offs[-1]) + 1, "payoff")
    engine.payoff_sums_chain(range(m), range(m, 2 * m) if need_q else None, forwards, ttms, spot_ptr,
                             [k.ravel() for k in strikes], [c.ravel() for c in codes], [s.ravel() for s in shifts], vt,
                             sums_ptr)
    comm.all_reduce_sum(engine, sums_handle)

    # phase 4
    sums = comm.to_host(engine, sums_ptr, sums_handle, int(offs[-1]))
    prices, stderrs = [], []
    if finalize is payoff_finalize and m > 1:
        # all expiries in one call into the library (the per-expiry loop below costs ~15 us of interpreter per expiry, time
        # in which the GPU has nothing queued: 120 -> 40 us for C4's 8 x 21 strikes); the same arithmetic, the same bits
        counts = [k.size for k in strikes]
        p_all, e_all = payoff_finalize_chain(sums, np.concatenate([s.ravel() for s in shifts]),
                                             np.repeat(np.asarray(discfactors, dtype=np.float64), counts), float(n_path_total))
        lo = 0
        for i in range(m):
            hi = lo + counts[i]
            shape = np.shape(strikes_ttms[i])
            prices.append(p_all[lo:hi] if len(shape) == 1 else p_all[lo:hi].reshape(shape))
            stderrs.append(e_all[lo:hi] if len(shape) == 1 else e_all[lo:hi].reshape(shape))
            lo = hi
        return prices, stderrs
    for i in range(m):
        p, e = finalize(sums[offs[i]:offs[i + 1]], shifts[i], float(discfactors[i]), float(n_path_total))
        prices.append(p.reshape(np.shape(strikes_ttms[i])))
        stderrs.append(e.reshape(np.shape(strikes_ttms[i])))
    return prices, stderrs
