"""
Time grid, seeding and timing helpers of the Monte Carlo path
(mirror of the reference's utils/funcs.py: set_time_grid :24-48, set_seed :51-60, timer :63-78,
to_flat_np_array :19-21).
"""
from __future__ import annotations

import functools
import logging
import os
import threading
import time
from typing import List, Optional, Tuple

import numpy as np


def to_flat_np_array(input_list: List[np.ndarray]) -> np.ndarray:
    return np.concatenate(input_list).ravel()


def set_time_grid(ttm: float, nb_steps_per_year: int = 360) -> Tuple[int, float, np.ndarray]:
    """nb_steps = int(ttm*nb_steps_per_year) + 1 (fp64 product, truncated); dt = grid[1] - grid[0] of the
    (nb_steps + 1)-point linspace over [0, ttm] (reference utils/funcs.py:44-47)."""
    nb_steps = int(ttm * nb_steps_per_year) + 1
    grid_t = np.linspace(0.0, ttm, nb_steps + 1)
    dt = grid_t[1] - grid_t[0]
    return nb_steps, float(dt), grid_t


def time_grid_steps(ttm: float, nb_steps_per_year: int = 360) -> Tuple[int, float]:
    """(nb_steps, dt) of set_time_grid without building the grid: the second point of an (nb_steps + 1)-point
    np.linspace over [0, ttm] is 1.0 * (ttm / nb_steps) + 0.0, i.e. exactly ttm / nb_steps (tests/test_host_logic.py checks
    it against set_time_grid bit for bit).  The chain drivers call this once per expiry ahead of their first launch,
    where np.linspace's 5 us per expiry were time with nothing queued on the GPU."""
    nb_steps = int(ttm * nb_steps_per_year) + 1
    return nb_steps, float(ttm) / nb_steps


# ---------------------------------------------------------------------------------------------------
# seeding.  The reference seeds Numba's hidden per-thread MT19937 (set_seed) and its high-level wrappers
# expose no seed.  Here the generators are counter-based (Philox4x32-7 keyed by a 64-bit seed and a
# 24-bit call id, see DESIGN.md "RNG"): `set_seed(value)` fixes the key and rewinds the call counter, an
# un-seeded process starts from OS entropy, and every generator call without an explicit `seed=` consumes
# the next call id -- successive calls draw fresh, but replayable, randoms.
# ---------------------------------------------------------------------------------------------------
_rng_lock = threading.Lock()
_rng_seed = int.from_bytes(os.urandom(8), "little")
_rng_calls = 0


def set_seed(value: int) -> None:
    global _rng_seed, _rng_calls
    with _rng_lock:
        _rng_seed = int(value) & 0xFFFFFFFFFFFFFFFF
        _rng_calls = 0


def get_rng_state() -> Tuple[int, int]:
    """(seed, calls consumed): what dist.init_from_env broadcasts from rank 0 so that all ranks share one stream"""
    with _rng_lock:
        return _rng_seed, _rng_calls


def set_rng_state(seed: int, calls: int) -> None:
    global _rng_seed, _rng_calls
    with _rng_lock:
        _rng_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        _rng_calls = int(calls) & 0xFFFFFF


def next_rng_call(seed: Optional[int] = None) -> Tuple[int, int]:
    """(seed, call_id) for one generator call: explicit seed -> call 0 (pure replay); otherwise the
    process seed and the next call id."""
    global _rng_calls
    if seed is not None:
        return int(seed) & 0xFFFFFFFFFFFFFFFF, 0
    with _rng_lock:
        call = _rng_calls
        _rng_calls = (_rng_calls + 1) & 0xFFFFFF
        if _rng_calls == 0:
            # the call id occupies 24 bits of the Philox counter (include/svmc.h): after 2^24 un-seeded calls under one
            # seed the next call would repeat the randoms of call 0
            import warnings
            warnings.warn("stochvolmodels_amd: 2^24 un-seeded generator calls under one seed -- the call counter wraps and "
                          "the next calls repeat earlier randoms; call set_seed() with a new value", RuntimeWarning)
        return _rng_seed, call


def timer(func):
    """log the wall-clock runtime of the wrapped call at debug level (reference utils/funcs.py:63-78)."""
    @functools.wraps(func)
    def wrapper_timer(*args, **kwargs):
        start = time.perf_counter()
        value = func(*args, **kwargs)
        logging.getLogger(func.__module__).debug("Finished %r in %.4f secs", func.__name__,
                                                 time.perf_counter() - start)
        return value
    return wrapper_timer
