"""
Markovian approximation of the rough (fractional) kernel  K(t) = t^(H-1/2) / Gamma(H+1/2)  by a sum of exponentials
sum_i w_i exp(-x_i t)  on [0, T]: the nodes x_i and weights w_i that LogSvParams.approximate_kernel hands to the rough
LogSV simulator for H <= 0.49 (reference pricers/logsv/logsv_params.py:96-118 -> pricers/rough_logsv/rough_kernel.py:
european_rule :927-1005, optimize_error_l2 :740-867, error_l2_optimal_weights :540-739).

Host-side set-up code (it runs once per parameter set, in microseconds to a second): SciPy's L-BFGS-B over the
log-nodes of the squared L2 error with the optimal weights eliminated in closed form.  Written from the mathematics
of that objective; the control flow of `european_rule` (how the node count and the upper bound on the nodes are grown,
which of three candidate bounds wins) follows the reference step by step, because the rule is DEFINED by that search,
not by a global optimum.  Pinned against the reference's outputs in tests/golden/rough_kernel.npz.

The objective.  With A_ij = a(x_i + x_j), a(s) = (1 - e^{-sT}) / s,  b_i = -2 P(H+1/2, x_i T) / x_i^(H+1/2)  (P the
regularised lower incomplete gamma function) and c = T^(2H) / (2H Gamma(H+1/2)^2):
    || K - sum_i w_i e^{-x_i .} ||^2_{L2[0,T]} = c + w^T A w + b^T w,   minimised over w by  w = -A^{-1} b / 2,
    err(x) = c - b^T v / 4  with  v = A^{-1} b,
    d err / d x_k = v_k ( sum_j a'(x_k + x_j) v_j - b_k' ) / 2,    a'(s) = (-1 + (1 + sT) e^{-sT}) / s^2,
    b_k' = -2 ( (x_k T)^(H+1/2) e^{-x_k T} / Gamma(H+1/2) - (H+1/2) P(H+1/2, x_k T) ) / x_k^(H+3/2).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
from scipy.optimize import lsq_linear, minimize
from scipy.special import gamma, gammainc


def _exp_neg(x: np.ndarray) -> np.ndarray:
    """exp(-x) with everything beyond x = 300 flushed to zero (the reference's exp_underflow, :53-70)"""
    x = np.asarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    ok = x < 300.0
    out[ok] = np.exp(-x[ok])
    return out


def l2_error_optimal_weights(H: float, T: float, nodes: np.ndarray, want_grad: bool = False):
    """squared L2 error on [0, T] with the best weights for the given nodes -> (err, weights) or (err, grad, weights);
    the gradient is with respect to the nodes.  Reference :540-739 (scalar T, H > 0)."""
    nodes = np.array(nodes, dtype=np.float64)
    g1 = gamma(H + 0.5)
    c = T ** (2.0 * H) / (2.0 * H * g1 * g1)
    if nodes.size == 1:
        x = max(1e-4, float(nodes[0]))
        P = gammainc(H + 0.5, x * T)
        e2, e1 = _exp_neg(np.array(2.0 * x * T))[()], _exp_neg(np.array(x * T))[()]
        A = (1.0 - e2) / (2.0 * x)
        b = -2.0 * P / x ** (H + 0.5)
        v = b / A
        err, w = c - 0.25 * b * v, np.array([-0.5 * v])
        if not want_grad:
            return err, w
        dA = (-1.0 + (1.0 + 2.0 * x * T) * e2) / (4.0 * x * x)
        db = -2.0 * ((x * T) ** (H + 0.5) * e1 / g1 - (H + 0.5) * P) / x ** (H + 1.5)
        return err, 0.5 * (dA * v - db) * v, w
    # keep the nodes apart: the smallest at least 1e-4, every next one at least 1 % above its predecessor (:621-627)
    order = np.argsort(nodes)
    x = nodes[order]
    x[0] = max(1e-4, x[0])
    for i in range(x.size - 1):
        if 1.01 * x[i] > x[i + 1]:
            x[i + 1] = 1.01 * x[i]
    nodes = x[np.argsort(order)]
    S = nodes[:, None] + nodes[None, :]
    eS = _exp_neg(S * T)
    A = (1.0 - eS) / S
    P = gammainc(H + 0.5, nodes * T)
    b = -2.0 * P / nodes ** (H + 0.5)
    try:
        v = np.linalg.solve(A, b)
    except np.linalg.LinAlgError:
        v = np.linalg.lstsq(A, b, rcond=None)[0]
    if np.amax(v) > 0.0:                       # a negative weight: fall back to the least-squares solution (:707-708)
        v = lsq_linear(A, b).x
    err = 0.25 * v @ A @ v - 0.5 * np.dot(b, v) + c
    w = -0.5 * v
    if not want_grad:
        return err, w
    dA = (-1.0 + (1.0 + S * T) * eS) / (S * S)
    db = -2.0 * ((nodes * T) ** (H + 0.5) * _exp_neg(nodes * T) / g1 - (H + 0.5) * P) / nodes ** (H + 1.5)
    return err, 0.5 * v * (dA @ v) - 0.5 * db * v, w


def optimize_nodes(H: float, N: int, T: float, tol: float, bound: Optional[float], init_nodes: np.ndarray
                   ) -> Tuple[float, np.ndarray, np.ndarray]:
    """minimise the error over the nodes (in log-coordinates, L-BFGS-B, box [lower bound, bound]) from `init_nodes`
    -> (sqrt of the error, nodes, weights); falls back to the start when the optimiser ends above twice its error.
    Reference optimize_error_l2 :740-867 with method='gradient', force_order=False and given initial nodes."""
    if bound is None:
        bound = 1e100
    lower = 1.0 / (10.0 * N * T) * ((0.5 - H) / 0.4) ** 2
    start = np.minimum(np.maximum(np.asarray(init_nodes, dtype=np.float64), lower), bound)
    err0, w0 = l2_error_optimal_weights(H, T, start)

    def objective(z):
        err, grad, _ = l2_error_optimal_weights(H, T, np.exp(z), want_grad=True)
        return err, np.exp(z) * grad

    res = minimize(objective, np.log(start), jac=True, tol=tol * tol, bounds=((np.log(lower), np.log(bound)),) * N)
    nodes = np.exp(res.x)
    err, w = l2_error_optimal_weights(H, T, nodes)
    if err > 2.0 * max(err0, 1e-9):
        return float(np.sqrt(max(err0, 0.0))), start, w0
    return float(np.sqrt(max(err, 0.0))), nodes, w


def european_rule(H: float, N: int, T: float) -> Tuple[np.ndarray, np.ndarray]:
    """the quadrature rule tuned for European options (reference :927-1005): grow the rule node by node, each time
    raising the upper bound on the nodes by 15 % (5 % once it helps) until the added node has improved the error twice
    in a row without collapsing onto its neighbour; for N = 2, 3 finish by taking the best of three bounds."""
    if not H > 0.0:
        raise NotImplementedError("european_rule is provided for H > 0")
    last = np.array([1.0 / T])

    def shrink(n):                             # the starting point is pulled below the previous optimum
        return 1.03 ** np.minimum(np.arange(1, n + 1) ** 2, 100)

    def solve(n, tol, bnd):
        if n == 1:
            start = np.array([1.0 / T])
        elif last.size == n:
            start = last
        else:
            start = np.concatenate([last, [bnd]])
        return optimize_nodes(H, n, T, tol, bnd, start / shrink(n))

    _, nodes, weights = solve(1, 1e-6, None)
    if N == 1:
        return nodes, weights
    bound = float(np.amax(nodes)) / 1.15
    last, n_cur = nodes, 1
    while n_cur < N:
        improved, step = 0, 1.15
        while improved < 2:
            bound *= step
            err, nodes, weights = solve(n_cur + 1, 1e-7 / n_cur, bound)
            order = np.argsort(nodes)
            nodes, weights = nodes[order], weights[order]
            crowded = (np.amin(nodes[1:] / nodes[:-1]) < 1.4 or abs(np.amin(weights)) < 1e-2
                       or abs(np.amin(weights[1:] / weights[:-1])) < 0.4)
            if crowded:
                improved, step = 0, 1.15
            elif err < solve(n_cur, 1e-7 / n_cur, bound)[0]:
                improved += 1
                if step > 1.06:
                    step = 1.05
                    bound /= 1.15
            else:
                improved, step = 0, 1.15
        n_cur += 1
        last = nodes
    if N >= 4:
        return nodes, weights
    candidates = (2.0 * bound, 3.0 * bound, 4.0 * bound) if N == 2 else (bound, 1.25 * bound, 1.5 * bound)
    best = None
    for bnd in candidates:
        err, nd, wt = solve(N, 1e-8, bnd)
        if best is None or err < best[0]:      # ties go to the smaller bound, as in the reference's <= chain
            best = (err, nd, wt)
    return best[1], best[2]
