// Black-76 implied volatility of one quote -- shared by the host routine svmc_black_implied_vols (svmc_runtime.hip) and by
// the kernel that ends the fixed-randoms graph (svmc_kernels.hip chain_implied_vols_kernel): the price -> implied-vol
// step of the calibration objective (reference data/option_chain.py:327-346, which delegates to the third-party
// vanilla_option_pricers; parity with that routine is unpinned).  Textbook inversion for 'C' and 'P'.
#pragma once

#include "svmc_math.h"

namespace svmc {

SVMC_HD double black_norm_cdf(double x) { return 0.5 * erfc(-x * 0.70710678118654752440); }

// undiscounted Black price of a call (is_call) or put, its vega and -- for the solver's second-order step -- d1 d2
SVMC_HD double black_undisc(double F, double K, double sqrt_t, double vol, bool is_call, double *vega, double *d1d2 = nullptr)
{
    const double sv = vol * sqrt_t;
    const double d1 = log(F / K) / sv + 0.5 * sv, d2 = d1 - sv;
    if (vega) *vega = F * exp(-0.5 * d1 * d1) * 0.39894228040143267794 * sqrt_t;
    if (d1d2) *d1d2 = d1 * d2;
    // each side from its own tail probabilities: a put taken from the call by parity loses an out-of-the-money put's
    // digits to the cancellation against F - K
    return is_call ? F * black_norm_cdf(d1) - K * black_norm_cdf(d2) : K * black_norm_cdf(-d2) - F * black_norm_cdf(-d1);
}

// implied vol of a DISCOUNTED price on [vol_lo, vol_hi]; NaN when the price is outside the band attainable there
// (or is NaN, or K <= 0).  Solved on the out-of-the-money side (put-call parity), where the price is all time value,
// and on the LOG of the price, which stays well-scaled down to the far tails (price ~ exp(-d^2/2)).  Safeguarded Halley
// iteration on f = ln price(vol) - ln target from the inflection point of price(vol) (Manaster-Koehler) -- f' = vega / p,
// f'' = f' (d1 d2 / vol - f'): the second-order step costs three multiplications on what the evaluation already has and
// takes 3-7 evaluations per quote where Newton took 4-15 (880 quotes, vol 0.05 .. 3, 1 week .. 3 years, strikes 0.4 .. 1.6:
// mean 5.2 against 8.2, worst 13 against 21; every lane of the graph's implied-vol kernel waits for the slowest).  A step that
// leaves the bracket is replaced by bisection, and the bracket always contains the root.  Relative accuracy 1e-13 where the
// price has the digits.
SVMC_HD double black_implied_vol(double price, double K, bool is_call, double forward, double ttm, double discfactor,
                                 double vol_lo, double vol_hi)
{
    const double nan = bits_to_double(0u, 0x7ff80000u);
    const double sqrt_t = sqrt(ttm), target = price / discfactor;
    if (!(K > 0.0) || !(target > black_undisc(forward, K, sqrt_t, vol_lo, is_call, nullptr)) ||
        !(target < black_undisc(forward, K, sqrt_t, vol_hi, is_call, nullptr)))
        return nan;
    const bool otm_call = K >= forward;
    const double otm_target = (is_call == otm_call) ? target : (is_call ? target - (forward - K) : target + (forward - K));
    double a = vol_lo, b = vol_hi;
    double v = sqrt(2.0 * fabs(log(forward / K)) / ttm);
    if (!(v > a && v < b)) v = 0.5 * (a + b);
    if (otm_target > 0.0) {
        const double log_target = log(otm_target);
        for (int it = 0; it < 200; ++it) {
            double vega, d1d2;
            const double p = black_undisc(forward, K, sqrt_t, v, otm_call, &vega, &d1d2);
            const bool pos = p > 0.0;
            const double f = pos ? log(p) - log_target : -1.0;       // p = 0: far below the target
            if (f > 0.0) b = v; else a = v;
            double next = 0.5 * (a + b);
            if (pos && vega > 0.0) {
                // Halley: 2 f f' / (2 f'^2 - f f''); Newton's f / f' where the denominator is not positive
                const double fp = vega / p, fpp = fp * (d1d2 / v - fp);
                const double den = 2.0 * fp * fp - f * fpp;
                const double step = (den > 0.0) ? 2.0 * f * fp / den : f / fp;
                // a step below 1e-13 v is the answer -- tested BEFORE the bracket: the iteration closes in from one
                // side, the other end of the bracket stays where the last overshoot left it, and at the root rounding
                // throws the candidate across the near end; bisecting then would walk away from a converged root
                const double cand = v - step;
                if (fabs(cand - v) <= 1e-13 * v) {
                    v = cand;
                    break;
                }
                if (cand > a && cand < b) next = cand;
            }
            v = next;
            if (b - a <= 1e-15 * v) break;
        }
    } else {
        // time value lost to rounding in the in-the-money quote: plain bisection on the quoted side
        for (int it = 0; it < 200 && b - a > 4e-16 * a; ++it) {
            v = 0.5 * (a + b);
            if (black_undisc(forward, K, sqrt_t, v, is_call, nullptr) < target) a = v; else b = v;
        }
        v = 0.5 * (a + b);
    }
    return v;
}

// utils/mc_payoffs.py:85-88 on one strike's reduced sums [sum (p - shift), sum (p - shift)^2, count]
SVMC_HD void payoff_finalize_one(double s, double s2, double cnt, double shift, double discfactor, double n_path_total,
                                 double *price, double *stderr_)
{
    const double dmean = s / cnt;                     // nanmean of (p - shift); 0/0 -> NaN like NumPy
    const double mean = shift + dmean;
    double var = s2 / cnt - dmean * dmean;            // nanstd^2, ddof = 0 (shift-invariant)
    if (var < 0.0) var = 0.0;
    *price = discfactor * mean;                                                                 // :85
    *stderr_ = discfactor * sqrt(var) / sqrt(n_path_total);                                     // :86-88
}

}  // namespace svmc
