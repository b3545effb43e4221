"""p-values of tools/r04/stream_battery.hip's histograms.

    python tools/r04/stream_battery.py out_r7.bin out_r10.bin [control.bin ...] > profiles/r04_stream_battery.txt

Every test is a chi-square of observed counts against the exact expectation (uniform bins; geometric gap lengths).  Reported:
the statistic, its degrees of freedom, z = (chi2 - dof) / sqrt(2 dof) and the two-sided p-value min(sf, cdf) x 2 -- a generator
is suspect when some p < 1e-4 (too bad OR too good a fit)."""
import sys

import numpy as np
from scipy.stats import chi2

NAMES = ["word r0 high 16 bits", "word r0 low 16 bits", "word r1 high 16 bits", "word r1 low 16 bits", "word r2 high 16 bits",
         "word r2 low 16 bits", "word r3 high 16 bits", "word r3 low 16 bits",
         "pair (r3 of call c, r0 of call c+1), one path, top 8 x top 8 bits  [across the call boundary]",
         "pair (r3 of call c, r0 of call c+1), one path, low 8 x low 8 bits  [across the call boundary]",
         "pair (r1, r2) inside a call (its two time steps), top 8 x top 8 bits",
         "pair (path p, path p+1) at the same call, word r0, top 8 x top 8 bits",
         "pair (path p, path p+1) at the same call, word r3, low 8 x low 8 bits",
         "pair (r0 of call c, r0 of call c+1), one path, top 8 x top 8 bits  [lag one call]"]


def p_two_sided(stat, dof):
    return float(2.0 * min(chi2.sf(stat, dof), chi2.cdf(stat, dof)))


def report(path):
    raw = np.fromfile(path, dtype=np.uint64)
    rounds, lp, lc, seed, n_hist, gap_bins = (int(v) for v in raw[:6])
    hist = raw[6:6 + n_hist * 65536].reshape(n_hist, 65536).astype(np.float64)
    gaps = raw[6 + n_hist * 65536:].astype(np.float64)
    words = 4.0 * 2.0 ** (lp + lc)
    print(f"Philox4x32-{rounds}: 2^{lp} paths x 2^{lc} calls = 2^{lp + lc + 2} words ({words:.3g}), seed {seed}, stream 0, call id 0")
    worst = 1.0
    for i in range(n_hist):
        n = hist[i].sum()
        e = n / 65536.0
        stat = float(((hist[i] - e) ** 2).sum() / e)
        dof = 65535
        p = p_two_sided(stat, dof)
        worst = min(worst, p)
        print(f"  {NAMES[i]:98s} n = 2^{np.log2(n):5.2f}  chi2 = {stat:10.1f}  dof = {dof}  z = {(stat - dof) / np.sqrt(2 * dof):+6.2f}  p = {p:.4f}")
    # gaps of a FINITE sequence (a path's 4 x 2^lc words): a gap of length g is seen between positions i and i + g + 1 only
    # while both fit, so E[count of g] = paths x (len - g - 1) x p^2 q^g exactly (not the untruncated geometric law, whose
    # long gaps a finite walk under-counts by g / len -- 3 % at g = 127 for 4096 words, hundreds of sigmas at these counts)
    n = gaps.sum()
    length, pv, qv = 4.0 * 2.0 ** lc, 1.0 / 16.0, 15.0 / 16.0
    g = np.arange(int(length) - 1, dtype=np.float64)
    e_all = 2.0 ** lp * (length - g - 1.0) * pv * pv * qv ** g
    expect = np.concatenate([e_all[:gap_bins - 1], [e_all[gap_bins - 1:].sum()]])
    stat = float(((gaps - expect) ** 2 / expect).sum())
    p = p_two_sided(stat, gap_bins)
    worst = min(worst, p)
    print(f"  {'gap test on the top 4 bits of the word sequence of a path, visits of [0, 1/16), gaps 0..126 and >= 127':98s} "
          f"n = 2^{np.log2(n):5.2f}  chi2 = {stat:10.1f}  dof = {gap_bins}  z = {(stat - gap_bins) / np.sqrt(2 * gap_bins):+6.2f}  p = {p:.4f}")
    print(f"  smallest two-sided p of the {n_hist + 1} tests: {worst:.3g}   (Bonferroni bound for {n_hist + 1} tests at 1e-4: "
          f"{'PASS' if worst >= 1e-4 else 'FAIL'})\n")
    return worst


if __name__ == "__main__":
    for path in sys.argv[1:]:
        report(path)
