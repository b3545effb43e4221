#!/usr/bin/env python
"""
Generate stochvolmodels_amd/csrc/svmc_icdf_table.h: the piecewise-polynomial inverse normal CDF of random stream
version 4 (csrc/svmc_rng.h, DESIGN.md "RNG").

A 32-bit word w is read as a signed integer k = (int32) w.  Stream version 4 (--lattice int, the default): t = k -- the lattice
of magnitudes is |k| = 0 .. 2^31, the two end points (k = 0 and k = -2^31, where -Phi^-1(1/2) = 0) both give z = 0 and every
other magnitude occurs with both signs, so the distribution is exactly symmetric without the add of 1/2 that version 3 paid
per normal (--lattice half: t = k + 1/2, never 0).  The normal is

    z(w) = sign(t) * P_j(|t| - c_j),        the exact function being  sign(t) * -Phi^-1(|t| 2^-32)   (0 at t = 0)

P_j a polynomial of degree DEG on segment j.  Segments: the fp64 representation of |t| splits (0, 2^31) into octaves
[2^e, 2^(e+1)), e = -1 .. 30, each cut into 2^M equal parts -- geometric spacing towards the tail, where Phi^-1 is
singular -- and j = (low 5 bits of the biased exponent) << M | (top M mantissa bits) is read off the high word of |t| with
one shift and one mask.  c_j is the segment's midpoint, or with --edge its lower edge -- which is |t| with the mantissa
bits below the segment bits cleared, so it costs one v_and_b32 instead of 8 table bytes (|t| - c_j is exact in fp64 either
way).  The table is arrays of 16-byte pieces so that a lane fetches its segment with ds_read_b128's off one address:

    midpoint form   piece 0 [j] = {c_j, a0}    piece 1 [j] = {a1, a2}    piece 2 [j] = {a3, a4 (0 for DEG 3)}
    --edge form     piece 0 [j] = {a0, a1}     piece 1 [j] = {a2, a3}    piece 2 [j] = {a4, a4} (DEG 4 only)

Coefficients: interpolation of -Phi^-1 (scipy.special.ndtri, double precision) at the segment's Chebyshev nodes, or, for
the segments of the lowest octaves that hold fewer than DEG + 1 lattice points, exact interpolation of those points.
The evaluation  P = fma(fma(fma(a3, d, a2), d, a1), d, a0)  is part of the stream's DEFINITION (the CPU twin in
oracle/svmc_oracle.c includes this header and evaluates the same expression); tests/test_oracle_golden.py pins the
table against scipy's Phi^-1 on a dense set of words at the accuracy printed below.

    python tools/gen_icdf_table.py --m 5 --deg 3 --raw         (the committed table: stream version 4)
"""
import argparse
import os

import numpy as np
from numpy.polynomial import chebyshev as C
from numpy.polynomial import polynomial as Pn
from scipy.special import ndtri

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def exact(t):
    """the magnitude of the normal at lattice magnitude t = k + 1/2: -Phi^-1(t 2^-32) >= 0"""
    return -ndtri(np.asarray(t, dtype=np.float64) * 2.0 ** -32)


def segment(e, i, m, deg, edge=False, raw=False, off=0.5):
    """-> (c, coefs[deg + 1]) in powers of d = t - c for t in [lo, hi); c = the midpoint, the lower edge, or 0 (raw);
    the lattice points are k + off"""
    width = 2.0 ** e / 2 ** m
    lo = 2.0 ** e * (1.0 + i / 2 ** m)
    hi = lo + width
    c = 0.0 if raw else (lo if edge else lo + 0.5 * width)
    edge = edge or raw
    # lattice points k + off inside [lo, hi)
    k0 = int(np.ceil(lo - off))
    k1 = int(np.ceil(hi - off)) - 1
    npts = max(0, k1 - k0 + 1)
    if npts <= deg + 1:
        coefs = np.zeros(deg + 1)
        if npts > 0:
            t = np.arange(k0, k1 + 1) + off
            coefs[:npts] = Pn.polyfit(t - c, exact(t), npts - 1) if npts > 1 else [exact(t[0])]
        return c, coefs
    j = np.arange(deg + 1)
    xn = np.cos(np.pi * (2 * j + 1) / (2 * (deg + 1)))
    mid = lo + 0.5 * width
    cheb = C.chebfit(xn, exact(mid + 0.5 * width * xn), deg)
    mono = C.cheb2poly(cheb)                                    # powers of xi = (t - mid) / (width / 2)
    coefs = np.array([mono[p] / (0.5 * width) ** p for p in range(deg + 1)])
    if edge:                                                    # re-expand about c: t - mid = (t - c) - (mid - c)
        shifted = np.zeros(deg + 1)
        for p_, a in enumerate(coefs):
            shifted[:p_ + 1] += a * Pn.polypow([-(mid - c), 1.0], p_)
        coefs = shifted
    return c, coefs


def build(m, deg, edge=False, raw=False, off=0.5):
    n = 32 << m
    cen = np.zeros(n)
    co = np.zeros((n, 5))
    # magnitudes: (0, 2^31) on the half lattice (octaves e = -1 .. 30), [1, 2^31] on the integer one (e = 0 .. 31)
    for e in (range(-1, 31) if off else range(0, 32)):
        for i in range(2 ** m):
            j = (((1023 + e) & 31) << m) | i
            if e == 31:
                continue                        # the one magnitude of this octave is 2^31 (k = -2^31): -Phi^-1(1/2) = 0, all zeros
            c, coefs = segment(e, i, m, deg, edge, raw, off)
            cen[j] = c
            co[j, :deg + 1] = coefs
    if not off:
        # t = 0 (k = 0): the high word of 0.0 is 0, i.e. segment 0 -- which is the octave e = 1's first piece and holds the one
        # lattice point |k| = 2.  In powers of |t| itself the line through (0, 0) and (2, z(2)) serves both; P(0) = 0 exactly.
        assert raw and m >= 1, "the integer lattice is generated for the raw form (polynomials in |t| itself)"
        assert np.count_nonzero(co[0, 1:]) == 0
        co[0, :] = 0.0
        co[0, 1] = exact(2.0) / 2.0
    return cen, co


def evaluate(words, cen, co, m, off=0.5):
    """the stream's definition restated in NumPy (fp64, Horner without FMA: differs from the device by rounding only)"""
    k = words.astype(np.uint32).view(np.int32).astype(np.float64)
    t = k + off
    a = np.abs(t)
    hi = (a.view(np.uint64) >> 32).astype(np.uint32)
    j = (hi >> (20 - m)) & ((32 << m) - 1)
    d = a - cen[j]
    p = co[j, 4]
    for q in (3, 2, 1, 0):
        p = p * d + co[j, q]
    return np.copysign(np.abs(p), t)


def max_error(cen, co, m, seed=1, off=0.5):
    rng = np.random.default_rng(seed)
    worst = 0.0
    # dense near every octave edge and uniformly at random; plus the extreme words
    words = [rng.integers(0, 1 << 32, size=1 << 22, dtype=np.uint64).astype(np.uint32),
             np.array([0, 1, 2, 3, 0x7FFFFFFF, 0x7FFFFFFE, 0x80000000, 0x80000001, 0xFFFFFFFF, 0xFFFFFFFE], dtype=np.uint32)]
    for e in range(0, 31):
        base = np.uint32(1 << e)
        words.append((base + rng.integers(0, 1 << e, size=4096, dtype=np.uint64).astype(np.uint32)).astype(np.uint32))
        words.append((-(base + rng.integers(0, 1 << e, size=4096, dtype=np.uint64).astype(np.uint32)).astype(np.int64)).astype(np.uint32))
    for w in words:
        k = w.view(np.int32).astype(np.float64)
        t = k + off
        with np.errstate(divide="ignore"):
            ref = np.where(t == 0.0, 0.0, np.copysign(exact(np.abs(t)), t))
        worst = max(worst, float(np.max(np.abs(evaluate(w, cen, co, m, off) - ref))))
    return worst


def c_double(v):
    return float(v).hex() if np.isfinite(v) else "0.0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=4)
    ap.add_argument("--deg", type=int, default=3, choices=(3, 4))
    ap.add_argument("--edge", action="store_true", help="polynomials in |t| - (segment's lower edge), no centre in the table")
    ap.add_argument("--raw", action="store_true", help="polynomials in |t| itself (no centre at all); implies the edge piece layout")
    ap.add_argument("--lattice", choices=("int", "half"), default="int",
                    help="int: t = k (stream version 4); half: t = k + 1/2 (version 3)")
    ap.add_argument("--out", default=None, help="one output file (A/B variants); default: the product's header "
                    "stochvolmodels_amd/csrc/svmc_icdf_table.h AND the oracle's copy oracle/svo_icdf_table.h (same bytes)")
    args = ap.parse_args()
    outs = [args.out] if args.out else [os.path.join(ROOT, "stochvolmodels_amd", "csrc", "svmc_icdf_table.h"),
                                        os.path.join(ROOT, "oracle", "svo_icdf_table.h")]
    args.edge = args.edge or args.raw
    off = 0.0 if args.lattice == "int" else 0.5
    cen, co = build(args.m, args.deg, args.edge, args.raw, off)
    err = max_error(cen, co, args.m, off=off)
    n = 32 << args.m
    import io
    fh = io.StringIO()
    if True:
        fh.write("// GENERATED by tools/gen_icdf_table.py -- do not edit.  Piecewise-polynomial inverse normal CDF of random stream\n"
                 f"// version {4 if args.lattice == 'int' else 3} (t = k{'' if args.lattice == 'int' else ' + 1/2'}): {n} segments (32 octaves of |t| x 2^{args.m}), degree {args.deg}; max |P - exact| on the 32-bit lattice"
                 f" {err:.2e}\n// (exact = scipy.special.ndtri).  Pieces: " + ("0 = {a0, a1}, 1 = {a2, a3} in powers of |t|" if args.raw else "0 = {a0, a1}, 1 = {a2, a3} in powers of |t| - (lower edge of the segment)" if args.edge else "0 = {c, a0}, 1 = {a1, a2}, 2 = {a3, a4} in powers of |t| - c") + ".\n"
                 "#pragma once\n"
                 f"#define SVMC_ICDF_M {args.m}\n#define SVMC_ICDF_DEG {args.deg}\n#define SVMC_ICDF_SEGMENTS {n}\n#define SVMC_ICDF_EDGE {int(args.edge)}\n#define SVMC_ICDF_RAW {int(args.raw)}\n"
                 f"#define SVMC_ICDF_MAX_ABS_ERROR {err:.3e}\n#define SVMC_ICDF_HALF_LATTICE {int(args.lattice == 'half')}\n")
        pieces = ((0, (0, 1)), (1, (2, 3)), (2, (4, 4))) if args.edge else ((0, None), (1, (1, 2)), (2, (3, 4)))
        if args.edge and args.deg == 3:
            pieces = pieces[:2]
        for piece, cols in pieces:
            fh.write(f"#define SVMC_ICDF_PIECE{piece}_INIT \\\n")
            rows = []
            for j in range(n):
                a, b = (cen[j], co[j, 0]) if cols is None else (co[j, cols[0]], co[j, cols[1]])
                rows.append("{" + c_double(a) + ", " + c_double(b) + "}")
            for s in range(0, n, 4):
                fh.write("    " + ", ".join(rows[s:s + 4]) + (", \\\n" if s + 4 < n else "\n"))
    for path in outs:
        with open(path, "w") as out:
            out.write(fh.getvalue())
    print(f"wrote {outs}: m = {args.m}, degree {args.deg}, {n} segments, {16 * len(pieces) * n} bytes, max abs error {err:.3e}")


if __name__ == "__main__":
    main()
