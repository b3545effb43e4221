#!/usr/bin/env python
"""
Generate the polynomial coefficients of stochvolmodels_amd/csrc/svmc_math.h.

Near-minimax fits by interpolation at Chebyshev nodes in 60-digit arithmetic (mpmath), printed as C hex
doubles.  The fitted "reduced" functions are chosen so that the leading terms are exact in fp64:
  exp:   e^r        = 1 + r + r^2 E(r),              E fitted on r in [-ln2/2, ln2/2]
  log:   log(1+f)   = 2 s + s z G(z), s = f/(2+f), z = s^2,   G fitted on z in [0, (3-2*sqrt2)^2]
  sin:   sin(pi/2 r) = r S(r^2),                     S fitted on z = r^2 in [0, 1/4]
  cos:   cos(pi/2 r) = 1 + r^2 C(r^2),               C fitted on z in [0, 1/4]
Run: python tools/gen_minimax.py
"""
import mpmath as mp

mp.mp.dps = 60


def cheb_fit(f, a, b, deg):
    """interpolate f at deg+1 Chebyshev nodes of [a,b]; return monomial coefficients (low -> high)."""
    n = deg + 1
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    A = mp.matrix(n, n)
    y = mp.matrix(n, 1)
    for i, x in enumerate(xs):
        for j in range(n):
            A[i, j] = x ** j
        y[i] = f(x)
    c = mp.lu_solve(A, y)
    return [c[i] for i in range(n)]


def max_err(f, coefs, a, b, scale=lambda x: 1, m=4001):
    worst = 0
    for k in range(m):
        x = a + (b - a) * mp.mpf(k) / (m - 1)
        p = sum(c * x ** j for j, c in enumerate(coefs))
        worst = max(worst, abs((p - f(x)) * scale(x)))
    return worst


def show(name, coefs):
    print(f"// {name}")
    for j, c in enumerate(coefs):
        print(f"    {float(c).hex()},  // z^{j}  {mp.nstr(c, 20)}")


def E(r):
    return mp.mpf(1) / 2 if r == 0 else (mp.exp(r) - 1 - r) / r ** 2


def G(z):
    if z == 0:
        return mp.mpf(2) / 3
    s = mp.sqrt(z)
    return (2 * mp.atanh(s) - 2 * s) / (s * z)


def S(z):
    if z == 0:
        return mp.pi / 2
    r = mp.sqrt(z)
    return mp.sin(mp.pi / 2 * r) / r


def C(z):
    if z == 0:
        return -(mp.pi / 2) ** 2 / 2
    r = mp.sqrt(z)
    return (mp.cos(mp.pi / 2 * r) - 1) / z


if __name__ == "__main__":
    h = mp.log(2) / 2 * mp.mpf("1.0001")
    for deg in (9, 10):
        c = cheb_fit(E, -h, h, deg)
        print("exp E deg", deg, "abs err of e^r:", mp.nstr(max_err(E, c, -h, h, lambda r: r * r), 5))
    show("EXP_E (deg 9 in r)", cheb_fit(E, -h, h, 9))
    zmax = (3 - 2 * mp.sqrt(2)) ** 2 * mp.mpf("1.0001")
    for deg in (5, 6, 7):
        c = cheb_fit(G, 0, zmax, deg)
        print("log G deg", deg, "abs err of log(1+f):", mp.nstr(max_err(G, c, 0, zmax, lambda z: z * mp.sqrt(z)), 5))
    show("LOG_G (deg 6 in z)", cheb_fit(G, 0, zmax, 6))
    q = mp.mpf(1) / 4
    for deg in (6, 7):
        c = cheb_fit(S, 0, q, deg)
        print("sin S deg", deg, "abs err:", mp.nstr(max_err(S, c, 0, q, lambda z: mp.sqrt(z)), 5))
        c = cheb_fit(C, 0, q, deg)
        print("cos C deg", deg, "abs err:", mp.nstr(max_err(C, c, 0, q, lambda z: z), 5))
    show("SIN_S (deg 6 in z: the 7 coefficients of cossin_diag)", cheb_fit(S, 0, q, 6))
    show("COS_C (deg 6 in z)", cheb_fit(C, 0, q, 6))
    print("ln2_hi", float.hex(0.6931471803691238), "ln2_lo", float(mp.log(2) - mp.mpf(0.6931471803691238)).hex())
    print("log2e", float(1 / mp.log(2)).hex())
