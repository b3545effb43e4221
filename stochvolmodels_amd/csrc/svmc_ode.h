// svmc_ode.h -- the coefficient ODE of the LogSV affine expansion, A' = A^T M A + L A + H
// (pricers/logsv/affine_expansion.py:67-205), in the form the transform-grid kernel integrates it: ONE COMPONENT PER LANE.
//
// A grid point's five complex components sit in lanes 0..4 of a 16-lane DPP row.  Component i's derivative is a fixed
// linear combination of the eight products A1 A1, A1 A2, A2 A2, A1 A3, A1 A4, A2 A3, A2 A4, A3 A3 (real coefficients), of
// A1..A4 (complex coefficients) and a constant -- the non-zero entries of :146-182 -- so every lane runs the SAME code on
// its own row of coefficients (OdeLane), and the only cross-lane traffic is the broadcast of A1..A4 ahead of each
// evaluation.  A0 never enters a right-hand side.  A step then costs about 700 vector instructions instead of the 1850 of
// one lane working through all five components, and the launch is latency-bound on exactly that count.
// Host-compilable (tests/native/ode_probe.cpp checks the rows against the CPU twin's right-hand side).
#pragma once
#include "svmc_math.h"

namespace svmc {

struct cd {
    double re, im;
};
SVMC_HD cd C(double re, double im = 0.0) { return cd{re, im}; }
SVMC_HD cd operator+(cd a, cd b) { return cd{a.re + b.re, a.im + b.im}; }
SVMC_HD cd operator-(cd a, cd b) { return cd{a.re - b.re, a.im - b.im}; }
SVMC_HD cd operator-(cd a) { return cd{-a.re, -a.im}; }
SVMC_HD cd operator*(cd a, cd b) { return cd{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
SVMC_HD cd operator*(double s, cd a) { return cd{s * a.re, s * a.im}; }
SVMC_HD cd operator+(cd a, double s) { return cd{a.re + s, a.im}; }
SVMC_HD cd operator+(double s, cd a) { return cd{a.re + s, a.im}; }
SVMC_HD cd operator-(cd a, double s) { return cd{a.re - s, a.im}; }
SVMC_HD cd operator-(double s, cd a) { return cd{s - a.re, -a.im}; }

struct OdeConsts {
    double theta, theta2, vartheta2, qv, qv2, b, eta2, lamda, kappa2_p, kappa_p;
    int spot, second;
};

// pricers/logsv/affine_expansion.py:126-182
static inline OdeConsts make_ode_consts(double theta, double kappa1, double kappa2, double beta, double volvol,
                                        int is_spot_measure, int expansion_order, double eta)
{
    OdeConsts c;
    c.theta = theta;
    c.theta2 = theta * theta;
    c.vartheta2 = beta * beta + volvol * volvol;
    c.qv = theta * c.vartheta2;
    c.qv2 = c.theta2 * c.vartheta2;
    c.b = beta * eta;
    c.eta2 = eta * eta;
    c.spot = is_spot_measure;
    c.second = (expansion_order == 2);
    if (is_spot_measure) {
        c.lamda = 0.0;
        c.kappa2_p = kappa2;
        c.kappa_p = kappa1 + kappa2 * theta;
    } else {
        c.lamda = beta * c.theta2 * eta;
        c.kappa2_p = kappa2 - beta * eta;
        c.kappa_p = kappa1 + kappa2 * theta - 2.0 * beta * theta * eta;
    }
    return c;
}

// one component's row: A_i' = sum_m q[m] P_m + sum_j l[j] A_(j+1) + h,  P = {A1A1, A1A2, A2A2, A1A3, A1A4, A2A3, A2A4, A3A3}
struct OdeLane {
    double q[8];
    cd l[4];
    cd h;
};

// the rows of :146-182 (first-order expansion: components 0..2 and the products of A1, A2 only; second: all five).
// comp outside 0..4 (or 3, 4 at first order) gives the zero row: such lanes ride along and integrate 0' = 0.
SVMC_HD OdeLane make_ode_lane(const OdeConsts &c, cd phi, cd psi, int comp)
{
    OdeLane k;
    for (int m = 0; m < 8; ++m) k.q[m] = 0.0;
    for (int j = 0; j < 4; ++j) k.l[j] = C(0.0);
    k.h = C(0.0);
    const double qv = c.qv, qv2 = c.qv2, v2 = c.vartheta2, th = c.theta, th2 = c.theta2;
    const cd bphi = c.b * phi;
    const cd rhs = (c.spot ? phi * (phi + 1.0) : phi * (phi - 1.0)) - 2.0 * psi;
    const cd L22 = (v2 - 2.0 * c.kappa_p) - 4.0 * th * bphi;
    const cd kb = c.kappa2_p + bphi;
    const bool second = c.second != 0;
    if (comp == 0) {
        k.q[0] = 0.5 * qv2;
        k.l[0] = c.lamda - th2 * bphi;
        k.l[1] = C(qv2);
        k.h = (0.5 * th2 * c.eta2) * rhs;
    } else if (comp == 1) {
        k.q[0] = qv;
        k.q[1] = 2.0 * qv2;
        k.l[0] = -c.kappa_p - 2.0 * th * bphi;
        k.l[1] = 2.0 * ((c.lamda + qv) - th2 * bphi);
        if (second) k.l[2] = C(3.0 * qv2);
        k.h = (th * c.eta2) * rhs;
    } else if (comp == 2) {
        k.q[0] = 0.5 * v2;
        k.q[1] = 4.0 * qv;
        k.q[2] = 2.0 * qv2;
        k.l[0] = -c.kappa2_p - bphi;
        k.l[1] = L22;
        if (second) {
            k.q[3] = 3.0 * qv2;
            k.l[2] = 3.0 * (2.0 * qv - th2 * bphi);
            k.l[3] = C(6.0 * qv2);
        }
        k.h = (0.5 * c.eta2) * rhs;
    } else if (comp == 3 && second) {
        k.q[1] = 2.0 * v2;
        k.q[2] = 4.0 * qv;
        k.q[3] = 6.0 * qv;
        k.q[4] = 4.0 * qv2;
        k.q[5] = 6.0 * qv2;
        k.l[1] = -(2.0 * kb);
        k.l[2] = 3.0 * ((v2 - c.kappa_p) - 2.0 * th * bphi);
        k.l[3] = 4.0 * (3.0 * qv - th2 * bphi);
    } else if (comp == 4 && second) {
        k.q[2] = 2.0 * v2;
        k.q[3] = 3.0 * v2;
        k.q[4] = 8.0 * qv;
        k.q[5] = 12.0 * qv;
        k.q[6] = 8.0 * qv2;
        k.q[7] = 4.5 * qv2;
        k.l[2] = -(3.0 * kb);
        k.l[3] = 2.0 * L22;
    }
    return k;
}

// the lane's derivative from the broadcast A1..A4 (second = expansion order 2, wave-uniform: A3 = A4 = 0 otherwise)
SVMC_HD cd ode_rhs_lane(const OdeLane &k, cd A1, cd A2, cd A3, cd A4, bool second)
{
    const cd A11 = A1 * A1, A12 = A1 * A2, A22 = A2 * A2;
    cd out = k.q[0] * A11 + k.q[1] * A12 + k.q[2] * A22 + k.l[0] * A1 + k.l[1] * A2 + k.h;
    if (second) {
        const cd A13 = A1 * A3, A14 = A1 * A4, A23 = A2 * A3, A24 = A2 * A4, A33 = A3 * A3;
        out = out + k.q[3] * A13 + k.q[4] * A14 + k.q[5] * A23 + k.q[6] * A24 + k.q[7] * A33 + k.l[2] * A3 + k.l[3] * A4;
    }
    return out;
}

}  // namespace svmc
