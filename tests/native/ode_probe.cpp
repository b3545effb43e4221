// Host build of stochvolmodels_amd/csrc/svmc_ode.h for tests/test_math_accuracy.py (g++ only): the one-component-per-lane
// rows of the LogSV coefficient ODE, evaluated for all five components.
#include <stddef.h>
#include "svmc_ode.h"
extern "C" void probe_ode_rhs_lanes(double theta, double kappa1, double kappa2, double beta, double volvol, int is_spot_measure,
                                    int expansion_order, double eta, const double *phi, const double *psi, const double *A,
                                    double *out)
{
    const svmc::OdeConsts c = svmc::make_ode_consts(theta, kappa1, kappa2, beta, volvol, is_spot_measure, expansion_order, eta);
    const svmc::cd ph = {phi[0], phi[1]}, ps = {psi[0], psi[1]};
    svmc::cd a[5];
    for (int i = 0; i < 5; ++i) a[i] = svmc::cd{A[2 * i], A[2 * i + 1]};
    const bool second = expansion_order == 2;
    for (int i = 0; i < 5; ++i) {
        const svmc::OdeLane k = svmc::make_ode_lane(c, ph, ps, i);
        const svmc::cd zero = {0.0, 0.0};
        const svmc::cd o = svmc::ode_rhs_lane(k, a[1], a[2], second ? a[3] : zero, second ? a[4] : zero, second);
        out[2 * i] = o.re;
        out[2 * i + 1] = o.im;
    }
}
