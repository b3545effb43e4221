/*
 * calibration_objective.c -- the inner loop of an MC calibration from plain C, keeping NO randoms: the chain's fixed randoms
 * are the counter-based stream of (seed, call_id), regenerated in registers by every evaluation
 * (svmc_logsv_chain_price_frozen_sets; reference pricers/logsv_pricer.py:244-265, 520-527 keeps RandomState arrays).
 * Prices a 4 x 13 chain for the base point of an optimizer iterate and its six bumped neighbours -- one call, one replayed
 * graph, prices + standard errors + Black implied vols per set -- and times the objective evaluation (one set) and the
 * gradient evaluation (seven sets).  JSON on stdout.
 *
 *   gcc -O2 -Iinclude examples/calibration_objective.c -o calibration_objective -Lstochvolmodels_amd -lsvmc \
 *       -Wl,-rpath,$PWD/stochvolmodels_amd -lm
 *   ./calibration_objective [n_path] [seed] [calls]
 *
 * Set q's numbers are those of svmc_logsv_chain_price(seed, call 0) with set q's parameters, bit for bit, and the Python host's
 * logsv_mc_chain_pricer(seed=...): tests/test_gpu_parity.py::test_c_host_calibration_objective.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "svmc.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != SVMC_OK) {                                                        \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, svmc_last_error()); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

enum { M = 4, K = 13, SETS = 7, ROW = 6 + M };

static double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec;
}

static int cmp_double(const void *a, const void *b)
{
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

static void print_array(const char *name, const double *a, size_t n, int last)
{
    printf("\"%s\": [", name);
    for (size_t i = 0; i < n; ++i) {
        if (isnan(a[i])) printf("%sNaN", i ? ", " : "");
        else printf("%s%.17g", i ? ", " : "", a[i]);
    }
    printf("]%s", last ? "" : ", ");
}

int main(int argc, char **argv)
{
    const size_t n_path = (argc > 1) ? (size_t)strtoull(argv[1], NULL, 10) : 100000;
    const uint64_t seed = (argc > 2) ? strtoull(argv[2], NULL, 10) : 10ull;
    const int calls = (argc > 3) ? atoi(argv[3]) : 200;

    CHECK(svmc_set_device(0));
    /* the chain: expiries 1m, 3m, 6m, 1y; 13 strikes 0.7 .. 1.3 of the forward, puts below it, calls from it on */
    const double ttms[M] = {1.0 / 12, 0.25, 0.5, 1.0}, forwards[M] = {1, 1, 1, 1}, discfactors[M] = {1, 1, 1, 1};
    double strikes[M * K];
    int8_t types[M * K];
    size_t offsets[M + 1];
    for (int i = 0; i < M; ++i) {
        offsets[i] = (size_t)i * K;
        for (int k = 0; k < K; ++k) {
            strikes[i * K + k] = 0.7 + 0.05 * k;
            types[i * K + k] = (k >= 6) ? SVMC_CALL : SVMC_PUT;
        }
    }
    offsets[M] = (size_t)M * K;
    /* the time grid of set_time_grid(ttm_i - ttm_{i-1}, 360 steps per year) (reference utils/funcs.py:108-133) */
    int nb_steps[M];
    double dts[M], t0 = 0.0;
    for (int i = 0; i < M; ++i) {
        const double span = ttms[i] - t0;
        nb_steps[i] = (int)(span * 360.0) + 1;          /* int(ttm * nb_steps_per_year) + 1 points after 0 */
        dts[i] = span / nb_steps[i];
        t0 = ttms[i];
    }
    /* LOGSV_BTC_PARAMS (pricers/logsv_pricer.py:102) and six bumped neighbours: [v0, theta, kappa1, kappa2, beta, volvol, eta_i] */
    double params[SETS * ROW];
    for (int q = 0; q < SETS; ++q) {
        double *p = params + q * ROW;
        p[0] = 0.8376; p[1] = 1.0413; p[2] = 3.1844; p[3] = 3.058; p[4] = 0.1514; p[5] = 1.8458;
        if (q > 0) p[q - 1] *= 1.0 + 1e-4;              /* forward difference in parameter q - 1 */
        for (int i = 0; i < M; ++i) p[6 + i] = 1.0;
    }
    svmc_session_t s;
    CHECK(svmc_session_create(&s, n_path, 8 * M, 8 * M * K));
    double prices[SETS * M * K], stderrs[SETS * M * K], ivols[SETS * M * K];
    double *ms = (double *)malloc(sizeof(double) * (size_t)calls);
    double med[2];
    for (int pass = 0; pass < 2; ++pass) {
        const int n_sets = pass ? SETS : 1;
        for (int it = -2; it < calls; ++it) {           /* the first call captures the graph, the rest replay it */
            const double t = now_ms();
            CHECK(svmc_logsv_chain_price_frozen_sets(s, ttms, forwards, discfactors, M, strikes, types, offsets, n_sets, params,
                                                     /*spot measure*/ 1, SVMC_LOG_RETURN, nb_steps, dts, seed, /*call id*/ 0,
                                                     prices, stderrs, ivols));
            if (it >= 0) ms[it] = now_ms() - t;
        }
        qsort(ms, (size_t)calls, sizeof(double), cmp_double);
        med[pass] = ms[calls / 2];
    }
    /* the base point once more through the on-device-RNG chain driver: the same stream, hence the same bits */
    double base_prices[M * K], base_stderrs[M * K];
    svmc_session_t s1;
    CHECK(svmc_session_create(&s1, n_path, M, M * K));
    CHECK(svmc_logsv_chain_price(s1, ttms, forwards, discfactors, NULL, M, strikes, types, offsets, params[0], params[1], params[2],
                                 params[3], params[4], params[5], 1, 360, SVMC_LOG_RETURN, seed, 0, base_prices, base_stderrs));
    int equal = 1;
    for (int k = 0; k < M * K; ++k) equal = equal && prices[k] == base_prices[k] && stderrs[k] == base_stderrs[k];
    printf("{\"n_path\": %zu, \"steps\": %d, \"calls\": %d, \"one_set_ms\": %.4f, \"seven_sets_ms\": %.4f, "
           "\"base_equals_chain_driver\": %s, ", n_path, nb_steps[0] + nb_steps[1] + nb_steps[2] + nb_steps[3], calls, med[0], med[1],
           equal ? "true" : "false");
    print_array("prices", prices, SETS * M * K, 0);
    print_array("stderrs", stderrs, SETS * M * K, 0);
    print_array("ivols", ivols, SETS * M * K, 1);
    printf("}\n");
    free(ms);
    CHECK(svmc_session_destroy(s1));
    CHECK(svmc_session_destroy(s));
    return equal ? 0 : 2;
}
